"""fragnet.model.gat.pretrain_heads -> fragnet_amd.model (reference file: model/gat/pretrain_heads.py)."""
from fragnet_amd.model import FragNetPreTrain, PretrainTask  # noqa: F401


def _outside(name):
    class _Outside:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} (masked pretraining, pretrain_heads.py:144-) is outside the accelerated FragNet "
                                      "hot path (SURVEY.md section 8): use FragNetPreTrain")
    _Outside.__name__ = name
    return _Outside


FragNetPreTrainMasked = _outside("FragNetPreTrainMasked")
FragNetPreTrainMasked2 = _outside("FragNetPreTrainMasked2")
