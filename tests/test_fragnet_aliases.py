"""The ``fragnet.*`` import paths the reference's drivers use resolve to this implementation (names as imported at
fragnet/train/finetune/finetune_gat2.py:2-9,121,144,166,216 and fragnet/train/pretrain/pretrain_gat2.py:3-17)."""
import importlib
import os
import pickle
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _this_repo_first():
    sys.path.insert(0, ROOT)
    for k in [k for k in sys.modules if k == "fragnet" or k.startswith("fragnet.")]:
        del sys.modules[k]
    yield
    sys.path.remove(ROOT)


def test_driver_imports_resolve_to_fragnet_amd():
    import fragnet_amd.data
    import fragnet_amd.model
    import fragnet_amd.train
    # finetune_gat2.py
    from fragnet.dataset.dataset import load_pickle_dataset                     # :2
    from fragnet.train.utils import EarlyStopping                              # :4
    from fragnet.dataset.data import collate_fn                                # :6
    from fragnet.train.utils import TrainerFineTune as Trainer                 # :9
    from fragnet.model.gat.gat2 import FragNetFineTune                         # :121
    from fragnet.model.gat.gat2_lite import FragNetFineTune as Lite            # :144
    from fragnet.model.gat.gat2_edge import FragNetFineTune as Edge            # :166
    from fragnet.model.gat.gat2_pretrain import FragNetPreTrain as PT2         # :216
    # pretrain_gat2.py
    from fragnet.dataset.dataset import load_data_parts                        # :6
    from fragnet.model.gat.pretrain_heads import FragNetPreTrain, FragNetPreTrainMasked, FragNetPreTrainMasked2   # :12
    from fragnet.train.pretrain.pretrain_utils import Trainer as PTrainer      # :13
    from fragnet.dataset.data import collate_fn_pt                             # :15
    from fragnet.model.gat.gat2 import FragNet                                 # :16
    assert importlib.import_module("fragnet").__file__.startswith(ROOT)
    assert FragNetFineTune is fragnet_amd.model.FragNetFineTune and FragNet is fragnet_amd.model.FragNet
    assert Lite is fragnet_amd.model.FragNetFineTuneLite and Edge is fragnet_amd.model.FragNetFineTuneEdge
    assert PT2 is FragNetPreTrain is fragnet_amd.model.FragNetPreTrain
    assert collate_fn is fragnet_amd.data.collate_fn and collate_fn_pt is fragnet_amd.data.collate_fn_pt
    assert Trainer is fragnet_amd.train.TrainerFineTune and PTrainer is fragnet_amd.train.PretrainTrainer
    assert EarlyStopping is fragnet_amd.train.EarlyStopping
    assert callable(load_pickle_dataset) and callable(load_data_parts)
    for cls in (FragNetPreTrainMasked, FragNetPreTrainMasked2):
        with pytest.raises(NotImplementedError, match="outside the accelerated"):
            cls()
    # constructor signatures the drivers call (finetune_gat2.py:122-135, pretrain_gat2.py:118-127)
    m = FragNetFineTune(n_classes=1, atom_features=167, frag_features=167, edge_features=17, num_layer=2, drop_ratio=0.1,
                        num_heads=4, emb_dim=128, h1=32, h2=32, h3=32, h4=32, act="relu", fthead="FTHead3")
    p = FragNetPreTrain(num_layer=2, drop_ratio=0.1, num_heads=4, emb_dim=128, atom_features=167, frag_features=167, edge_features=17)
    assert hasattr(m, "pretrain") and hasattr(m, "fthead") and hasattr(p, "pretrain") and hasattr(p, "head")
    # the fine-tune driver copies the pretrained encoder over by state dict (finetune_gat2.py:228-229)
    m.pretrain.load_state_dict(p.pretrain.state_dict())


def test_pickled_dataset_loaders(tmp_path):
    from fragnet.dataset.dataset import load_data_parts, load_pickle_dataset
    from fragnet_amd import synth
    mols = synth.synth_molecules(5, seed=3, profile="esol")
    for i, part in enumerate((mols[:2] + [None], mols[2:])):
        with open(tmp_path / f"train_{i}.pkl", "wb") as f:
            pickle.dump(part, f)
    with open(tmp_path / "val_0.pkl", "wb") as f:
        pickle.dump(mols[:1], f)
    assert len(load_pickle_dataset(tmp_path / "train_0.pkl")) == 2          # the None entry is dropped
    assert len(load_data_parts(str(tmp_path), select_name="train")) == 5
    assert len(load_data_parts(str(tmp_path))) == 6
