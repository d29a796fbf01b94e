"""fragnet.model.gat.gat2_pretrain -> fragnet_amd.model.FragNetPreTrain (reference twin: model/gat/gat2_pretrain.py:7-27)."""
from fragnet_amd.model import FragNetPreTrain  # noqa: F401
