"""fragnet.model.gat.gat2 -> fragnet_amd.model (reference file: model/gat/gat2.py)."""
from fragnet_amd.model import (FragNet, FragNetFineTune, FragNetLayerA, FTHead1, FTHead2, FTHead3, FTHead4,  # noqa: F401
                               FTHead5)
