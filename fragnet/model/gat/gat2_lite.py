"""fragnet.model.gat.gat2_lite -> the gat2_lite variant of fragnet_amd.model (reference file: model/gat/gat2_lite.py)."""
from fragnet_amd.model import FragNetFineTuneLite as FragNetFineTune  # noqa: F401
