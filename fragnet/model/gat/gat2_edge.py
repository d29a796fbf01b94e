"""fragnet.model.gat.gat2_edge -> the gat2_edge variant of fragnet_amd.model (reference file: model/gat/gat2_edge.py)."""
from fragnet_amd.model import FragNetFineTuneEdge as FragNetFineTune, FragNetLayerEdge as FragNetLayerA  # noqa: F401
