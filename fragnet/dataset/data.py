"""fragnet.dataset.data -> fragnet_amd.data (reference file: dataset/data.py:877-1032, the two collate functions)."""
from fragnet_amd.data import batch_to, collate_fn, collate_fn_pt  # noqa: F401
