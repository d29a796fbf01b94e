"""fragnet.dataset.dataset -> fragnet_amd.dataset (reference file: dataset/dataset.py:273-292)."""
from fragnet_amd.dataset import FlatMolStore, load_data_parts, load_pickle_dataset  # noqa: F401
