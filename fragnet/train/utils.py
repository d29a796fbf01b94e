"""fragnet.train.utils -> fragnet_amd.train (reference file: train/utils.py)."""
from fragnet_amd.train import EarlyStopping, TrainerFineTune, compute_bce_loss  # noqa: F401
