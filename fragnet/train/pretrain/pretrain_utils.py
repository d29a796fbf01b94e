"""fragnet.train.pretrain.pretrain_utils -> fragnet_amd.train.PretrainTrainer (reference file: train/pretrain/pretrain_utils.py)."""
from fragnet_amd.train import PretrainTrainer as Trainer  # noqa: F401
