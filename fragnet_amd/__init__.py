"""fragnet_amd: MI355X-native implementation of FragNet's message-passing hot path."""


def prefer_rocblas_for_dense_heads() -> bool:
    """The prediction heads are plain library GEMMs at M = batch size (512).  On MI355X / ROCm 7 torch's default
    hipBLASLt heuristics pick poorly tiled kernels for them (e.g. 64 us for the 1024x1024 weight gradient);
    rocBLAS runs the same head 1.5x faster (tools/headbench.py).  Drivers and bench.py opt in explicitly."""
    import torch
    try:
        torch.backends.cuda.preferred_blas_library("cublas")      # "cublas" = rocBLAS on ROCm builds
        return True
    except Exception:                                              # older torch: keep the default
        return False
