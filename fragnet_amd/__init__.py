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


def tune_library_gemms(max_ms: int = 30, filename: str = None) -> bool:
    """Lets PyTorch's TunableOp time the rocBLAS / hipBLASLt solutions for every dense head product the first time a
    shape is seen (M = batch size: 14 shapes for FTHead3, ~15 s once) and use the fastest from then on: +1.2 % on the
    whole ESOL step over the library heuristics.  The first (eager) steps do the tuning, so call this before a
    graphstep.GraphedTrainStep is captured.  ``filename``: reuse / store the selections across runs."""
    import torch
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(int(max_ms))
        if not filename:             # TunableOp writes its selections at exit: keep them out of the working directory
            import os, tempfile
            filename = os.path.join(tempfile.gettempdir(), f"fragnet_amd_tunableop_{os.getuid()}.csv")
        tunable.set_filename(filename, insert_device_ordinal=True)
        return True
    except Exception:                                              # older torch: library heuristics
        return False


def graph_capture_head(model, batch_size: int, attr: str = "fthead") -> bool:
    """Replaces ``model.<attr>`` (an MLP head with a static [batch_size, 256] input) by a HIP-graph-captured
    callable (torch.cuda.make_graphed_callables): its ~45 small launches per step become two graph launches.
    Call AFTER the optimiser has re-pointed the parameters (parallel.FlatAdam).  Eval mode and other batch sizes
    keep running eagerly.  Returns False (and leaves the head untouched) if capture is not possible."""
    import torch
    head = getattr(model, attr)
    dev = next(head.parameters()).device
    if dev.type != "cuda":
        return False
    if getattr(head, "rng", None) is not None:
        head.rng = None      # torch's own dropout is graph-safe; the fused Philox epilogue would replay one fixed mask here
    try:
        was_training = head.training
        head.train()
        sample = torch.randn(batch_size, head_input_width(head), device=dev, requires_grad=True)
        graphed = torch.cuda.make_graphed_callables(head, (sample,))
        head.train(was_training)
    except Exception as exc:      # pragma: no cover - depends on the runtime
        import warnings
        warnings.warn(f"head graph capture failed, running it eagerly: {exc}")
        return False
    # make_graphed_callables returns the same module with a patched forward: keep it in place, so state_dict keys,
    # parameter identities and checkpoints are unchanged
    setattr(model, attr, graphed)
    return True


def head_input_width(head) -> int:
    import torch
    for m in head.modules():
        if isinstance(m, torch.nn.Linear):
            return m.in_features
    raise ValueError("head has no Linear layer")
