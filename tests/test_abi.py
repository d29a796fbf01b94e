"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly the entry
points include/fragnet_hip.h declares; CPU tensors are refused (no fallback)."""
import os
import re

import pytest
import torch

from tests.conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "fragnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from fragnet_amd import _lib
    from fragnet_amd.build import build_lib
    build_lib()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in fragnet_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree"
    assert lib.fn_abi_version() == _lib.ABI_VERSION == 12


def test_library_exports_nothing_but_the_declared_entry_points():
    """Every defined function symbol of the .so is a declared fn_* entry point (helpers must be static: a namespace-scope
    function inside the extern "C" block is exported under its plain name)."""
    import shutil
    import subprocess
    from fragnet_amd.build import OUT, build_lib
    build_lib()
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", OUT], capture_output=True, text=True, check=True).stdout
    funcs = sorted(line.split()[-1] for line in out.splitlines() if len(line.split()) == 3 and line.split()[1] == "T")
    assert funcs == _declared(), f"exported but not declared: {sorted(set(funcs) - set(_declared()))}"


def test_plan_layout_is_host_side_and_validates():
    import ctypes as C
    from fragnet_amd import _lib
    lib = _lib.load()
    tasks = (_lib.CsrTask * 16)()
    tasks[0] = _lib.CsrTask(None, None, 10, 3, 5, 0, 0, 0, -1)
    tasks[1] = _lib.CsrTask(None, None, 7, 0, 4, 0, 0, 0, -1)
    ti, ts = C.c_int64(), C.c_int64()
    assert lib.fn_plan_layout(tasks, 2, C.byref(ti), C.byref(ts)) == 0
    assert (ti.value, ts.value) == (20, 9)
    assert (tasks[1].item_base, tasks[1].seg_base) == (13, 5)
    assert lib.fn_plan_layout(tasks, 17, C.byref(ti), C.byref(ts)) == -3
    assert b"FN_MAX_TASKS" in lib.fn_last_error()
    tasks[0].n_real = -1
    assert lib.fn_plan_layout(tasks, 1, C.byref(ti), C.byref(ts)) == -1


def test_argument_errors_do_not_touch_the_gpu():
    from fragnet_amd import _lib
    lib = _lib.load()
    assert lib.fn_segment_sum_f32(None, 128, None, None, 0, None, 4, 128, 16, None) == -1
    assert lib.fn_row_dots_sorted_f32(None, None, 128, 0, 9, None, None, None) == -1
    assert lib.fn_dropout_act_f32(None, None, 8, 1.5, 0, 0, None, 1, None) == -1


def test_cpu_tensors_are_refused_not_silently_computed():
    from fragnet_amd import _lib, ops
    with pytest.raises(_lib.FragnetHipError):
        ops.scatter_add(torch.ones(4, 2), torch.tensor([0, 1, 0, 1]))
    from fragnet_amd import data, synth
    from fragnet_amd.model import FragNetFineTune
    model = FragNetFineTune(num_layer=1, h1=8, h2=8, h3=8, h4=8)
    batch = data.collate_fn(synth.synth_molecules(2, seed=1))
    with pytest.raises(_lib.FragnetHipError):
        model(batch)


def test_state_dict_layout_matches_reference_key_order():
    """Same module tree as the reference => same keys in the same order (tests/golden pkeys)."""
    from fragnet_amd.model import FragNetFineTune, FragNetPreTrain
    from tests.helpers import check_params_match, load_case
    for case, cls in (("ft_esol_b8", FragNetFineTune), ("ft_tox21_b4", FragNetFineTune), ("pt_esol_b4", FragNetPreTrain)):
        cfg, _, _, _, pkeys, psums = load_case(case)
        torch.manual_seed(cfg["seed"])
        check_params_match(cls(**cfg["ctor"]), pkeys, psums)
