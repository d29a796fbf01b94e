"""Host-side helpers of the measurement tooling (no GPU): kernel-name matching of the PMC summary, physical-core count."""
import importlib.util
import os

from tests.conftest import ROOT


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_summary_tells_the_two_forward_variants_apart():
    t = _load("tools/pmc_traffic.py", "pmc_traffic")
    plain = "void (anonymous namespace)::k_gat_fwd<4, 1, false>((anonymous namespace)::GatFwdArgs)"
    o2 = "void (anonymous namespace)::k_gat_fwd<4, 1, true>((anonymous namespace)::GatFwdArgs)"
    assert t._match(plain, "k_gat_fwd", False) and not t._match(plain, "k_gat_fwd", True)
    assert t._match(o2, "k_gat_fwd", True) and not t._match(o2, "k_gat_fwd", False)
    assert t._match("_ZN12_GLOBAL__N_19k_gat_fwdILi4ELi1ELb1EEEvNS_10GatFwdArgsE", "k_gat_fwd", True)
    assert t._match("_ZN12_GLOBAL__N_19k_gat_fwdILi4ELi1ELb0EEEvNS_10GatFwdArgsE", "k_gat_fwd", False)
    assert not t._match("void (anonymous namespace)::k_gat_fwd_pair<4, 1, 8, true, true>(...)", "k_gat_fwd", True)
    assert t._match("void (anonymous namespace)::k_gat_bwd_one<4, 1, 8>(...)", "k_gat_bwd_one", None)
    assert not t._match("void (anonymous namespace)::k_gat_bwd_one3<4, 8>(...)", "k_gat_bwd_one", None)
    # round 6: the forward's third argument is a KIND (0 plain, 1 / 2 with the second output), the backward's fifth an engine-constant flag
    assert t._match("void (anonymous namespace)::k_gat_fwd<4, 1, 2>(fni::GatFwdArgs)", "k_gat_fwd", True)
    assert t._match("void (anonymous namespace)::k_gat_fwd<4, 1, 0>(fni::GatFwdArgs)", "k_gat_fwd", False)
    assert t._match("_ZN12_GLOBAL__N_19k_gat_fwdILi4ELi1ELi2EEEvN3fni10GatFwdArgsE", "k_gat_fwd", True)
    assert t._match("void (anonymous namespace)::k_gat_bwd_one<4, 1, 8, false, true>(...)", "k_gat_bwd_one", False)      # engine-constant, not deferred
    assert t._match("void (anonymous namespace)::k_gat_bwd_one<4, 1, 8, true>(...)", "k_gat_bwd_one", True)
    assert t._match("void (anonymous namespace)::k_gat_bwd_one<4, 1, 8>(...)", "k_gat_bwd_one", False)
    assert t._match("_ZN12_GLOBAL__N_113k_gat_bwd_oneILi4ELi1ELi8ELb0ELb1EEEvN3fni13GatBwdOneArgsE", "k_gat_bwd_one", False)


def test_physical_core_count_is_a_positive_integer_not_above_the_logical_count():
    b = _load("bench.py", "bench_mod")
    n = b._physical_cores()
    assert isinstance(n, int) and 1 <= n <= (os.cpu_count() or 1)


def test_whole_step_traffic_table_joins_the_step_sequence(tmp_path):
    """tools/pmc_step_traffic.py --join: every kernel of the committed whole-step PMC table gets its in-step duration and the
    bandwidth it moved (MB / us = TB/s); a sequence that does not match the counter passes is reported, not joined."""
    t = _load("tools/pmc_step_traffic.py", "pmc_step_traffic")
    table = {"hbm_GB": 0.3, "sequence": [{"kernel": "k_a", "workgroups": 10, "hbm_MB": 100.0}, {"kernel": "k_b", "workgroups": 20, "hbm_MB": 200.0}]}
    seq = tmp_path / "seq.txt"
    seq.write_text("# one step = ...: 2 kernels, 75.0 us wall\n  0 k_a   blocks     10 dur   25.00 us  gap   0.00 us\n"
                   "  1 k_b   blocks     20 dur   50.00 us  gap   0.00 us\n# GPU busy 75.0 us\n")
    out = t.join_sequence(table, str(seq))
    assert [e["us_in_step"] for e in out["sequence"]] == [25.0, 50.0]
    assert [e["moved_TBps"] for e in out["sequence"]] == [4.0, 4.0] and out["moved_TBps_whole_step"] == 4.0
    seq.write_text("  0 k_a   blocks     11 dur   25.00 us  gap   0.00 us\n  1 k_b   blocks     20 dur   50.00 us  gap   0.00 us\n")
    bad = t.join_sequence({"hbm_GB": 0.3, "sequence": [dict(e) for e in table["sequence"]]}, str(seq))
    assert "not joined" in bad["durations_from"]


def test_committed_whole_step_table_is_consistent():
    """profiles/r04_pmc_step.json: the per-kernel bytes add up to the totals and the traffic ratio it states."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_step.json")))
    assert d["kernels_per_step"] == len(d["sequence"])
    assert abs(sum(e["hbm_MB"] for e in d["sequence"]) / 1e3 - d["hbm_GB"]) < 2e-3
    assert abs(d["hbm_GB"] / d["step_algorithmic_GB"] - d["traffic_over_algorithmic"]) < 2e-3


def test_isa_mix_classifies_instructions_and_finds_loops():
    """tools/isa_mix.py: the class of an instruction (DPP forms by their operand modifiers, LDS broadcasts, Philox's integer
    multiplies, scalar bookkeeping) and the loops of a kernel body (a backward branch to a label; a second back edge into the same
    body is the same loop)."""
    m = _load("tools/isa_mix.py", "isa_mix")
    assert m.classify("global_load_dwordx4", False) == "vmem load"
    assert m.classify("ds_bpermute_b32", False) == "lds / bpermute"
    assert m.classify("v_add_f32_e32", True) == "dpp / lane"
    assert m.classify("v_fmac_f32_e32", False) == "fp32 fma / mul / add"
    assert m.classify("v_mul_hi_u32", False).startswith("int multiply")
    assert m.classify("v_cndmask_b32_e32", False) == "fp32 max / min / cmp-select"
    assert m.classify("s_and_saveexec_b64", False) == "scalar ALU"
    assert m.classify("s_waitcnt", False) == "s_waitcnt"
    body = ["k:", "\ts_mov_b32 s0, 0", ".LBB0_1:", "\tv_add_f32_e32 v0, v1, v2", ".LBB0_2:", "\tglobal_load_dword v3, v[4:5], off",
            "\ts_cbranch_scc1 .LBB0_2", "\tv_exp_f32_e32 v0, v0", "\ts_cbranch_vccnz .LBB0_1", "\ts_endpgm"]
    loops = m.loops(body)
    assert loops[0] == (2, 8)                     # the outer loop first (largest body)
    c, vb = m.mix(body, *loops[0])
    assert c["vmem load"] == 1 and c["transcendental"] == 1 and c["fp32 fma / mul / add"] == 1 and vb["vmem load x4B"] == 1


def test_committed_round6_tables_are_consistent():
    """profiles/r06_pmc_step.json: per-kernel bytes add up to the totals, and the table is stamped with the source digest it was
    collected under (bench.py drops it when that differs from the library's)."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_step.json")))
    assert d["kernels_per_step"] == len(d["sequence"]) == 35
    assert abs(sum(e["hbm_MB"] for e in d["sequence"]) / 1e3 - d["hbm_GB"]) < 2e-3
    assert len(d["source_digest"]) == 64
