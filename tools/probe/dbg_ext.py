import sys, os, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from fragnet_amd import _lib, data, synth
from fragnet_amd.plan import GraphPlan
sys.path.insert(0, "/root/repo/tools")
import molbwd_check as mc
dev = torch.device("cuda", 0)
batch = data.batch_to(data.collate_fn(synth.synth_molecules(64, seed=1000, profile="esol")), dev)
plan = GraphPlan.from_batch(batch)
ext = mc.mol_extents(plan, dev)
torch.cuda.synchronize()
print(ext[:4].cpu())
names = ["a0","na","b0","nb","f0","nf","c0","nc","eb0","meb","ea0","mea","ef0","mef","ec0","mec"]
e = ext.cpu()
for name, (r0i, e0i) in {"bond": (2, 8), "atom": (0, 10), "fbond": (6, 12), "frag": (4, 14)}.items():
    lv = plan.levels[name]
    c = lv.c
    def arr(ptr, n):
        return torch.frombuffer((C.c_int32 * n).from_address(0), dtype=torch.int32) if False else None
    # reconstruct views from the plan arena
    rowptr = plan.rowptr; perm = plan.perm; aux_a = plan.aux_a; aux_b = plan.aux_b
    base = plan.rowptr.data_ptr()
    def view(t, ptr, n):
        off = (ptr - t.data_ptr()) // 4
        return t[off: off + n].cpu()
    rpd = view(plan.rowptr, c.rowptr_d, lv.n + 1) - c.pos_base_d
    rps = view(plan.rowptr, c.rowptr_s, lv.n + 1) - c.pos_base_s
    dst_s = view(plan.aux_a, c.dst_s, lv.m)
    dpos_s = view(plan.aux_b, c.dpos_s, lv.m)
    bad = 0
    for k in range(e.shape[0]):
        r0, nr, e0, me = int(e[k, r0i]), int(e[k, r0i + 1]), int(e[k, e0i]), int(e[k, e0i + 1])
        if rpd[r0] != e0 or rpd[r0 + nr] != e0 + me or rps[r0] != e0 or rps[r0 + nr] != e0 + me:
            bad += 1
            if bad < 4: print(name, "mol", k, "extent mismatch", r0, nr, e0, me, int(rpd[r0]), int(rpd[r0+nr]), int(rps[r0]), int(rps[r0+nr]))
            continue
        ds = dst_s[e0:e0 + me]; dp = dpos_s[e0:e0 + me]
        if me and (ds.min() < r0 or ds.max() >= r0 + nr or dp.min() < e0 or dp.max() >= e0 + me):
            bad += 1
            if bad < 4: print(name, "mol", k, "closure violated")
    print(name, "n", lv.n, "m", lv.m, "m_real", lv.m_real, "bad", bad, "last ext end", int(e[-1, r0i] + e[-1, r0i+1]), int(e[-1, e0i] + e[-1, e0i+1]))
