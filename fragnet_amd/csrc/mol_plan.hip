// Graph plan of a molecule-contiguous batch in ONE launch (fn_plan_build_mol).
//
// fn_plan_build sorts every CSR of the batch with four grid-wide passes (histogram -> look-back scan -> unordered fill -> rank
// sort: four dependent launches, ~40 us at ESOL batch 512 whatever the size, because each is a latency-floor kernel).  collate_fn
// concatenates molecules (reference dataset/data.py:877-948), so every key of molecule i is smaller than every key of molecule
// i + 1 and the stable sort of a whole CSR is the concatenation of the stable sorts of its molecules: with the per-molecule
// offsets of the seven index spaces at hand (the collate has them: they ARE its cumulative counts) a workgroup can sort its own
// molecule in LDS and write the finished slices -- no global histogram, no scan, no atomics on global memory.
//
//   * molecule workgroups (one per molecule, 8 waves): two global round trips (the molecule's extents; the keys of every CSR
//     task -> LDS), then LDS-only phases that all tasks walk in step, each task on its own group of 4 / 2 / 1 waves
//     (by-destination and by-source order of a graph are two tasks; small tasks share a wave): counts per node (LDS atomics)
//     -> exclusive scan -> unordered fill -> rank of every item among its segment's ids (ascending original id = the
//     reference's sequential scatter order; the same rule as k_plan_ranksort) -> perm / inverse permutation to their global
//     slices.  After one more barrier the two orders of each graph exchange their cross references (other endpoint, position
//     in the destination order and back) through LDS.
//   * padding (static-shape batches, fn_stage_padded): items behind the real ones point at the last `pad_mod` nodes of their
//     target space, item c at node hi - (c - n_real) % pad_mod, as self-loops.  Counts, positions and permutation of that tail
//     are closed forms of (n_real, capacity, pad_mod): extra workgroups write them, one thread per item / node.
//
// Bit-exact against the stable argsort (tests/test_gpu_parity.py, tests/test_gpu_plan_mol.py) like fn_plan_build; a key outside
// its molecule's node range sets status bit 1 (value 2: "not molecule-contiguous"), a molecule beyond the LDS tile bit 2 (4).
#include <algorithm>

#include "fn_internal.h"

namespace {

using fni::fail;
using fni::launch_status;

constexpr int kMpThreads = 512, kMpWaves = kMpThreads / 64;

struct MpTask {
    fn_csr_task t;
    int node_space, item_space;
    int lds_off;                    // first 4-byte word of the task's LDS slice
    int cap_items, cap_nodes;       // per molecule
};
struct MpArgs {
    MpTask t[FN_MAX_TASKS];
    int n_tasks, n_spaces;          // t[] is in SCHEDULE order (largest first); `partner` indexes t[]
    int first_wave[FN_MAX_TASKS], n_waves[FN_MAX_TASKS];      // the waves of the molecule's workgroup that work on task e
    const int32_t* off;             // [n_spaces][n_mols + 1]
    int n_mols;
    const int32_t* counts_dev;      // nullable: number of real molecules
    int64_t cap[FN_MAX_SPACES], pad_mod[FN_MAX_SPACES];
    int32_t *rowptr, *perm, *aux_a, *aux_b, *aux_c, *status;
    int pad_blocks;
    unsigned long long* stamps;     // nullable profiling aid (fn_debug_set_stamps): 16 values per molecule workgroup
};

// slices of one task inside its LDS words: cnt [cap_nodes + 1] | cur [cap_nodes + 1] | key16 [cap_items] | tmp16 | pos16
struct MpSlice {
    int32_t *cnt, *cur;
    uint16_t *key, *tmp, *pos;
};
__device__ __forceinline__ MpSlice mp_slice(int32_t* lds, const MpTask& T) {
    MpSlice s;
    s.cnt = lds + T.lds_off;
    s.cur = s.cnt + T.cap_nodes + 1;
    s.key = reinterpret_cast<uint16_t*>(s.cur + T.cap_nodes + 1);
    s.tmp = s.key + T.cap_items;
    s.pos = s.tmp + T.cap_items;
    return s;
}
__host__ __device__ inline int mp_slice_words(int cap_items, int cap_nodes) {
    return 2 * (cap_nodes + 1) + (3 * cap_items + 1) / 2 + 1;
}

struct MpMol { int n0, nn, e0, ne, L, loops; };       // the molecule's nodes / items of a task
// ext: the molecule's [first, end) in every index space (LDS, loaded once per workgroup)
__device__ __forceinline__ MpMol mp_mol(const MpTask& T, const int32_t* ext) {
    MpMol m;
    m.n0 = ext[2 * T.node_space];  m.nn = ext[2 * T.node_space + 1] - m.n0;
    m.e0 = ext[2 * T.item_space];  m.ne = ext[2 * T.item_space + 1] - m.e0;
    m.loops = T.t.n_loops > 0 ? 1 : 0;
    m.L = m.ne + (m.loops ? m.nn : 0);
    return m;
}
__device__ __forceinline__ bool mp_fits(const MpTask& T, const MpMol& m) { return m.L <= T.cap_items && m.nn <= T.cap_nodes; }

// inclusive prefix sum over the 64 lanes with DPP row operations (LLVM's scan idiom for GCN / CDNA: shifts 1, 2, 4, 8 inside
// each row of 16, then row_bcast:15 / row_bcast:31 carry the row totals over) -- six VALU adds instead of six ds_bpermute
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int mp_dpp(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int mp_wave_scan(int x) {
    x += mp_dpp<0x111, 0xF>(x);          // row_shr:1
    x += mp_dpp<0x112, 0xF>(x);          // row_shr:2
    x += mp_dpp<0x114, 0xF>(x);          // row_shr:4
    x += mp_dpp<0x118, 0xF>(x);          // row_shr:8
    x += mp_dpp<0x142, 0xA>(x);          // row_bcast:15 into rows 1 and 3
    x += mp_dpp<0x143, 0xC>(x);          // row_bcast:31 into rows 2 and 3
    return x;
}

// The molecule's workgroup.  Round trip 1 (with the copy of the argument block): the molecule's extents; every wave then looks up ITS tasks once (the schedule gives
// a task a group of 4 / 2 / 1 waves: the two orders of the bond graph, ~390 items each, take four waves each, small tasks share
// a wave) and keeps their parameters in registers.  Round trip 2: the keys of its tasks -> LDS (all loads in flight together,
// none inside a divergent branch); then LDS-only phases that all tasks walk in step: counts (LDS atomics) | scan by the
// group's first wave, rowptr | unordered fill | rank of each item among its segment's ids = final position, perm and the
// inverse permutation | cross references of the two orders of each graph.
constexpr int kMpMine = 4;          // tasks per wave (the schedule never gives a wave more)
// Everything but `lg` is the same in all lanes: it is forced into scalar registers (readfirstlane) -- four such contexts in
// vector registers made the kernel 121 VGPRs = two workgroups per CU, and 548 workgroups then ran in two rounds.
struct MpCtx {
    int32_t *cnt, *cur;
    uint16_t *key, *tmp, *pos, *pkey, *ppos;       // this task's slice; the partner's keys / positions (cross references)
    const int64_t* gkey;
    int n0, nn, e0, ne, L, lg, stride, role;
    int base, item_base, pitem_base, seg_base, n_real;             // positions and counts are below 2^31 (fn_plan_layout)
    bool first, ok;
};
__device__ __forceinline__ int mp_u(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename P> __device__ __forceinline__ P* mp_up(P* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    return reinterpret_cast<P*>(((uint64_t)(uint32_t)mp_u((int)(a >> 32)) << 32) | (uint32_t)mp_u((int)(uint32_t)a));
}

__device__ void mp_molecule(const MpArgs& A, int mol, int32_t* lds) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned long long* sp = A.stamps ? A.stamps + (size_t)mol * 16 : nullptr;
    int si = 0;
    auto stamp = [&]() { if (sp && tid == 0 && si < 14) sp[si] = __builtin_amdgcn_s_memtime();  ++si; };
    if (sp && tid == 0) sp[14] = wall_clock64();
    stamp();
    int32_t* ext = lds;                                            // [FN_MAX_SPACES][2], written by the kernel's first lines
    int32_t* tiles = lds + 2 * FN_MAX_SPACES;
    stamp();
    // ---- this wave's tasks
    MpCtx c[kMpMine];
    int n_my = 0;
    {
        const int fw = lane < A.n_tasks ? A.first_wave[lane] : 99, nw = lane < A.n_tasks ? A.n_waves[lane] : 0;
        uint64_t mask = __ballot(w >= fw && w < fw + nw);
#pragma unroll
        for (int q = 0; q < kMpMine; ++q) {
            c[q].ok = false;  c[q].L = 0;  c[q].nn = 0;  c[q].role = FN_ROLE_PLAIN;  c[q].first = false;
            if (!mask) continue;
            const int e = __ffsll((unsigned long long)mask) - 1;
            mask &= mask - 1;
            n_my = q + 1;
            const MpTask& T = A.t[e];
            const MpMol m = mp_mol(T, ext);
            const MpSlice s = mp_slice(tiles, T);
            c[q].cnt = mp_up(s.cnt);  c[q].cur = mp_up(s.cur);  c[q].key = mp_up(s.key);  c[q].tmp = mp_up(s.tmp);  c[q].pos = mp_up(s.pos);
            c[q].gkey = mp_up(T.t.key);
            c[q].n0 = mp_u(m.n0);  c[q].nn = mp_u(m.nn);  c[q].e0 = mp_u(m.e0);  c[q].ne = mp_u(m.ne);  c[q].L = mp_u(m.L);
            const int fw_e = mp_u(A.first_wave[e]), nw_e = mp_u(A.n_waves[e]);
            c[q].lg = (w - fw_e) * 64 + lane;  c[q].stride = 64 * nw_e;  c[q].first = w == fw_e;
            c[q].role = mp_u(T.t.role);
            c[q].item_base = mp_u((int)T.t.item_base);  c[q].seg_base = mp_u((int)T.t.seg_base);  c[q].n_real = mp_u((int)T.t.n_real);
            c[q].base = mp_u((int)T.t.item_base + m.e0 + (m.loops ? m.n0 : 0));      // global position of the molecule's first item
            bool ok = mp_fits(T, m);
            c[q].pkey = c[q].key;  c[q].ppos = c[q].pos;  c[q].pitem_base = 0;
            if (c[q].role != FN_ROLE_PLAIN) {
                const MpTask& P = A.t[mp_u(T.t.partner)];
                const MpSlice p = mp_slice(tiles, P);
                c[q].pkey = mp_up(p.key);  c[q].ppos = mp_up(p.pos);  c[q].pitem_base = mp_u((int)P.t.item_base);
                ok = ok && mp_fits(P, m);
            }
            c[q].ok = mp_u(ok ? 1 : 0) != 0;
            if (!c[q].ok && c[q].first && lane == 0) atomicOr(A.status, 4);
        }
    }
    stamp();
    // ---- keys of my tasks -> LDS: two items per thread and task in flight, unconditional clamped loads
    {
        int64_t kk[kMpMine][2];
#pragma unroll
        for (int q = 0; q < kMpMine; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                kk[q][u] = 0;
                if (q < n_my && c[q].ok && c[q].n_real > 0) {      // wave-uniform
                    int idx = c[q].e0 + c[q].lg + u * c[q].stride;
                    idx = idx < c[q].n_real ? idx : c[q].n_real - 1;
                    kk[q][u] = c[q].gkey[idx];
                }
            }
#pragma unroll
        for (int q = 0; q < kMpMine; ++q) {
            if (q >= n_my || !c[q].ok) continue;
            for (int k = c[q].lg; k <= c[q].nn; k += c[q].stride) { c[q].cnt[k] = 0;  c[q].cur[k] = 0; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int li = c[q].lg + u * c[q].stride;
                if (li >= c[q].L) continue;
                int64_t k = li < c[q].ne ? kk[q][u] - c[q].n0 : li - c[q].ne;
                if (k < 0 || k >= c[q].nn) { atomicOr(A.status, 2);  k = 0; }
                c[q].key[li] = (uint16_t)k;
            }
            for (int li = c[q].lg + 2 * c[q].stride; li < c[q].L; li += c[q].stride) {      // a big molecule
                int64_t k = li < c[q].ne ? c[q].gkey[c[q].e0 + li] - c[q].n0 : li - c[q].ne;
                if (k < 0 || k >= c[q].nn) { atomicOr(A.status, 2);  k = 0; }
                c[q].key[li] = (uint16_t)k;
            }
        }
    }
    __syncthreads();
    stamp();
    // ---- counts per node
#pragma unroll
    for (int q = 0; q < kMpMine; ++q)
        if (q < n_my && c[q].ok)
            for (int li = c[q].lg; li < c[q].L; li += c[q].stride) atomicAdd(&c[q].cnt[c[q].key[li]], 1);
    __syncthreads();
    stamp();
    // ---- exclusive scan over the molecule's nodes by the group's first wave; their rowptr
#pragma unroll
    for (int q = 0; q < kMpMine; ++q) {
        if (q >= n_my || !c[q].ok || !c[q].first) continue;
        int carry = 0;
        for (int k0 = 0; k0 < c[q].nn; k0 += 64) {
            const int k = k0 + lane;
            const int v = k < c[q].nn ? c[q].cnt[k] : 0;
            const int x = mp_wave_scan(v);
            if (k < c[q].nn) {
                c[q].cnt[k] = carry + x - v;
                A.rowptr[c[q].seg_base + c[q].n0 + k] = c[q].base + carry + x - v;
            }
            carry += __shfl(x, 63);
        }
        if (lane == 0) c[q].cnt[c[q].nn] = carry;
    }
    __syncthreads();
    stamp();
    // ---- unordered fill
#pragma unroll
    for (int q = 0; q < kMpMine; ++q)
        if (q < n_my && c[q].ok)
            for (int li = c[q].lg; li < c[q].L; li += c[q].stride) {
                const int k = c[q].key[li];
                c[q].tmp[c[q].cnt[k] + atomicAdd(&c[q].cur[k], 1)] = (uint16_t)li;
            }
    __syncthreads();
    stamp();
    // ---- rank inside the segment = final position; perm, inverse permutation
#pragma unroll
    for (int q = 0; q < kMpMine; ++q)
        if (q < n_my && c[q].ok)
            for (int li = c[q].lg; li < c[q].L; li += c[q].stride) {
                const int k = c[q].key[li];
                const int b = c[q].cnt[k], e = c[q].cnt[k + 1];
                int rank = 0;
#pragma unroll 4
                for (int t = b; t < e; ++t) rank += c[q].tmp[t] < li ? 1 : 0;
                const int pos = b + rank;
                c[q].pos[li] = (uint16_t)pos;
                const int v = li < c[q].ne ? c[q].e0 + li : c[q].n_real + c[q].n0 + (li - c[q].ne);
                A.perm[c[q].base + pos] = v;
                if (c[q].role == FN_ROLE_DST) A.aux_b[c[q].item_base + v] = c[q].base + pos - c[q].item_base;
            }
    __syncthreads();
    stamp();
    // ---- the two orders of a graph: an item's other endpoint is its key in the partner task (in LDS: no second pass over the
    // index tensors); the by-source task also writes where each item sits in the by-destination order (aux_b) and back (aux_c)
#pragma unroll
    for (int q = 0; q < kMpMine; ++q) {
        if (q >= n_my || !c[q].ok || c[q].role == FN_ROLE_PLAIN) continue;
        const int lbase = c[q].base - c[q].item_base;              // the molecule's first position, task-local (both orders)
        for (int li = c[q].lg; li < c[q].L; li += c[q].stride) {
            const int mine = lbase + c[q].pos[li];
            A.aux_a[c[q].item_base + mine] = c[q].n0 + c[q].pkey[li];
            if (c[q].role == FN_ROLE_SRC) {
                const int pd = lbase + c[q].ppos[li];
                A.aux_b[c[q].item_base + mine] = pd;
                A.aux_c[c[q].pitem_base + pd] = mine;
            }
        }
    }
    stamp();
    if (sp && tid == 0) sp[15] = wall_clock64();
}

// the padding tail of a task in closed form; r = residue of a reserved node (node hi - r), count(r) = items pointing at it.
// 32-bit arithmetic (every position is below 2^31: fn_plan_layout checks), no 64-bit divisions
struct MpPad {
    int Rn, Nn, Ri, Ni, Mn, q, rem, P0, loops;
    __device__ int count(int r) const { return q + (r < rem ? 1 : 0); }
    __device__ int edges_before(int k) const {                     // padding edges of the reserved nodes below node k
        if (k <= Nn - Mn) return 0;
        const int r = Nn - 1 - k;                                  // nodes below k have residues > r
        return (Mn - 1 - r) * q + (rem - 1 - r > 0 ? rem - 1 - r : 0);
    }
    __device__ int node_pos(int k) const { return P0 + loops * (k - Rn) + edges_before(k); }      // k in [Rn, Nn]
};
__device__ __forceinline__ MpPad mp_pad(const MpArgs& A, const MpTask& T, const int* real) {
    MpPad p;
    p.Rn = real[T.node_space];  p.Ri = real[T.item_space];
    p.Nn = (int)A.cap[T.node_space];  p.Ni = (int)A.cap[T.item_space];  p.Mn = (int)A.pad_mod[T.node_space];
    p.loops = T.t.n_loops > 0 ? 1 : 0;
    const int n_pad = p.Ni - p.Ri;
    p.q = n_pad / p.Mn;  p.rem = n_pad % p.Mn;
    p.P0 = (int)T.t.item_base + p.Ri + (p.loops ? p.Rn : 0);
    return p;
}

// The argument block is 2.6 KB.  Scalar loads from the kernel-argument segment cost a memory round trip per cache line the
// first time a CU touches it, and the phases below touch all of it in dependent steps (measured: 9 us before the first key
// was loaded): every workgroup copies the block into LDS with one parallel vector load and works from there.
constexpr int kMpArgWords = (sizeof(MpArgs) + 7) / 8 * 2;
// The four leading scalars repeat fields of the block: they arrive in preloaded registers, so the molecule's extents and the
// real-molecule count are requested in the same round trip as the copy of the block, not behind it.
__global__ __launch_bounds__(kMpThreads) void k_plan_mol(const int32_t* off, int n_mols, int n_spaces, const int32_t* counts_dev,
                                                         const MpArgs A_) {
    extern __shared__ int32_t lds_all[];
    int32_t* lds = lds_all + kMpArgWords;
    const int n_real = counts_dev ? *counts_dev : n_mols;
    if ((int)blockIdx.x < n_mols && threadIdx.x < 2 * FN_MAX_SPACES)           // mp_molecule's `ext`
        lds[threadIdx.x] = (int)(threadIdx.x >> 1) < n_spaces ? off[(size_t)(threadIdx.x >> 1) * (n_mols + 1) + blockIdx.x + (threadIdx.x & 1)] : 0;
    {
        const int32_t* src = reinterpret_cast<const int32_t*>(&A_);
        for (int i = threadIdx.x; i < (int)(sizeof(MpArgs) / 4); i += kMpThreads) lds_all[i] = src[i];
    }
    __syncthreads();
    const MpArgs& A = *reinterpret_cast<const MpArgs*>(lds_all);
    if ((int)blockIdx.x < A.n_mols) {
        if ((int)blockIdx.x < n_real) mp_molecule(A, (int)blockIdx.x, lds);
        return;
    }
    // ---- padding tail and the arrays' end entries
    const int tid = ((int)blockIdx.x - A.n_mols) * kMpThreads + threadIdx.x, span = A.pad_blocks * kMpThreads;
    int real[FN_MAX_SPACES];                                       // real extent of every index space: one round trip
#pragma unroll
    for (int sidx = 0; sidx < FN_MAX_SPACES; ++sidx) real[sidx] = sidx < A.n_spaces ? A.off[(size_t)sidx * (A.n_mols + 1) + n_real] : 0;
    if (tid < A.n_tasks) {                                         // end of every task's rowptr slice (= the next task's start)
        const fn_csr_task& t = A.t[tid].t;
        A.rowptr[t.seg_base + t.n_seg] = (int32_t)(t.item_base + t.n_real + t.n_loops);
    }
#pragma unroll
    for (int ti = 0; ti < FN_MAX_TASKS; ++ti) {
        if (ti >= A.n_tasks) break;
        const MpTask& T = A.t[ti];
        const MpPad p = mp_pad(A, T, real);
        const bool pair = T.t.role != FN_ROLE_PLAIN, dst = T.t.role == FN_ROLE_DST;
        const int ib = (int)T.t.item_base, pib = pair ? (int)A.t[T.t.partner].t.item_base : 0;
        const int nreal_items = (int)T.t.n_real, sb = (int)T.t.seg_base;
        // nodes: rowptr, and the loop item
        for (int k = p.Rn + tid; k < p.Nn; k += span) {
            const int pos0 = p.node_pos(k);
            A.rowptr[sb + k] = pos0;
            if (p.loops) {
                const int pos = pos0 + (k >= p.Nn - p.Mn ? p.count(p.Nn - 1 - k) : 0), v = nreal_items + k;
                A.perm[pos] = v;
                if (pair) {
                    A.aux_a[pos] = k;
                    if (dst) A.aux_b[ib + v] = pos - ib;
                    else { A.aux_b[pos] = pos - ib;  A.aux_c[pib + (pos - ib)] = pos - ib; }
                }
            }
        }
        // padding edges: item c points at node hi - (c - Ri) % Mn, the j-th of that node
        for (int c = p.Ri + tid; c < p.Ni; c += span) {
            const int r = (c - p.Ri) % p.Mn, j = (c - p.Ri) / p.Mn, k = p.Nn - 1 - r;
            const int pos = p.node_pos(k) + j;
            A.perm[pos] = c;
            if (pair) {
                A.aux_a[pos] = k;
                if (dst) A.aux_b[ib + c] = pos - ib;
                else { A.aux_b[pos] = pos - ib;  A.aux_c[pib + (pos - ib)] = pos - ib; }
            }
        }
    }
}

__global__ void k_zero_word(int32_t* p) { *p = 0; }

}  // namespace

extern "C" int fn_plan_build_mol(const fn_csr_task* tasks, int n_tasks, const fn_mol_layout* lay, int32_t* rowptr_all,
                                 int32_t* perm_all, int32_t* aux_a, int32_t* aux_b, int32_t* aux_c, int32_t* ws_i32, int32_t flags,
                                 fn_stream_t stream) {
    if (!tasks || n_tasks < 1 || !lay || !lay->offsets || !rowptr_all || !perm_all || !aux_a || !aux_b || !aux_c || !ws_i32)
        return fail(FN_EINVAL, "fn_plan_build_mol: null argument");
    if (n_tasks > FN_MAX_TASKS) return fail(FN_ETOOMANY, "fn_plan_build_mol: more than FN_MAX_TASKS tasks");
    if (lay->n_spaces < 1 || lay->n_spaces > FN_MAX_SPACES || lay->n_mols < 1 || lay->n_mols > (1 << 20))
        return fail(FN_EINVAL, "fn_plan_build_mol: bad layout");
    MpArgs A{};
    A.n_tasks = n_tasks;
    int64_t items = 0, segs = 0, est[FN_MAX_TASKS];
    int words = 0;
    for (int i = 0; i < n_tasks; ++i) {
        const fn_csr_task& t = tasks[i];
        if (t.item_base != items || t.seg_base != segs) return fail(FN_EINVAL, "fn_plan_build_mol: run fn_plan_layout first");
        if (t.n_real > 0 && !t.key) return fail(FN_EINVAL, "fn_plan_build_mol: null key");
        if (t.role != FN_ROLE_PLAIN) {
            if (t.n_real > 0 && !t.other_key) return fail(FN_EINVAL, "fn_plan_build_mol: null other_key");
            if (t.partner < 0 || t.partner >= n_tasks || tasks[t.partner].partner != i || tasks[t.partner].n_real != t.n_real ||
                tasks[t.partner].n_loops != t.n_loops || tasks[t.partner].n_seg != t.n_seg)
                return fail(FN_EINVAL, "fn_plan_build_mol: by-destination / by-source tasks must be paired");
        }
        const int ns = lay->node_space[i], is = lay->item_space[i];
        if (ns < 0 || ns >= lay->n_spaces || is < 0 || is >= lay->n_spaces) return fail(FN_EINVAL, "fn_plan_build_mol: bad index space");
        if (lay->cap[ns] != t.n_seg || lay->cap[is] != t.n_real || (t.n_loops != 0 && t.n_loops != t.n_seg))
            return fail(FN_EINVAL, "fn_plan_build_mol: task sizes disagree with the index spaces");
        if (lay->pad_mod[ns] < 1) return fail(FN_EINVAL, "fn_plan_build_mol: pad_mod must be positive");
        MpTask& T = A.t[i];
        T.t = t;  T.node_space = ns;  T.item_space = is;
        const int64_t ci = lay->max_per_mol[is] + (t.n_loops ? lay->max_per_mol[ns] : 0), cn = lay->max_per_mol[ns];
        if (ci < 0 || cn < 0 || ci > 65535 || cn > 65535) return fail(FN_EUNSUPPORTED, "fn_plan_build_mol: a molecule exceeds 65535 items");
        T.cap_items = (int)std::max<int64_t>(ci, 1);  T.cap_nodes = (int)std::max<int64_t>(cn, 1);
        T.lds_off = words;
        words += mp_slice_words(T.cap_items, T.cap_nodes);
        est[i] = ci;
        items += t.n_real + t.n_loops;
        segs += t.n_seg;
    }
    words += 2 * FN_MAX_SPACES + kMpArgWords;                      // the molecule's extents, the argument block
    if ((size_t)words * 4 > 64 * 1024) return fail(FN_EUNSUPPORTED, "fn_plan_build_mol: the largest molecule does not fit the 64 KB LDS tile");
    // schedule: tasks by decreasing size; a big one takes four waves of the molecule's workgroup, a middle one two, the rest
    // one -- always the least loaded aligned group.  A.t is re-ordered into schedule order (partner indices follow).
    int order[FN_MAX_TASKS], where[FN_MAX_TASKS];
    for (int i = 0; i < n_tasks; ++i) order[i] = i;
    std::stable_sort(order, order + n_tasks, [&](int a, int b) { return est[a] > est[b]; });
    MpTask sorted[FN_MAX_TASKS];
    int64_t load[kMpWaves] = {};
    int taken[kMpWaves] = {};
    for (int e = 0; e < n_tasks; ++e) {
        const int i = order[e];
        where[i] = e;
        sorted[e] = A.t[i];
        int nw = est[i] > 160 ? 4 : est[i] > 64 ? 2 : 1;
        int best = -1;
        int64_t best_load = -1;
        for (; best < 0; nw >>= 1) {                               // (a wave takes at most kMpMine tasks: narrower groups if need be)
            for (int f = 0; f + nw <= kMpWaves; f += nw) {
                int64_t l = 0;
                bool room = true;
                for (int q = 0; q < nw; ++q) { l = std::max(l, load[f + q]);  room = room && taken[f + q] < kMpMine; }
                if (room && (best < 0 || l < best_load)) { best_load = l;  best = f; }
            }
            if (best >= 0 || nw == 1) break;
        }
        if (best < 0) return fail(FN_EUNSUPPORTED, "fn_plan_build_mol: too many tasks for the molecule workgroup's schedule");
        A.first_wave[e] = best;  A.n_waves[e] = nw;
        for (int q = 0; q < nw; ++q) { load[best + q] = best_load + (est[i] + nw - 1) / nw + 8;  ++taken[best + q]; }
    }
    for (int e = 0; e < n_tasks; ++e) {
        A.t[e] = sorted[e];
        if (A.t[e].t.partner >= 0) A.t[e].t.partner = where[A.t[e].t.partner];
    }
    A.n_spaces = lay->n_spaces;
    A.off = lay->offsets;  A.n_mols = (int)lay->n_mols;  A.counts_dev = lay->counts_dev;
    for (int s = 0; s < FN_MAX_SPACES; ++s) { A.cap[s] = s < lay->n_spaces ? lay->cap[s] : 0;  A.pad_mod[s] = s < lay->n_spaces ? lay->pad_mod[s] : 1; }
    A.rowptr = rowptr_all;  A.perm = perm_all;  A.aux_a = aux_a;  A.aux_b = aux_b;  A.aux_c = aux_c;
    A.status = ws_i32 + segs + items;
    // padding workgroups: the tail is a few per cent of the items (none for an unpadded batch: one block writes the end entries)
    int64_t pad_work = 0;
    for (int s = 0; s < lay->n_spaces; ++s) pad_work = std::max(pad_work, lay->pad_hint[s]);
    A.pad_blocks = (int)std::min<int64_t>(std::max<int64_t>((pad_work + kMpThreads - 1) / kMpThreads, 1), 64);
    int64_t n_stamps = 0;
    unsigned long long* sbuf = fni::stamps(&n_stamps);
    A.stamps = n_stamps >= (int64_t)A.n_mols * 16 ? sbuf : nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!(flags & FN_PLAN_PREZEROED)) hipLaunchKernelGGL(k_zero_word, dim3(1), dim3(1), 0, st, A.status);
    hipLaunchKernelGGL(k_plan_mol, dim3((unsigned)(A.n_mols + A.pad_blocks)), dim3(kMpThreads), (size_t)words * 4, st, A.off, A.n_mols,
                       A.n_spaces, A.counts_dev, A);
    return launch_status("fn_plan_build_mol");
}
