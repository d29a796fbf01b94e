// Graph plan of a molecule-contiguous batch in ONE launch (fn_plan_build_mol).
//
// fn_plan_build sorts every CSR of the batch with four grid-wide passes (histogram -> look-back scan -> unordered fill -> rank
// sort: four dependent launches, ~40 us at ESOL batch 512 whatever the size, because each is a latency-floor kernel).  collate_fn
// concatenates molecules (reference dataset/data.py:877-948), so every key of molecule i is smaller than every key of molecule
// i + 1 and the stable sort of a whole CSR is the concatenation of the stable sorts of its molecules: with the per-molecule
// offsets of the seven index spaces at hand (the collate has them: they ARE its cumulative counts) a workgroup can sort its own
// molecule in LDS and write the finished slices -- no global histogram, no scan, no atomics on global memory.
//
//   * molecule workgroups (one per molecule, 8 waves): each wave takes whole CSR tasks (by-destination and by-source order of a
//     graph are two tasks) and runs, wave-private in its LDS slice: counts per node (LDS atomics) -> exclusive scan -> unordered
//     fill -> rank of every item among its segment's ids (ascending original id = the reference's sequential scatter order; the
//     same rule as k_plan_ranksort) -> perm / other endpoint / inverse permutation written to their global slices.  After one
//     workgroup barrier the by-source tasks add the cross references (position in the destination order and back).
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
    int order[FN_MAX_TASKS];        // tasks by decreasing size, dealt to the 8 waves in snake order
    int n_tasks;
    const int32_t* off;             // [n_spaces][n_mols + 1]
    int n_mols;
    const int32_t* counts_dev;      // nullable: number of real molecules
    int64_t cap[FN_MAX_SPACES], pad_mod[FN_MAX_SPACES];
    int32_t *rowptr, *perm, *aux_a, *aux_b, *aux_c, *status;
    int pad_blocks;
    int64_t total_segs;
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
__device__ __forceinline__ MpMol mp_mol(const MpArgs& A, const MpTask& T, int mol) {
    const int32_t* on = A.off + (size_t)T.node_space * (A.n_mols + 1);
    const int32_t* oi = A.off + (size_t)T.item_space * (A.n_mols + 1);
    MpMol m;
    m.n0 = on[mol];  m.nn = on[mol + 1] - m.n0;
    m.e0 = oi[mol];  m.ne = oi[mol + 1] - m.e0;
    m.loops = T.t.n_loops > 0 ? 1 : 0;
    m.L = m.ne + (m.loops ? m.nn : 0);
    return m;
}

// one CSR task of one molecule, by one wave
__device__ void mp_task(const MpArgs& A, const MpTask& T, int mol, int32_t* lds) {
    const int lane = threadIdx.x & 63;
    const MpMol m = mp_mol(A, T, mol);
    if (m.L == 0 && m.nn == 0) return;
    if (m.L > T.cap_items || m.nn > T.cap_nodes) { if (lane == 0) atomicOr(A.status, 4);  return; }
    const MpSlice s = mp_slice(lds, T);
    const int64_t base = T.t.item_base + m.e0 + (m.loops ? m.n0 : 0);          // global position of the molecule's first item
    for (int k = lane; k <= m.nn; k += 64) { s.cnt[k] = 0;  s.cur[k] = 0; }
    __builtin_amdgcn_wave_barrier();
    // the molecule's keys -> LDS, eight loads per lane in flight (one dependent round trip for up to 512 items, not one per 64)
    for (int l0 = lane; l0 < m.L; l0 += 64 * 8) {
        int64_t kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int li = l0 + 64 * u;
            kk[u] = li < m.ne ? T.t.key[m.e0 + li] - m.n0 : li - m.ne;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int li = l0 + 64 * u;
            if (li >= m.L) continue;
            if (kk[u] < 0 || kk[u] >= m.nn) { atomicOr(A.status, 2);  kk[u] = 0; }
            s.key[li] = (uint16_t)kk[u];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // counts per node
    for (int li = lane; li < m.L; li += 64) atomicAdd(&s.cnt[s.key[li]], 1);
    __builtin_amdgcn_wave_barrier();
    // exclusive scan over the molecule's nodes (chunks of 64, running carry); rowptr of the molecule's nodes
    int carry = 0;
    for (int k0 = 0; k0 < m.nn; k0 += 64) {
        const int k = k0 + lane;
        const int c = k < m.nn ? s.cnt[k] : 0;
        int x = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (k < m.nn) {
            s.cnt[k] = carry + x - c;
            A.rowptr[T.t.seg_base + m.n0 + k] = (int32_t)(base + carry + x - c);
        }
        carry += __shfl(x, 63);
    }
    if (lane == 0) s.cnt[m.nn] = carry;
    __builtin_amdgcn_wave_barrier();
    // unordered fill
    for (int li = lane; li < m.L; li += 64) {
        const int k = s.key[li];
        s.tmp[s.cnt[k] + atomicAdd(&s.cur[k], 1)] = (uint16_t)li;
    }
    __builtin_amdgcn_wave_barrier();
    // rank inside the segment = final position; outputs
    for (int li = lane; li < m.L; li += 64) {
        const int k = s.key[li];
        const int b = s.cnt[k], e = s.cnt[k + 1];
        int rank = 0;
#pragma unroll 4
        for (int q = b; q < e; ++q) rank += s.tmp[q] < li ? 1 : 0;
        const int pos = b + rank;
        s.pos[li] = (uint16_t)pos;
        const int64_t v = li < m.ne ? m.e0 + li : T.t.n_real + m.n0 + (li - m.ne);
        A.perm[base + pos] = (int32_t)v;
        if (T.t.role == FN_ROLE_DST) A.aux_b[T.t.item_base + v] = (int32_t)(base + pos - T.t.item_base);
    }
}

// the two orders of a graph after the barrier: an item's other endpoint is its key in the partner task (already in LDS: no
// second pass over the index tensors); the by-source task also writes where each item sits in the by-destination order
// (aux_b) and back (aux_c)
__device__ void mp_cross(const MpArgs& A, const MpTask& T, int mol, int32_t* lds) {
    const int lane = threadIdx.x & 63;
    const MpTask& P = A.t[T.t.partner];
    const MpMol m = mp_mol(A, T, mol);
    if (m.L == 0 || m.L > T.cap_items || m.nn > T.cap_nodes || m.L > P.cap_items || m.nn > P.cap_nodes) return;
    const MpSlice s = mp_slice(lds, T), p = mp_slice(lds, P);
    const int64_t lbase = m.e0 + (m.loops ? m.n0 : 0);             // the molecule's first position, task-local (both orders)
    const bool src = T.t.role == FN_ROLE_SRC;
    for (int li = lane; li < m.L; li += 64) {
        const int64_t mine = lbase + s.pos[li];
        A.aux_a[T.t.item_base + mine] = m.n0 + p.key[li];
        if (src) {
            const int64_t pd = lbase + p.pos[li];
            A.aux_b[T.t.item_base + mine] = (int32_t)pd;
            A.aux_c[P.t.item_base + pd] = (int32_t)mine;
        }
    }
}

// the padding tail of a task in closed form; r = residue of a reserved node (node hi - r), count(r) = items pointing at it
struct MpPad {
    int64_t Rn, Nn, Ri, Ni, Mn, q, rem, P0;
    int loops;
    __device__ int64_t count(int64_t r) const { return q + (r < rem ? 1 : 0); }
    __device__ int64_t edges_before(int64_t k) const {             // padding edges of the reserved nodes below node k
        if (k <= Nn - Mn) return 0;
        const int64_t r = Nn - 1 - k;                              // nodes above k have residues < r ... below k: residues > r
        return (Mn - 1 - r) * q + (rem - 1 - r > 0 ? rem - 1 - r : 0);
    }
    __device__ int64_t node_pos(int64_t k) const { return P0 + loops * (k - Rn) + edges_before(k); }     // k in [Rn, Nn]
};
__device__ __forceinline__ MpPad mp_pad(const MpArgs& A, const MpTask& T, int n_real_mols) {
    MpPad p;
    p.Rn = A.off[(size_t)T.node_space * (A.n_mols + 1) + n_real_mols];
    p.Ri = A.off[(size_t)T.item_space * (A.n_mols + 1) + n_real_mols];
    p.Nn = A.cap[T.node_space];  p.Ni = A.cap[T.item_space];  p.Mn = A.pad_mod[T.node_space];
    p.loops = T.t.n_loops > 0 ? 1 : 0;
    const int64_t n_pad = p.Ni - p.Ri;
    p.q = n_pad / p.Mn;  p.rem = n_pad % p.Mn;
    p.P0 = T.t.item_base + p.Ri + (p.loops ? p.Rn : 0);
    return p;
}

__global__ __launch_bounds__(kMpThreads) void k_plan_mol(MpArgs A) {
    extern __shared__ int32_t lds[];
    const int n_real = A.counts_dev ? *A.counts_dev : A.n_mols;
    if ((int)blockIdx.x < A.n_mols) {
        const int mol = (int)blockIdx.x, w = threadIdx.x >> 6;
        if (mol >= n_real) return;
        // tasks by decreasing size, dealt out in snake order: wave w takes the w-th largest, then the (15 - w)-th, ...
        auto mine = [&](int i) { const int r = i % (2 * kMpWaves);  return (r < kMpWaves ? r : 2 * kMpWaves - 1 - r) == w; };
        for (int i = 0; i < A.n_tasks; ++i)
            if (mine(i)) mp_task(A, A.t[A.order[i]], mol, lds);
        __syncthreads();
        for (int i = 0; i < A.n_tasks; ++i) {
            const MpTask& T = A.t[A.order[i]];
            if (mine(i) && T.t.role != FN_ROLE_PLAIN) mp_cross(A, T, mol, lds);
        }
        return;
    }
    // ---- padding tail and the arrays' end entries
    const int64_t tid = (int64_t)((int)blockIdx.x - A.n_mols) * kMpThreads + threadIdx.x, span = (int64_t)A.pad_blocks * kMpThreads;
    if (tid < A.n_tasks) {                                         // end of every task's rowptr slice (= the next task's start)
        const fn_csr_task& t = A.t[tid].t;
        A.rowptr[t.seg_base + t.n_seg] = (int32_t)(t.item_base + t.n_real + t.n_loops);
    }
    for (int ti = 0; ti < A.n_tasks; ++ti) {
        const MpTask& T = A.t[ti];
        const MpPad p = mp_pad(A, T, n_real);
        const bool pair = T.t.role != FN_ROLE_PLAIN, dst = T.t.role == FN_ROLE_DST;
        const int64_t ib = T.t.item_base, pib = pair ? A.t[T.t.partner].t.item_base : 0;
        // nodes: rowptr, and the loop item
        for (int64_t k = p.Rn + tid; k < p.Nn; k += span) {
            const int64_t pos0 = p.node_pos(k);
            A.rowptr[T.t.seg_base + k] = (int32_t)pos0;
            if (p.loops) {
                const int64_t pos = pos0 + (k >= p.Nn - p.Mn ? p.count(p.Nn - 1 - k) : 0), v = T.t.n_real + k;
                A.perm[pos] = (int32_t)v;
                if (pair) {
                    A.aux_a[pos] = (int32_t)k;
                    if (dst) A.aux_b[ib + v] = (int32_t)(pos - ib);
                    else { A.aux_b[pos] = (int32_t)(pos - ib);  A.aux_c[pib + (pos - ib)] = (int32_t)(pos - ib); }
                }
            }
        }
        // padding edges: item c points at node hi - (c - Ri) % Mn, the j-th of that node
        for (int64_t c = p.Ri + tid; c < p.Ni; c += span) {
            const int64_t r = (c - p.Ri) % p.Mn, j = (c - p.Ri) / p.Mn, k = p.Nn - 1 - r;
            const int64_t pos = p.node_pos(k) + j;
            A.perm[pos] = (int32_t)c;
            if (pair) {
                A.aux_a[pos] = (int32_t)k;
                if (dst) A.aux_b[ib + c] = (int32_t)(pos - ib);
                else { A.aux_b[pos] = (int32_t)(pos - ib);  A.aux_c[pib + (pos - ib)] = (int32_t)(pos - ib); }
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
        A.order[i] = i;
        items += t.n_real + t.n_loops;
        segs += t.n_seg;
    }
    if ((size_t)words * 4 > 64 * 1024) return fail(FN_EUNSUPPORTED, "fn_plan_build_mol: the largest molecule does not fit the 64 KB LDS tile");
    std::stable_sort(A.order, A.order + n_tasks, [&](int a, int b) { return est[a] > est[b]; });
    A.off = lay->offsets;  A.n_mols = (int)lay->n_mols;  A.counts_dev = lay->counts_dev;
    for (int s = 0; s < FN_MAX_SPACES; ++s) { A.cap[s] = s < lay->n_spaces ? lay->cap[s] : 0;  A.pad_mod[s] = s < lay->n_spaces ? lay->pad_mod[s] : 1; }
    A.rowptr = rowptr_all;  A.perm = perm_all;  A.aux_a = aux_a;  A.aux_b = aux_b;  A.aux_c = aux_c;
    A.status = ws_i32 + segs + items;
    A.total_segs = segs;
    // padding workgroups: the tail is a few per cent of the items (none for an unpadded batch: one block writes the end entries)
    int64_t pad_work = 0;
    for (int s = 0; s < lay->n_spaces; ++s) pad_work = std::max(pad_work, lay->pad_hint[s]);
    A.pad_blocks = (int)std::min<int64_t>(std::max<int64_t>((pad_work + kMpThreads - 1) / kMpThreads, 1), 64);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!(flags & FN_PLAN_PREZEROED)) hipLaunchKernelGGL(k_zero_word, dim3(1), dim3(1), 0, st, A.status);
    hipLaunchKernelGGL(k_plan_mol, dim3((unsigned)(A.n_mols + A.pad_blocks)), dim3(kMpThreads), (size_t)words * 4, st, A);
    return launch_status("fn_plan_build_mol");
}
