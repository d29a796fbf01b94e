// batch_io.hip -- everything between a collated batch and the encoder's first launch: the graph plan of hand-built batches (four
// grid-wide integer passes; collated batches take mol_plan.hip), the bond-graph / fragment-bond-graph topology, the staging launch of
// the static-shape step and the one-launch collate of a resident store.  Integer / byte work; a translation unit of its own since
// round 5 (it shares nothing with the attention and matrix kernels but the error string).
#include <algorithm>

#include "fn_internal.h"

namespace {
using fni::fail;
using fni::launch_status;
using fni::tune;

__global__ void k_zero2_i32(int32_t* __restrict__ a, int64_t na, int32_t* __restrict__ b, int64_t nb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < na) a[i] = 0;
        else b[i - na] = 0;
    }
}

// =====================================================================================
// Graph plan
// =====================================================================================
struct PlanTasks {
    int n;
    int64_t total_items;
    int64_t total_segs;
    fn_csr_task t[FN_MAX_TASKS];
};

__device__ __forceinline__ int find_task(const PlanTasks& P, int64_t g) {
    int ti = 0;
    while (ti + 1 < P.n && g >= P.t[ti + 1].item_base) ++ti;
    return ti;
}

__device__ __forceinline__ int64_t item_key(const fn_csr_task& T, int64_t local) {
    return local < T.n_real ? T.key[local] : local - T.n_real;
}
__device__ __forceinline__ int64_t item_other(const fn_csr_task& T, int64_t local) {
    return local < T.n_real ? T.other_key[local] : local - T.n_real;
}

// Runs of equal keys in consecutive lanes (destination-sorted edge lists, batch / frag_batch vectors) are counted once
// per run: the first lane of a run issues one atomic for the whole run.  Unsorted keys degenerate to one per lane.
__device__ __forceinline__ int run_after(uint64_t heads, int lane) {      // lanes from `lane` to the next run head
    const uint64_t later = lane == 63 ? 0 : heads >> (lane + 1);
    return later ? __ffsll((unsigned long long)later) : 64 - lane;
}
__global__ void k_plan_hist(PlanTasks P, int32_t* __restrict__ rowptr_all, int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t span = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g0 = (int64_t)blockIdx.x * blockDim.x; g0 < P.total_items; g0 += span) {     // uniform trip count per wave
        const int64_t g = g0 + threadIdx.x;
        int64_t seg = -1 - lane;                                   // distinct negative values: never equal to a neighbour
        if (g < P.total_items) {
            const fn_csr_task& T = P.t[find_task(P, g)];
            const int64_t k = item_key(T, g - T.item_base);
            if (k < 0 || k >= T.n_seg) atomicOr(status, 1);
            else seg = T.seg_base + k;
        }
        const int64_t prev = __shfl_up(seg, 1);
        const bool head = lane == 0 || seg != prev;
        const uint64_t heads = __ballot(head);
        if (head && seg >= 0) atomicAdd(&rowptr_all[seg + 1], run_after(heads, lane));
    }
}

constexpr int kScanChunk = 2048;   // items per block in the multi-block scan (256 threads x 8)

// Single-pass inclusive scan (decoupled look-back): block b scans its 2048-item chunk, publishes its total as
// state[b] = (1 << 32 | total), sums its predecessors' words 64 at a time until it meets one that already carries an
// inclusive prefix (2 << 32 | prefix), publishes its own inclusive prefix and adds the exclusive one to its chunk.
// A block only ever waits for lower-numbered blocks, which the dispatcher started earlier.  `state` must be zero.
__global__ __launch_bounds__(256) void k_scan_lookback(int32_t* __restrict__ a, int64_t len, unsigned long long* __restrict__ state) {
    __shared__ int32_t wave_tot[4];
    __shared__ int32_t s_prefix;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, b = blockIdx.x;
    const int64_t i0 = (int64_t)b * kScanChunk + (int64_t)tid * 8;
    int32_t v[8];
    int32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sum += (i0 + k < len) ? a[i0 + k] : 0;
        v[k] = sum;
    }
    int32_t x = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int32_t y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wid] = x;
    __syncthreads();
    const int32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    if (wid == 0) {
        if (lane == 0)
            __hip_atomic_store(&state[b], ((unsigned long long)(b == 0 ? 2 : 1) << 32) | (uint32_t)total, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_AGENT);
        int32_t prefix = 0;
        for (int hi = b - 1; hi >= 0; hi -= 64) {                   // window of 64 predecessors: lane l looks at block hi - l
            const int j = hi - lane;
            unsigned long long w = 3ull << 32;                      // lanes below block 0: "nothing, stop"
            if (j >= 0) {
                do { w = __hip_atomic_load(&state[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while ((w >> 32) == 0);
            }
            const uint64_t done = __ballot((w >> 32) >= 2);         // inclusive prefix (or the start of the array) seen
            const int stop = done ? __ffsll((unsigned long long)done) - 1 : 63;
            int32_t part = (lane <= stop && j >= 0) ? (int32_t)(uint32_t)w : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            prefix += part;
            if (done) break;
        }
        if (lane == 0) {
            if (b > 0)
                __hip_atomic_store(&state[b], (2ull << 32) | (uint32_t)(prefix + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = prefix;
        }
    }
    __syncthreads();
    int32_t before = s_prefix;
#pragma unroll
    for (int w = 0; w < 4; ++w) before += (w < wid) ? wave_tot[w] : 0;
    const int32_t excl = before + (x - sum);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (i0 + k < len) a[i0 + k] = v[k] + excl;
}

// unordered fill (integer atomics): tmp[pos] = task-local item id, seg_of[pos] = global segment id
__global__ void k_plan_fill(PlanTasks P, const int32_t* __restrict__ rowptr_all, int32_t* __restrict__ cursor,
                            int32_t* __restrict__ tmp, int32_t* __restrict__ seg_of) {
    const int lane = threadIdx.x & 63;
    const int64_t span = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g0 = (int64_t)blockIdx.x * blockDim.x; g0 < P.total_items; g0 += span) {
        const int64_t g = g0 + threadIdx.x;
        int64_t seg = -1 - lane, local = 0;
        if (g < P.total_items) {
            const fn_csr_task& T = P.t[find_task(P, g)];
            local = g - T.item_base;
            const int64_t k = item_key(T, local);
            if (k >= 0 && k < T.n_seg) seg = T.seg_base + k;
        }
        const int64_t prev = __shfl_up(seg, 1);
        const bool head = lane == 0 || seg != prev;
        const uint64_t heads = __ballot(head);
        int32_t base = 0;
        if (head && seg >= 0) base = atomicAdd(&cursor[seg], run_after(heads, lane));     // one slot range per run
        const int head_lane = 63 - __clzll((long long)(heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1))));
        base = __shfl(base, head_lane);
        if (seg >= 0) {
            const int32_t pos = rowptr_all[seg] + base + (lane - head_lane);
            tmp[pos] = (int32_t)local;
            seg_of[pos] = (int32_t)seg;
        }
    }
}

// rank sort inside each segment: one thread per filled slot counts the smaller ids of its segment and
// writes its id to that rank => ascending item id = summation order of the reference's sequential scatter_add.
// Work is sum(len^2) independent cached loads (len <= ~30 for every molecular index space).
// The thread knows the item's final position, so it also writes what used to be separate passes over the finished
// permutation: the item's other endpoint (aux_a), for by-destination tasks the inverse permutation (aux_b), and -- round 3 --
// for by-source tasks where the edge sits in the DESTINATION order (aux_b) and back (aux_c): the item's rank in its
// destination segment is counted here from that segment's unordered fill, so the pass needs no finished by-destination
// permutation and k_plan_aux_src is gone (one launch of ~6 us less per plan).
__device__ __forceinline__ int find_task_by_seg(const PlanTasks& P, int64_t seg) {
    int ti = 0;
    while (ti + 1 < P.n && seg >= P.t[ti + 1].seg_base) ++ti;
    return ti;
}
__global__ void k_plan_ranksort(PlanTasks P, const int32_t* __restrict__ rowptr_all, const int32_t* __restrict__ tmp,
                                const int32_t* __restrict__ seg_of, int32_t* __restrict__ perm_all,
                                int32_t* __restrict__ aux_a, int32_t* __restrict__ aux_b, int32_t* __restrict__ aux_c) {
    const int32_t filled = rowptr_all[P.total_segs];        // < total_items only if some keys were out of range
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < filled;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int32_t seg = seg_of[g];
        const int32_t b = rowptr_all[seg], e = rowptr_all[seg + 1];
        const int32_t v = tmp[g];
        int32_t rank = 0;
        for (int32_t q = b; q < e; ++q) rank += (tmp[q] < v) ? 1 : 0;
        const int32_t pos = b + rank;
        perm_all[pos] = v;
        const fn_csr_task& T = P.t[find_task_by_seg(P, seg)];
        if (T.role != FN_ROLE_PLAIN && v >= 0 && v < T.n_real + T.n_loops) {
            aux_a[pos] = (int32_t)item_other(T, v);
            if (T.role == FN_ROLE_DST) aux_b[T.item_base + v] = (int32_t)(pos - T.item_base);   // inverse permutation
            else {
                // the same edge in the partner's (by-destination) order: its segment there is its other endpoint
                const fn_csr_task& D = P.t[T.partner];
                const int64_t kd = item_key(D, v);
                if (kd >= 0 && kd < D.n_seg) {
                    const int32_t bd = rowptr_all[D.seg_base + kd], ed = rowptr_all[D.seg_base + kd + 1];
                    int32_t rd = 0;
                    for (int32_t q = bd; q < ed; ++q) rd += (tmp[q] < v) ? 1 : 0;
                    const int32_t dpos = bd + rd - (int32_t)D.item_base;
                    aux_b[pos] = dpos;                                          // position of this edge in the DST order
                    aux_c[D.item_base + dpos] = (int32_t)(pos - T.item_base);   // and, for that DST position, its SRC position
                }
            }
        }
    }
}

// =====================================================================================
// Bond-graph topology on the GPU (reference fragnet/dataset/data.py:116-127, 157-182, 403-410):
// edge_index_bonds_graph = ordered pairs (i, j) of directed bonds of one molecule that share exactly one atom,
// i-major with j ascending, followed per molecule by the mutual pairs of its two-atom components ("one-bond
// fragments", lowest atom first).  Bond id = position in the batched edge_index.  One thread per bond walks its
// molecule's bonds (the reference's O(k^2) loop; k ~ 55 for ESOL): neighbouring threads read the same few hundred
// bytes, and j ascending falls out of the walk, so nothing is sorted.
// =====================================================================================
struct BondGraphArgs {
    const int64_t *src, *dst, *atom_mol;      // edge_index rows, molecule of every atom
    int64_t E, B;
    int32_t *cnt;        // [E+1] pass 1: cnt[1+i] = pairs of bond i; after the scan: cnt[i] = pairs before bond i
    int32_t *flag;       // [E]   1 if bond i is the a -> b (a < b) direction of a two-atom component
    int32_t *mol_first;  // [B+1] pass 1: bonds per molecule at [1+m]; after the scan: first bond of molecule m
    int32_t *mol_ob;     // [B+1] likewise for the flagged bonds
    int mode;            // 0: bond graph (data.py:116-127 + one-bond fragments); 1: fragment-bond graph (data.py:131-154: a molecule
                         // with exactly two connection nodes pairs those whose (begin, end) differ, no extras)
};

__global__ void k_bg_mol_hist(BondGraphArgs A) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.E; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&A.mol_first[1 + A.atom_mol[A.src[i]]], 1);
}

__global__ void k_bg_count(BondGraphArgs A) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.E; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = A.src[i], v = A.dst[i];
        const int64_t m = A.atom_mol[u];
        const int32_t j0 = A.mol_first[m], j1 = A.mol_first[m + 1];
        int c = 0, du = 0, dv = 0;
        const bool two = A.mode == 1 && j1 - j0 == 2;
        for (int32_t j = j0; j < j1; ++j) {
            const int64_t a = A.src[j], b = A.dst[j];
            const int common = (int)(a == u || a == v) + (int)(b != a && (b == u || b == v));
            c += two ? (a != u || b != v) : common == 1;                 // |set(b_i) & set(b_j)| == 1
            du += a == u;
            dv += a == v;
        }
        A.cnt[1 + i] = c;
        const int ob = (A.mode == 0 && u < v && du == 1 && dv == 1) ? 1 : 0;
        A.flag[i] = ob;
        if (ob) atomicAdd(&A.mol_ob[1 + m], 1);
    }
}

__global__ void k_bg_fill(BondGraphArgs A, int64_t* __restrict__ out, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.E; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = A.src[i], v = A.dst[i];
        const int64_t m = A.atom_mol[u];
        const int32_t j0 = A.mol_first[m], j1 = A.mol_first[m + 1];
        int64_t pos = (int64_t)A.cnt[i] + 2 * (int64_t)A.mol_ob[m];      // pairs of earlier bonds + extras of earlier molecules
        int rev = -1, rank = 0;
        const bool two = A.mode == 1 && j1 - j0 == 2;
        for (int32_t j = j0; j < j1; ++j) {
            const int64_t a = A.src[j], b = A.dst[j];
            const int common = (int)(a == u || a == v) + (int)(b != a && (b == u || b == v));
            if (two ? (a != u || b != v) : common == 1) {
                out[pos] = i;
                out[total + pos] = j;
                ++pos;
            }
            if (a == v && b == u) rev = j;
            rank += (A.flag[j] && a < u) ? 1 : 0;                        // two-atom components with a lower first atom
        }
        if (A.flag[i] && rev >= 0) {                                      // the molecule's extras follow ALL its pairs
            const int64_t p = (int64_t)A.cnt[j1] + 2 * (int64_t)A.mol_ob[m] + 2 * rank;
            out[p] = i;          out[total + p] = rev;
            out[p + 1] = rev;    out[total + p + 1] = i;
        }
    }
}

__global__ void k_bg_total(BondGraphArgs A, int64_t* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = (int64_t)A.cnt[A.E] + 2 * (int64_t)A.mol_ob[A.B];
}

// one block range per field, sized by the field (the grid used to be 1024 x fields: 25 k blocks for a batch whose ids, masks
// and counters need a handful -- 8.7 us of block scheduling at any batch size)
struct StageFields {
    fn_stage_field f[FN_MAX_STAGE_FIELDS];
    int first[FN_MAX_STAGE_FIELDS + 1];
    int n;
};
__global__ void k_stage_padded(StageFields F) {
    int fi = 0;
    while (fi + 1 < F.n && (int)blockIdx.x >= F.first[fi + 1]) ++fi;
    const fn_stage_field& f = F.f[fi];
    const int64_t stride = (int64_t)(F.first[fi + 1] - F.first[fi]) * blockDim.x;
    const int64_t t0 = (int64_t)((int)blockIdx.x - F.first[fi]) * blockDim.x + threadIdx.x;
    if (f.kind == FN_STAGE_ROWS) {
        const float* src = static_cast<const float*>(f.src);
        float* dst = static_cast<float*>(f.dst);
        const int64_t real = f.n_real * f.width, all = f.cap * f.width;
        if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
            // rows are contiguous: copy as one flat array, 16 bytes per thread; the group that straddles the end of the
            // real data and the tail of the buffer go element by element
            const int64_t all4 = all >> 2, real4 = real >> 2;
            for (int64_t i = t0; i < all4; i += stride)
                st4(dst + i * 4, i < real4 ? ld4(src + i * 4) : (i * 4 >= real ? make_float4(0.f, 0.f, 0.f, 0.f)
                    : make_float4(src[i * 4], i * 4 + 1 < real ? src[i * 4 + 1] : 0.f, i * 4 + 2 < real ? src[i * 4 + 2] : 0.f, 0.f)));
            for (int64_t i = (all4 << 2) + t0; i < all; i += stride) dst[i] = i < real ? src[i] : 0.f;
        } else {
            for (int64_t i = t0; i < all; i += stride) dst[i] = i < real ? src[i] : 0.f;
        }
    } else if (f.kind == FN_STAGE_MASK) {
        float* dst = static_cast<float*>(f.dst);
        for (int64_t i = t0; i < f.cap; i += stride) dst[i] = i < f.n_real ? 1.f : 0.f;
    } else if (f.kind == FN_STAGE_COUNT) {
        if (t0 == 0) *static_cast<int32_t*>(f.dst) = (int32_t)f.n_real;
    } else if (f.kind == FN_STAGE_BUMP) {
        if (t0 == 0) *static_cast<int64_t*>(f.dst) += f.n_real;
    } else if (f.kind == FN_STAGE_OFFSETS) {
        const int32_t* src = static_cast<const int32_t*>(f.src);
        int32_t* dst = static_cast<int32_t*>(f.dst);
        for (int64_t i = t0; i < f.width * (f.cap + 1); i += stride) {
            const int64_t s = i / (f.cap + 1), m = i % (f.cap + 1);
            dst[i] = src[s * (f.n_real + 1) + (m < f.n_real ? m : f.n_real)];
        }
    } else if (f.kind == FN_STAGE_ZERO) {
        int32_t* dst = static_cast<int32_t*>(f.dst);
        for (int64_t i = t0; i < f.cap; i += stride) dst[i] = 0;
    } else {
        const int64_t* src = static_cast<const int64_t*>(f.src);
        int64_t* dst = static_cast<int64_t*>(f.dst);
        const int rows = f.kind == FN_STAGE_COLS ? 2 : 1;
        for (int64_t i = t0; i < rows * f.cap; i += stride) {
            const int64_t r = i >= f.cap ? 1 : 0, c = i - r * f.cap;
            dst[i] = c < f.n_real ? src[r * f.n_real + c] : f.pad_hi - (c - f.n_real) % f.pad_mod;
        }
    }
}
// ---- a batch from a resident flat store in ONE launch (dataset.FlatMolStore.collate: the reference's collate_fn, dataset/data.py:877-948,
// on molecules that live concatenated in HBM).  Molecule b of the batch is store molecule idx[b]; in index space s its rows are the
// store rows start[s][b] .. and land at batch rows off[s][b] .. off[s][b+1] (the batch's offsets table, plan.CollatedBatch.offsets,
// computed on the host from the store's molecule lengths).  A field = one output tensor.
//
// Round 5: MOLECULE-OWNER mapping.  A half-wave owns one (field, molecule, chunk): the segment's extent is three loads (off[b],
// off[b + 1], start[b]) and everything after that is a straight copy -- 128 contiguous bytes per instruction, several in flight.
// Round 4's flat mapping (a thread = four words of the output) found the molecule of every group of four words by a binary search
// of the offsets row: thirteen dependent loads to move 16 bytes at 8192 molecules per batch, 2.1 ms of a 6.8-ms step from the store.
// Heavy fields first (the launch's tail is then made of the short ones); a molecule's segment is cut into chunks of kCollChunk
// words so that one 27-KB feature segment does not become the launch's critical path at small batches (max_seg_rows: the store's
// bound on a molecule's rows in the field's space; 0 = unknown: one chunk walks the whole segment).
constexpr int kCollChunk = 2048;        // 4-byte words per chunk (8 KB)
struct CollateFields {
    fn_collate_field f[FN_MAX_COLLATE_FIELDS];
    int64_t first[FN_MAX_COLLATE_FIELDS + 1];      // first half-wave item of every field
    int chunks[FN_MAX_COLLATE_FIELDS];
    int n;
};
__global__ void k_copy_words(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_collate_store(CollateFields F, const int64_t* __restrict__ starts, const int32_t* __restrict__ offs, int B) {
    const int64_t item = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int lane = threadIdx.x & 31;
    if (item >= F.first[F.n]) return;
    int fi = 0;
    while (fi + 1 < F.n && item >= F.first[fi + 1]) ++fi;
    const fn_collate_field& f = F.f[fi];
    const int nch = F.chunks[fi];
    const int64_t local = item - F.first[fi];
    const int b = (int)(local / nch), ch = (int)(local - (int64_t)b * nch);
    const int32_t* off = offs + (size_t)f.space * (B + 1);
    const int64_t r0 = off[b], nrow = (int64_t)off[b + 1] - r0, s0 = starts[(size_t)f.space * B + b];
    if (f.kind == FN_COLLATE_ROWS) {
        const int64_t w = f.width_words, words = nrow * w;
        const int64_t lo = (int64_t)ch * kCollChunk, hi = ch + 1 == nch ? words : (lo + kCollChunk < words ? lo + kCollChunk : words);
        const int32_t* src = static_cast<const int32_t*>(f.src) + s0 * w;
        int32_t* dst = static_cast<int32_t*>(f.dst) + r0 * w;
        int64_t i = lo + lane;
        for (; i + 96 < hi; i += 128) {                       // four 128-byte pieces in flight
            const int32_t a0 = src[i], a1 = src[i + 32], a2 = src[i + 64], a3 = src[i + 96];
            dst[i] = a0;  dst[i + 32] = a1;  dst[i + 64] = a2;  dst[i + 96] = a3;
        }
        for (; i < hi; i += 32) dst[i] = src[i];
    } else if (f.kind == FN_COLLATE_BATCH) {
        int64_t* dst = static_cast<int64_t*>(f.dst) + r0;
        const int64_t lo = (int64_t)ch * kCollChunk, hi = ch + 1 == nch ? nrow : (lo + kCollChunk < nrow ? lo + kCollChunk : nrow);
        for (int64_t r = lo + lane; r < hi; r += 32) dst[r] = b;
    } else {                                              // FN_COLLATE_IDS: [width, rows] int64 out of [width, src_rows], values rebased
        // a stored index counts from the molecule's own first row of the space it points into (the store keeps them molecule-local
        // or store-global: f.src_global says which): batch value = stored - (store-global ? first store row : 0) + first batch row
        const int64_t shift = (int64_t)offs[(size_t)f.rebase_space * (B + 1) + b] - (f.src_global ? starts[(size_t)f.rebase_space * B + b] : 0);
        const int64_t per = kCollChunk / 2;               // rows per chunk (8-byte elements)
        const int64_t lo = (int64_t)ch * per, hi = ch + 1 == nch ? nrow : (lo + per < nrow ? lo + per : nrow);
        for (int c = 0; c < f.width_words; ++c) {
            const int64_t* src = static_cast<const int64_t*>(f.src) + (int64_t)c * f.src_rows + s0;
            int64_t* dst = static_cast<int64_t*>(f.dst) + (int64_t)c * f.rows + r0;
            int64_t r = lo + lane;
            for (; r + 96 < hi; r += 128) {
                const int64_t a0 = src[r], a1 = src[r + 32], a2 = src[r + 64], a3 = src[r + 96];
                dst[r] = a0 + shift;  dst[r + 32] = a1 + shift;  dst[r + 64] = a2 + shift;  dst[r + 96] = a3 + shift;
            }
            for (; r < hi; r += 32) dst[r] = src[r] + shift;
        }
    }
}
}  // namespace

extern "C" {

int fn_plan_layout(fn_csr_task* tasks, int n_tasks, int64_t* total_items, int64_t* total_segs) {
    if (!tasks || n_tasks < 0 || !total_items || !total_segs) return fail(FN_EINVAL, "fn_plan_layout: null argument");
    if (n_tasks > FN_MAX_TASKS) return fail(FN_ETOOMANY, "fn_plan_layout: more than FN_MAX_TASKS tasks");
    int64_t items = 0, segs = 0;
    for (int i = 0; i < n_tasks; ++i) {
        fn_csr_task& t = tasks[i];
        if (t.n_real < 0 || t.n_loops < 0 || t.n_seg < 0) return fail(FN_EINVAL, "fn_plan_layout: negative size");
        if (t.role != FN_ROLE_PLAIN && (t.partner < 0 || t.partner >= n_tasks)) return fail(FN_EINVAL, "fn_plan_layout: bad partner");
        t.item_base = items;
        t.seg_base = segs;
        items += t.n_real + t.n_loops;
        segs += t.n_seg;
    }
    if (items >= (1ll << 31) - 1 || segs >= (1ll << 31) - 1) return fail(FN_EINVAL, "fn_plan_layout: plan exceeds int32 positions");
    *total_items = items;
    *total_segs = segs;
    return 0;
}

int fn_plan_build(const fn_csr_task* tasks, int n_tasks, int32_t* rowptr_all, int32_t* perm_all, int32_t* aux_a,
                  int32_t* aux_b, int32_t* aux_c, int32_t* ws_i32, int32_t flags, fn_stream_t stream) {
    if (!tasks || n_tasks < 1 || !rowptr_all || !perm_all || !aux_a || !aux_b || !aux_c || !ws_i32)
        return fail(FN_EINVAL, "fn_plan_build: null argument");
    if (n_tasks > FN_MAX_TASKS) return fail(FN_ETOOMANY, "fn_plan_build: more than FN_MAX_TASKS tasks");
    PlanTasks P;
    P.n = n_tasks;
    int64_t items = 0, segs = 0;
    bool any_pair = false;
    for (int i = 0; i < n_tasks; ++i) {
        P.t[i] = tasks[i];
        if (tasks[i].item_base != items || tasks[i].seg_base != segs) return fail(FN_EINVAL, "fn_plan_build: run fn_plan_layout first");
        if (tasks[i].n_real > 0 && !tasks[i].key) return fail(FN_EINVAL, "fn_plan_build: null key");
        if (tasks[i].role != FN_ROLE_PLAIN) {
            any_pair = true;
            if (tasks[i].n_real > 0 && !tasks[i].other_key) return fail(FN_EINVAL, "fn_plan_build: null other_key");
        }
        items += tasks[i].n_real + tasks[i].n_loops;
        segs += tasks[i].n_seg;
    }
    P.total_items = items;
    P.total_segs = segs;
    hipStream_t st = S(stream);
    int32_t* cursor = ws_i32;
    int32_t* tmp = ws_i32 + segs;          // unordered fill target (total_items)
    int32_t* status = ws_i32 + segs + items;
    // zeroing by kernel, not hipMemsetAsync: the call must be capturable in a hipGraph and replayable (memset nodes
    // were observed to fault on the second replay on ROCm 7.2), and it is one launch for both regions
    const int nb = (int)((segs + kScanChunk - 1) / kScanChunk);
    int32_t* state_i32 = ws_i32 + segs + items + 4;
    if ((uintptr_t)state_i32 & 7) ++state_i32;                      // 64-bit look-back words
    const int64_t zero_ws = (state_i32 - ws_i32) + 2 * (int64_t)nb;
    if (!(flags & FN_PLAN_PREZEROED))
        hipLaunchKernelGGL(k_zero2_i32, dim3(flat_grid(segs + 1 + zero_ws, kGridCap)), dim3(kBlock), 0, st, rowptr_all,
                           segs + 1, ws_i32, zero_ws);
    if (items > 0) {
        const int g = flat_grid(items, kGridCap);
        hipLaunchKernelGGL(k_plan_hist, dim3(g), dim3(kBlock), 0, st, P, rowptr_all, status);
        // inclusive scan of the histogram in one launch
        hipLaunchKernelGGL(k_scan_lookback, dim3(nb), dim3(256), 0, st, rowptr_all + 1, segs,
                           reinterpret_cast<unsigned long long*>(state_i32));
        int32_t* seg_of = ws_i32 + (segs + items + 4 + 2 * (segs / 2048 + 1) + 2);      // the last `items` words of FN_PLAN_WS: never zeroed
        hipLaunchKernelGGL(k_plan_fill, dim3(g), dim3(kBlock), 0, st, P, rowptr_all, cursor, tmp, seg_of);
        hipLaunchKernelGGL(k_plan_ranksort, dim3(g), dim3(kBlock), 0, st, P, rowptr_all, tmp, seg_of, perm_all, aux_a, aux_b, aux_c);
        (void)any_pair;
    }
    return launch_status("fn_plan_build");
}

int fn_stage_padded(const fn_stage_field* fields, int n_fields, fn_stream_t stream) {
    if (!fields || n_fields < 1 || n_fields > FN_MAX_STAGE_FIELDS) return fail(FN_EINVAL, "fn_stage_padded: bad field count");
    StageFields F;
    F.n = n_fields;
    int blocks = 0;
    // blocks of a field: one per 1024 work items (a work item = 16 bytes of an aligned row table), at least one, 1024 at the top
    auto take = [&](int i, int64_t work) {
        F.first[i] = blocks;
        blocks += (int)std::min<int64_t>(std::max<int64_t>((work + 4 * kBlock - 1) / (4 * kBlock), 1), 1024);
    };
    for (int i = 0; i < n_fields; ++i) {
        const fn_stage_field& f = fields[i];
        if (f.kind == FN_STAGE_BUMP) {
            if (!f.dst || ((uintptr_t)f.dst & 7)) return fail(FN_EINVAL, "fn_stage_padded: bad bump field");
            F.f[i] = f;
            take(i, 1);
            continue;
        }
        if (f.kind == FN_STAGE_ZERO) {
            if (f.cap < 0 || (f.cap > 0 && !f.dst)) return fail(FN_EINVAL, "fn_stage_padded: bad zero field");
            F.f[i] = f;
            take(i, f.cap);
            continue;
        }
        if (f.kind == FN_STAGE_OFFSETS) {
            if (f.n_real < 0 || f.cap < f.n_real || f.width < 1 || !f.dst || !f.src) return fail(FN_EINVAL, "fn_stage_padded: bad offsets field");
            F.f[i] = f;
            take(i, f.width * (f.cap + 1));
            continue;
        }
        if (f.kind == FN_STAGE_COUNT) {
            if (f.n_real < 0 || !f.dst) return fail(FN_EINVAL, "fn_stage_padded: bad count field");
            F.f[i] = f;
            take(i, 1);
            continue;
        }
        if (f.n_real < 0 || f.cap < f.n_real || f.width < 1 || f.kind < 0 || f.kind > FN_STAGE_MASK || (f.cap > 0 && !f.dst) ||
            (f.n_real > 0 && f.kind != FN_STAGE_MASK && !f.src) ||
            ((f.kind == FN_STAGE_IDS || f.kind == FN_STAGE_COLS) && f.cap > f.n_real && f.pad_mod < 1))
            return fail(FN_EINVAL, "fn_stage_padded: bad field");
        F.f[i] = f;
        take(i, f.cap * (f.kind == FN_STAGE_ROWS ? (f.width + 3) / 4 : f.kind == FN_STAGE_COLS ? 2 : 1));
    }
    F.first[n_fields] = blocks;
    hipLaunchKernelGGL(k_stage_padded, dim3(blocks), dim3(kBlock), 0, S(stream), F);
    return launch_status("fn_stage_padded");
}

int fn_collate_store(const fn_collate_field* fields, int n_fields, const int64_t* starts, const int32_t* offsets, int n_spaces, int64_t B,
                     const void* tables_host, fn_stream_t stream) {
    if (!fields || n_fields < 1 || n_fields > FN_MAX_COLLATE_FIELDS || !starts || !offsets || n_spaces < 1 || B < 1 || B > INT32_MAX - 1)
        return fail(FN_EINVAL, "fn_collate_store: bad argument (1 .. FN_MAX_COLLATE_FIELDS fields, B >= 1)");
    CollateFields F{};
    int order[FN_MAX_COLLATE_FIELDS];
    int64_t weight[FN_MAX_COLLATE_FIELDS];
    for (int i = 0; i < n_fields; ++i) {
        const fn_collate_field& f = fields[i];
        if (f.rows < 0 || f.width_words < 1 || f.space < 0 || f.space >= n_spaces || f.kind < FN_COLLATE_ROWS || f.kind > FN_COLLATE_IDS ||
            (f.rows > 0 && (!f.dst || (f.kind != FN_COLLATE_BATCH && !f.src))) || f.max_seg_rows < 0 ||
            (f.kind == FN_COLLATE_IDS && (f.rebase_space < 0 || f.rebase_space >= n_spaces || f.src_rows < 0)))
            return fail(FN_EINVAL, "fn_collate_store: bad field");
        order[i] = i;
        weight[i] = f.rows * (f.kind == FN_COLLATE_ROWS ? f.width_words : f.kind == FN_COLLATE_IDS ? 2 * f.width_words : 2);     // 4-byte words moved
    }
    std::stable_sort(order, order + n_fields, [&](int a, int b) { return weight[a] > weight[b]; });      // heavy fields first
    int64_t items = 0;
    int live = 0;
    for (int q = 0; q < n_fields; ++q) {
        const fn_collate_field& f = fields[order[q]];
        if (f.rows == 0) continue;
        // words (ROWS) / rows (BATCH) / 8-byte rows (IDS) of the largest molecule over the chunk size
        const int64_t unit = f.kind == FN_COLLATE_ROWS ? (int64_t)f.max_seg_rows * f.width_words : f.kind == FN_COLLATE_IDS ? 2 * (int64_t)f.max_seg_rows : f.max_seg_rows;
        F.f[live] = f;
        F.first[live] = items;
        F.chunks[live] = (int)std::max<int64_t>(1, (unit + kCollChunk - 1) / kCollChunk);
        items += B * F.chunks[live];
        ++live;
    }
    F.first[live] = items;
    F.n = live;
    const int64_t blocks = (items + 7) / 8;
    if (blocks > INT32_MAX) return fail(FN_EUNSUPPORTED, "fn_collate_store: batch too large for one launch");
    if (tables_host) {
        // the two small tables arrive in ONE pinned host buffer laid out [starts | offsets]; a kernel reads it over the bus and writes
        // `starts` (whose allocation continues into `offsets`): a copy-engine transfer in front of every batch cost ~20 us plus the
        // switch between the copy engine and the compute queue, twice per step
        const int64_t words = (int64_t)n_spaces * B * 2 + (int64_t)n_spaces * (B + 1);
        if (reinterpret_cast<const char*>(offsets) != reinterpret_cast<const char*>(starts) + (size_t)n_spaces * B * 8)
            return fail(FN_EINVAL, "fn_collate_store: with tables_host the offsets table must follow the starts table in one allocation");
        hipLaunchKernelGGL(k_copy_words, dim3(flat_grid(words, 64)), dim3(kBlock), 0, S(stream), static_cast<const int32_t*>(tables_host),
                           reinterpret_cast<int32_t*>(const_cast<int64_t*>(starts)), words);
        if (int rc = launch_status("fn_collate_store (tables)")) return rc;
    }
    if (live == 0) return 0;
    hipLaunchKernelGGL(k_collate_store, dim3((unsigned)blocks), dim3(kBlock), 0, S(stream), F, starts, offsets, (int)B);
    return launch_status("fn_collate_store");
}

}  // extern "C"

namespace {
struct BondGraphWs {
    BondGraphArgs A;
    unsigned long long *st_cnt, *st_mol, *st_ob;
    int nb_cnt, nb_mol;
    int64_t zero_from, zero_n;
};
// workspace (int32): cnt[E+1] | flag[E] | mol_first[B+1] | mol_ob[B+1] | pad | three look-back state arrays (64-bit words)
BondGraphWs bond_graph_ws(const int64_t* edge_index, const int64_t* atom_mol, int64_t E, int64_t B, int32_t* ws, int mode) {
    BondGraphWs w{};
    w.A.mode = mode;
    w.A.src = edge_index;  w.A.dst = edge_index + E;  w.A.atom_mol = atom_mol;  w.A.E = E;  w.A.B = B;
    w.A.cnt = ws;
    w.A.flag = w.A.cnt + E + 1;
    w.A.mol_first = w.A.flag + E;
    w.A.mol_ob = w.A.mol_first + B + 1;
    int32_t* p = w.A.mol_ob + B + 1;
    if ((uintptr_t)p & 7) ++p;
    w.nb_cnt = (int)((E + kScanChunk - 1) / kScanChunk);
    w.nb_mol = (int)((B + kScanChunk - 1) / kScanChunk);
    w.st_cnt = reinterpret_cast<unsigned long long*>(p);
    w.st_mol = w.st_cnt + w.nb_cnt;
    w.st_ob = w.st_mol + w.nb_mol;
    w.zero_n = (reinterpret_cast<int32_t*>(w.st_ob + w.nb_mol) - ws);
    return w;
}
}  // namespace

extern "C" int64_t fn_bond_graph_ws(int64_t E, int64_t B) {
    if (E < 0 || B < 0) return 0;
    return (2 * E + 1) + 2 * (B + 1) + 2 + 2 * ((E + kScanChunk - 1) / kScanChunk + 2 * ((B + kScanChunk - 1) / kScanChunk));
}

extern "C" int fn_bond_graph_count(const int64_t* edge_index, const int64_t* atom_mol, int64_t E, int64_t N, int64_t B, int mode, int32_t* ws,
                        int64_t* total, fn_stream_t stream) {
    if (E < 0 || N < 0 || B < 0 || E >= (1ll << 31) - 1 || !ws || !total || (E > 0 && (!edge_index || !atom_mol)) || (mode != 0 && mode != 1))
        return fail(FN_EINVAL, "fn_bond_graph_count: bad argument");
    BondGraphWs w = bond_graph_ws(edge_index, atom_mol, E, B, ws, mode);
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(k_zero2_i32, dim3(flat_grid(w.zero_n, kGridCap)), dim3(kBlock), 0, st, ws, w.zero_n, ws, (int64_t)0);
    if (E > 0) {
        const int g = flat_grid(E, kGridCap);
        hipLaunchKernelGGL(k_bg_mol_hist, dim3(g), dim3(kBlock), 0, st, w.A);
        hipLaunchKernelGGL(k_scan_lookback, dim3(w.nb_mol), dim3(256), 0, st, w.A.mol_first + 1, B, w.st_mol);
        hipLaunchKernelGGL(k_bg_count, dim3(g), dim3(kBlock), 0, st, w.A);
        hipLaunchKernelGGL(k_scan_lookback, dim3(w.nb_cnt), dim3(256), 0, st, w.A.cnt + 1, E, w.st_cnt);
        hipLaunchKernelGGL(k_scan_lookback, dim3(w.nb_mol), dim3(256), 0, st, w.A.mol_ob + 1, B, w.st_ob);
    }
    hipLaunchKernelGGL(k_bg_total, dim3(1), dim3(64), 0, st, w.A, total);
    return launch_status("fn_bond_graph_count");
}

extern "C" int fn_bond_graph_fill(const int64_t* edge_index, const int64_t* atom_mol, int64_t E, int64_t N, int64_t B, int mode, const int32_t* ws,
                       int64_t* out, int64_t total, fn_stream_t stream) {
    if (E < 0 || N < 0 || B < 0 || total < 0 || !ws || (total > 0 && !out) || (E > 0 && (!edge_index || !atom_mol)) || (mode != 0 && mode != 1))
        return fail(FN_EINVAL, "fn_bond_graph_fill: bad argument");
    if (E == 0 || total == 0) return 0;
    BondGraphWs w = bond_graph_ws(edge_index, atom_mol, E, B, const_cast<int32_t*>(ws), mode);
    hipLaunchKernelGGL(k_bg_fill, dim3(flat_grid(E, kGridCap)), dim3(kBlock), 0, S(stream), w.A, out, total);
    return launch_status("fn_bond_graph_fill");
}
