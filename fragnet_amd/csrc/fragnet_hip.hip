// libfragnet_hip.so -- gfx950 (MI355X / CDNA4) kernels for FragNet's four-level message passing.
// C-ABI declared in include/fragnet_hip.h (which cites the reference call sites replaced).
//
// Layout conventions used by every row kernel below:
//   * node tables are [rows, 128] fp32; one 32-lane half-wavefront owns one row, each lane one
//     float4 (16 B) => a wave64 moves two 512-B rows per load instruction, fully coalesced;
//   * with H heads, head h owns the 32/H consecutive lanes [h*LPH, (h+1)*LPH) of the half-wave, so
//     every per-head reduction is an xor-butterfly over LPH lanes and never leaves the half-wave;
//   * a block is 256 threads = 8 rows; grids are capped and grid-strided so that the number of
//     per-block partial-sum rows any backward kernel writes is bounded by FN_MAX_PART.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>
#include <functional>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "fragnet_hip.h"
#include "fn_internal.h"

namespace {

thread_local char tl_err[256] = "";

int fail(int code, const char* what) {
    snprintf(tl_err, sizeof(tl_err), "%s", what);
    return code;
}

int launch_status(const char* where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(tl_err, sizeof(tl_err), "%s: %s", where, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// (S(), kBlock, kRows, kGridCap, kBwdRows, row_grid, flat_grid, edge_class, lin_blocks, FN_TRY, FN_DISPATCH_H: fn_internal.h)
constexpr int kRowDotsBwdBlocks = 512;    // blocks of k_row_dots_sorted_bwd (each writes one J*128-wide partial row)
using fni::GatFwdArgs;
using fni::prep_gat_fwd;
using fni::launch_gat_fwd;
using fni::launch_gat_fwd_pair;
using fni::launch_gat_fwd_lin;
using fni::launch_gat_fwd_pair_lin;
using fni::GatBwdOneArgs;
using fni::CuTask;
using fni::CuTasks;
using fni::GsdSegTask;
using fni::GsdSegTasks;
using fni::prep_gat_bwd_one;
using fni::launch_gat_bwd_one3;
using fni::launch_gat_cu;
using fni::launch_gsd_seg;

inline int bwd_grid(int64_t rows) {
    int64_t g = (rows + kBwdRows - 1) / kBwdRows;
    if (g < 1) g = 1;
    if (g > FN_MAX_PART) g = FN_MAX_PART;
    return (int)g;
}


// =====================================================================================
// Attention level
// =====================================================================================
#include "gat_fwd.inc"

#include "gat_bwd_two.inc"

// =====================================================================================
// Full-width edge term (atom graph / fragment graph), produced directly in destination-sorted order
// =====================================================================================
// s_sorted[pos, j] = <feat[eid(pos), :], A[j, off:off+128]>, 0 at loop positions (eid >= m_real)
__global__ __launch_bounds__(kBlock) void k_row_dots_sorted(const float* __restrict__ feat, const float* __restrict__ A,
                                                            int lda, int off, int J, fn_gat_plan pl,
                                                            float* __restrict__ s_sorted) {
    const int lane = threadIdx.x & 31;
    float4 a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = (q < J) ? ld4(A + q * lda + off + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t g0, g1;
    block_groups(pl.m, kRows, g0, g1);
    for (int64_t gi = g0; gi < g1; ++gi) {
        const int64_t pos = gi * kRows + (threadIdx.x >> 5);
        if (pos >= pl.m) continue;
        const int eid = pl.eid_d[pos];
        float mine = 0.f;
        if (eid < pl.m_real) {                       // uniform inside the half-wave
            const float4 v = ld4(feat + (size_t)eid * FN_D + lane * 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q < J) {
                    const float d = head_sum<32>(dot4(v, a[q]));
                    if (lane == q) mine = d;
                }
            }
        }
        if (lane < J) s_sorted[(size_t)lane * pl.m + pos] = mine;        // head-major [J][m]
    }
}

// g_feat[e,:] = sum_j g_s_sorted[inv(e), j] A[j];  part [grid, J*128]: partial sums of g_A[j,:] = sum_e g_s[e,j] feat[e,:]
struct RowDotsBwdArgs {
    const float *g_s_sorted, *feat, *A;
    int lda, off, J;
    fn_gat_plan pl;
    float* g_feat;
    float* part;
    const float* addend;
    int g_is_orig, nblk;
    // one-pass backward (gat_bwd_one.inc): g_feat rows are the finished gradient rows of the level whose RAW output is `feat`; their
    // dots c = <g, feat>, u = <g, out2> - c sigma are written with them (cu_c == null: not wanted; engine path with J = 4 heads only)
    const float *cu_out2, *cu_sigma;
    float *cu_c, *cu_u;
    const int32_t* n_real;   // nullable device word: edges (rows of feat) at or behind *n_real are padding: not read, not written
};
// edges [e0, e1) in original order, taken interleaved by the block's half-waves; vb: the block's slot (row) in T.part
__device__ __forceinline__ void row_dots_sorted_bwd_range(const RowDotsBwdArgs& T, float (*sR)[FN_D], int64_t e0, int64_t e1, int vb) {
    const float* __restrict__ g_s_sorted = T.g_s_sorted;
    const float* __restrict__ feat = T.feat;
    const float* __restrict__ A = T.A;
    const int lda = T.lda, off = T.off, J = T.J, g_is_orig = T.g_is_orig;
    const fn_gat_plan& pl = T.pl;
    float* g_feat = T.g_feat;
    float* __restrict__ part = T.part;
    const float* addend = T.addend;
    const int lane = threadIdx.x & 31, hw = threadIdx.x >> 5;
    float4 a[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (i < J) ? ld4(A + i * lda + off + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // rows are walked in original edge order (sequential feat / g_feat rows).  The per-head gradient comes either in
    // original order, edge-major [m_real][J] (written that way by the destination pass: one 16-byte read), or in
    // destination-sorted head-major order [J][m] through the inverse permutation (autograd path)
    if (g_is_orig && J == 4) {
        // engine path: 4 heads, gradient in original edge order.  Four rows per trip, every load issued before the
        // first use (a single-row loop is one dependent round trip per row: 15 us for 28 k rows)
        for (int64_t base = e0; base < e1; base += 4 * kRows) {
            float4 v[4], ad[4], gs[4], o2[4];
            float sgm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t e = base + u * kRows + hw;
                e = e < e1 ? e : e1 - 1;
                v[u] = ld4(feat + e * FN_D + lane * 4);
                gs[u] = ld4(g_s_sorted + e * 4);
                ad[u] = addend ? ld4(addend + e * FN_D + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                o2[u] = (g_feat && T.cu_c && T.cu_out2) ? ld4(T.cu_out2 + e * FN_D + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                sgm[u] = (g_feat && T.cu_c && T.cu_out2) ? T.cu_sigma[e * 4 + (lane >> 3)] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t e = base + u * kRows + hw;
                if (e >= e1) continue;
                fma4(q[0], gs[u].x, v[u]);  fma4(q[1], gs[u].y, v[u]);  fma4(q[2], gs[u].z, v[u]);  fma4(q[3], gs[u].w, v[u]);
                if (!g_feat) continue;          // parameter partials only: the rows' term rides in a GEMM epilogue (RowAdd)
                float4 acc = ad[u];
                fma4(acc, gs[u].x, a[0]);  fma4(acc, gs[u].y, a[1]);  fma4(acc, gs[u].z, a[2]);  fma4(acc, gs[u].w, a[3]);
                st4(g_feat + e * FN_D + lane * 4, acc);
                if (T.cu_c) {            // four heads of eight lanes: the row's dots with its level's raw and second output rows
                    const float cc = head_sum<8>(dot4(acc, v[u])), uu = head_sum<8>(dot4(acc, o2[u]));
                    if ((lane & 7) == 0) {
                        const int64_t at = e * 4 + (lane >> 3);
                        T.cu_c[at] = cc;
                        if (T.cu_u) T.cu_u[at] = uu - cc * sgm[u];      // (null: the deferred one-pass backward wants c only)
                    }
                }
            }
        }
    } else
    for (int64_t e = e0 + hw; e < e1; e += kRows) {
        const size_t pos = g_is_orig ? 0 : (size_t)pl.inv_d[e];
        const float4 v = ld4(feat + e * FN_D + lane * 4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < J) {
                const float gs = g_is_orig ? g_s_sorted[(size_t)e * J + i] : g_s_sorted[(size_t)i * pl.m + pos];
                fma4(acc, gs, a[i]);
                fma4(q[i], gs, v);
            }
        }
        if (addend) { const float4 a0 = ld4(addend + e * FN_D + lane * 4); acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w; }
        st4(g_feat + e * FN_D + lane * 4, acc);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < J) {                                    // J is a kernel argument: uniform branch
            st4(&sR[hw][lane * 4], q[i]);
            __syncthreads();
            if (threadIdx.x < FN_D) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < kRows; ++w) v += sR[w][threadIdx.x];
                part[(size_t)(i * FN_D + threadIdx.x) * FN_MAX_PART + vb] = v;   // column-major
            }
            __syncthreads();
        }
    }
}
__device__ __forceinline__ void row_dots_sorted_bwd_body(const RowDotsBwdArgs& T, float (*sR)[FN_D], int vb, int nb) {
    // block_groups() for a virtual block index (the kernel may share its launch with another body)
    const int64_t m_live = T.n_real && *T.n_real < T.pl.m_real ? (int64_t)*T.n_real : T.pl.m_real;
    const int64_t groups = (T.pl.m_real + kRows - 1) / kRows, per = (groups + nb - 1) / nb;
    const int64_t live_blocks = (m_live + per * kRows - 1) / (per * kRows);
    const int64_t g0 = (int64_t)xcd_block_real(vb, nb, (int)(live_blocks < nb ? live_blocks : nb)) * per, g1 = g0 + per < groups ? g0 + per : groups;
    const int64_t e1 = g1 * kRows < m_live ? g1 * kRows : m_live;
    row_dots_sorted_bwd_range(T, sR, g0 * kRows < e1 ? g0 * kRows : e1, e1, vb);
}
__global__ __launch_bounds__(kBlock) void k_row_dots_sorted_bwd(RowDotsBwdArgs T) {
    __shared__ float sR[kRows][FN_D];
    row_dots_sorted_bwd_body(T, sR, (int)blockIdx.x, (int)gridDim.x);
}
// The source pass of a level and the backward of its edge term both depend on the destination pass only, never on each
// other: one launch (a dependent launch costs ~5 us however small the kernel is; 5 such pairs per backward pass).
template <int H, int RB>
__global__ __launch_bounds__(RB * 32) void k_gat_bwd_src_rd(GatBwdSrcArgs A, RowDotsBwdArgs T) {
    static_assert(RB * 32 == kBlock && RB == kRows, "both bodies run 8 half-waves per block");
    __shared__ float sA[RB][2 * FN_D];
    if ((int)blockIdx.x < A.nblk) gat_bwd_src_body<H, RB>(A, sA, (int)blockIdx.x, A.nblk);
    else row_dots_sorted_bwd_body(T, reinterpret_cast<float(*)[FN_D]>(&sA[0][0]), (int)blockIdx.x - A.nblk, T.nblk);
}

// x_sorted[pos, :] = x[eid(pos), :] (zeros at loop positions): raw edge attributes are permuted once per batch
__device__ __forceinline__ void sort_edge_attr_body(const float* __restrict__ x, int K, const fn_gat_plan& pl,
                                                    float* __restrict__ x_sorted, int vb, int nb) {
    const int64_t total = pl.m * K;
    for (int64_t i = (int64_t)vb * blockDim.x + threadIdx.x; i < total; i += (int64_t)nb * blockDim.x) {
        const int64_t pos = i / K;
        const int k = (int)(i % K);
        const int eid = pl.eid_d[pos];
        x_sorted[(size_t)k * pl.m + pos] = eid < pl.m_real ? x[(size_t)eid * K + k] : 0.f;   // [K][m]
    }
}
// the same attribute in SOURCE order (x_src[:, q] = the attribute of the edge at source-order position q): the one-pass backward
// streams it; x_raw != null: from the original edge order ([m_real][K]), else from the destination-sorted copy ([K][m])
__device__ __forceinline__ void sort_edge_attr_src_body(const float* __restrict__ x_raw, const float* __restrict__ x_sorted, int K,
                                                        const fn_gat_plan& pl, float* __restrict__ x_src, int vb, int nb) {
    const int64_t total = pl.m * K;
    for (int64_t i = (int64_t)vb * blockDim.x + threadIdx.x; i < total; i += (int64_t)nb * blockDim.x) {
        const int64_t pos = i / K;
        const int k = (int)(i % K);
        const int dq = pl.dpos_s[pos];
        float v;
        if (x_raw) { const int eid = pl.eid_d[dq];  v = eid < pl.m_real ? x_raw[(size_t)eid * K + k] : 0.f; }
        else v = x_sorted[(size_t)k * pl.m + dq];
        x_src[(size_t)k * pl.m + pos] = v;
    }
}
__global__ void k_sort_edge_attr(const float* __restrict__ x, int K, fn_gat_plan pl, float* __restrict__ x_sorted, int by_source) {
    if (by_source) sort_edge_attr_src_body(x, nullptr, K, pl, x_sorted, (int)blockIdx.x, (int)gridDim.x);
    else sort_edge_attr_body(x, K, pl, x_sorted, (int)blockIdx.x, (int)gridDim.x);
}

// one block per column of the column-major partials [cols][FN_MAX_PART]
__device__ __forceinline__ void colsum_body(int vb, float* s16, const float* __restrict__ part, int n_rows,
                                            float* __restrict__ out, int ld, int off) {
    const int col = vb;
    float mine = 0.f;
    for (int r = threadIdx.x; r < n_rows; r += 1024) mine += part[(size_t)col * FN_MAX_PART + r];
    const float v = block_sum_1024(mine, s16);
    if (threadIdx.x == 0) out[(col / FN_D) * ld + off + (col % FN_D)] = v;
}
__global__ __launch_bounds__(1024) void k_colsum(const float* __restrict__ part, int n_rows, int cols,
                                                  float* __restrict__ out, int ld, int off) {
    __shared__ float s16[16];
    colsum_body(blockIdx.x, s16, part, n_rows, out, ld, off);
}

// =====================================================================================
// Segment sum / gather / segment softmax (the torch-scatter operator surface)
// =====================================================================================
__global__ __launch_bounds__(kBlock) void k_segment_sum128(const float* __restrict__ src, int64_t src_ld,
                                                           const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ perm, int32_t pos_base,
                                                           float* __restrict__ out, int64_t n_seg) {
    const int lane = threadIdx.x & 31;
    int64_t g0, g1;
    block_groups(n_seg, kRows, g0, g1);
    for (int64_t gi = g0; gi < g1; ++gi) {
        const int64_t s = gi * kRows + (threadIdx.x >> 5);
        if (s >= n_seg) continue;
        const int beg = rowptr[s] - pos_base, deg = rowptr[s + 1] - rowptr[s];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

        for (int i = 0; i < deg; ++i) {
            const float4 v = ld4(src + (size_t)perm[beg + i] * src_ld + lane * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        st4(out + s * FN_D + lane * 4, acc);
    }
}

// long segments (pooling: ~26 atoms per molecule, few hundred segments): one block per segment, its 8 half-waves
// take rows hw, hw+8, ... and the 8 partial rows are added in a fixed order
__global__ __launch_bounds__(kBlock) void k_segment_sum128_wide(const float* __restrict__ src, int64_t src_ld,
                                                                const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ perm, int32_t pos_base,
                                                                float* __restrict__ out, int64_t n_seg) {
    __shared__ float sS[kRows][FN_D];
    const int lane = threadIdx.x & 31, hw = threadIdx.x >> 5;
    for (int64_t s = blockIdx.x; s < n_seg; s += gridDim.x) {
        const int beg = rowptr[s] - pos_base, deg = rowptr[s + 1] - rowptr[s];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = hw; i < deg; i += kRows) {
            const float4 v = ld4(src + (size_t)perm[beg + i] * src_ld + lane * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        st4(&sS[hw][lane * 4], acc);
        __syncthreads();
        if (threadIdx.x < FN_D) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < kRows; ++w) v += sS[w][threadIdx.x];
            out[s * FN_D + threadIdx.x] = v;
        }
        __syncthreads();
    }
}

// Last layer's fragment tail, first half, as ONE launch: blocks [0, nblk_seg) = the atom -> fragment sum (the body of
// k_segment_sum128_wide) followed by the fragment's node scalars <row, att[h, dst/src block]> from the finished row (a head's
// 128/H columns are consecutive threads: shuffle reduction) -- two launches less than sum, node scalars, edge term; blocks
// [nblk_seg, +nblk_rd) = the fragment graph's edge term <new_fbond[e], att[h, mid block]> (the body of k_row_dots_sorted),
// which depends on neither.  Each of the three was a ~5 us latency-floor launch.
struct FragTailArgs {
    const float* src;  const int32_t* rowptr;  const int32_t* perm;  int32_t pos_base;  float* out;  int64_t n_seg;
    const float* att;  int att_w, dst_off, src_off;  float* s_dst;  float* s_src;  int nblk_seg;
    const float* feat;  const float* A;  int lda, off;  fn_gat_plan pl;  float* s_sorted;  int nblk_rd;
};
template <int H>
__global__ __launch_bounds__(kBlock) void k_frag_tail(FragTailArgs T) {
    __shared__ float sS[kRows][FN_D];
    const int lane = threadIdx.x & 31, hw = threadIdx.x >> 5;
    if ((int)blockIdx.x < T.nblk_seg) {
        constexpr int d = FN_D / H;
        static_assert(d <= 64, "a head's columns must lie inside one wave");
        const int t = threadIdx.x, head = (t & 127) / d, c = (t & 127) % d;
        const float a_d = T.att[head * T.att_w + T.dst_off + c], a_s = T.att[head * T.att_w + T.src_off + c];
        for (int64_t s = blockIdx.x; s < T.n_seg; s += T.nblk_seg) {
            const int beg = T.rowptr[s] - T.pos_base, deg = T.rowptr[s + 1] - T.rowptr[s];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = hw; i < deg; i += kRows) {
                const float4 v = ld4(T.src + (size_t)T.perm[beg + i] * FN_D + lane * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            st4(&sS[hw][lane * 4], acc);
            __syncthreads();
            if (t < FN_D) {                                  // waves 0 and 1, whole
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < kRows; ++w) v += sS[w][t];
                T.out[s * FN_D + t] = v;
                float pd = v * a_d, ps = v * a_s;
#pragma unroll
                for (int off = d / 2; off > 0; off >>= 1) { pd += __shfl_xor(pd, off);  ps += __shfl_xor(ps, off); }
                if (c == 0) { T.s_dst[s * H + head] = pd;  T.s_src[s * H + head] = ps; }
            }
            __syncthreads();
        }
        return;
    }
    const int vb = (int)blockIdx.x - T.nblk_seg, nb = T.nblk_rd;
    const fn_gat_plan& pl = T.pl;
    float4 a[H];
#pragma unroll
    for (int q = 0; q < H; ++q) a[q] = ld4(T.A + q * T.lda + T.off + lane * 4);
    const int64_t groups = (pl.m + kRows - 1) / kRows, per = (groups + nb - 1) / nb;
    const int64_t g0 = (int64_t)xcd_block(vb, nb) * per, g1 = g0 + per < groups ? g0 + per : groups;
    for (int64_t gi = g0; gi < g1; ++gi) {
        const int64_t pos = gi * kRows + hw;
        if (pos >= pl.m) continue;
        const int eid = pl.eid_d[pos];
        float mine = 0.f;
        if (eid < pl.m_real) {                               // uniform inside the half-wave
            const float4 v = ld4(T.feat + (size_t)eid * FN_D + lane * 4);
#pragma unroll
            for (int q = 0; q < H; ++q) {
                const float dd = head_sum<32>(dot4(v, a[q]));
                if (lane == q) mine = dd;
            }
        }
        if (lane < H) T.s_sorted[(size_t)lane * pl.m + pos] = mine;      // head-major [H][m]
    }
}

// pooled = cat(sum of a molecule's atom rows, sum of its fragment rows): [B, 256] in one launch (gat2.py:820-823).
// grid (B, 2): y = 0 atoms -> columns 0..127, y = 1 fragments -> columns 128..255; one block per molecule and half.
__global__ __launch_bounds__(kBlock) void k_pool_cat(const float* __restrict__ x_atoms, const float* __restrict__ x_frags,
                                                     fn_seg_plan sa, fn_seg_plan sf, float* __restrict__ out) {
    __shared__ float sS[kRows][FN_D];
    const fn_seg_plan& sp = blockIdx.y ? sf : sa;
    const float* src = blockIdx.y ? x_frags : x_atoms;
    const int lane = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int64_t s = blockIdx.x;
    const int beg = sp.rowptr[s] - sp.pos_base, deg = sp.rowptr[s + 1] - sp.rowptr[s];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = hw; i < deg; i += kRows) {
        const float4 v = ld4(src + (size_t)sp.perm[beg + i] * FN_D + lane * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    st4(&sS[hw][lane * 4], acc);
    __syncthreads();
    if (threadIdx.x < FN_D) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kRows; ++w) v += sS[w][threadIdx.x];
        out[s * 2 * FN_D + blockIdx.y * FN_D + threadIdx.x] = v;
    }
}
// its backward: g_atoms[i,:] = g[batch[i], 0:128], g_frags[f,:] = g[frag_batch[f], 128:256]
__global__ void k_pool_cat_bwd(const float* __restrict__ g, const int64_t* __restrict__ batch, const int64_t* __restrict__ frag_batch,
                               float* __restrict__ g_atoms, float* __restrict__ g_frags, int64_t N, int64_t F) {
    const int64_t total = (N + F) * 32;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i >> 5;
        const int c4 = (int)(i & 31);
        if (r < N) st4(g_atoms + r * FN_D + c4 * 4, ld4(g + batch[r] * 2 * FN_D + c4 * 4));
        else st4(g_frags + (r - N) * FN_D + c4 * 4, ld4(g + frag_batch[r - N] * 2 * FN_D + FN_D + c4 * 4));
    }
}

// loss = sum_i w[i] * sum_t (out[i,t] - y[i,t])^2 / (sum_i w[i] * T) and its gradient w.r.t. out, one block
// (MSELoss of train/utils.py:341 restricted to the rows with weight 1; B*T is a few hundred to a few thousand)
__global__ __launch_bounds__(1024) void k_masked_mse(const float* __restrict__ out, const float* __restrict__ y,
                                                     const float* __restrict__ w, int64_t B, int T, float* __restrict__ loss,
                                                     float* __restrict__ g_out) {
    __shared__ float s16[16];
    __shared__ float sW;
    float sq = 0.f, ws = 0.f;
    for (int64_t i = threadIdx.x; i < B * T; i += 1024) {
        const float d = out[i] - y[i], wi = w[i / T];
        sq = fmaf(wi * d, d, sq);
    }
    for (int64_t i = threadIdx.x; i < B; i += 1024) ws += w[i];
    const float S = block_sum_1024(sq, s16);
    __syncthreads();
    const float W = block_sum_1024(ws, s16);
    const float denom = W * (float)T;
    if (threadIdx.x == 0) { loss[0] = S / denom;  sW = denom; }
    __syncthreads();
    const float scale = 2.f / sW;
    for (int64_t i = threadIdx.x; i < B * T; i += 1024) g_out[i] = scale * w[i / T] * (out[i] - y[i]);
}

// multi-task classification loss (compute_bce_loss, train/utils.py:297-304): BCE-with-logits averaged over the valid
// entries (target > -0.5 = label present, row weight > 0 = real molecule) and its gradient, one block
__global__ __launch_bounds__(1024) void k_masked_bce(const float* __restrict__ out, const float* __restrict__ y,
                                                     const float* __restrict__ w, int64_t B, int T, float* __restrict__ loss,
                                                     float* __restrict__ g_out) {
    __shared__ float s16[16];
    __shared__ float sN;
    float sum = 0.f, cnt = 0.f;
    for (int64_t i = threadIdx.x; i < B * T; i += 1024) {
        const float x = out[i], t = y[i];
        if (t > -0.5f && w[i / T] > 0.f) {
            const float tt = fmaxf(t, 0.f);
            sum += fmaxf(x, 0.f) - x * tt + log1pf(expf(-fabsf(x)));
            cnt += 1.f;
        }
    }
    const float S = block_sum_1024(sum, s16);
    __syncthreads();
    const float N = block_sum_1024(cnt, s16);
    if (threadIdx.x == 0) { loss[0] = S / N;  sN = N; }
    __syncthreads();
    const float inv = 1.f / sN;
    for (int64_t i = threadIdx.x; i < B * T; i += 1024) {
        const float x = out[i], t = y[i];
        const bool valid = t > -0.5f && w[i / T] > 0.f;
        g_out[i] = valid ? (1.f / (1.f + expf(-x)) - fmaxf(t, 0.f)) * inv : 0.f;
    }
}

// ---- sum_k coef_k * masked_mse_k over up to four (prediction, target, row weight) triples in two multi-block launches:
// pass 1 writes per-block partial (sum w d^2, sum w) of every task, pass 2 lets every block re-add the partials in a
// fixed order (so all blocks agree bit for bit), block 0 writes the total loss and all blocks write their slice of the
// gradients coef_k * 2 / (W_k T_k) * w * (out - y).  coef_k = c_k * (scale_dev[idx_k] if idx_k >= 0 else 1).
constexpr int kMseBlocks = 64;
struct MseTask {
    const float *out, *y, *w;
    float* g;
    int64_t B;
    int T, scale_idx;
    float c;
};
struct MseTasks {
    MseTask t[4];
    int n;
    const float* scale_dev;
};
__global__ __launch_bounds__(256) void k_mse_multi_partial(MseTasks M, float* __restrict__ part /*[n][kMseBlocks][2]*/) {
    __shared__ float s4[4];
    const MseTask& t = M.t[blockIdx.y];
    float sq = 0.f, ws = 0.f;
    const int64_t total = t.B * t.T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)kMseBlocks * 256) {
        const float d = t.out[i] - t.y[i], wi = t.w[i / t.T];
        sq = fmaf(wi * d, d, sq);
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < t.B; i += (int64_t)kMseBlocks * 256) ws += t.w[i];
    for (int pass = 0; pass < 2; ++pass) {
        float v = pass ? ws : sq;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) part[((size_t)blockIdx.y * kMseBlocks + blockIdx.x) * 2 + pass] = s4[0] + s4[1] + s4[2] + s4[3];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_mse_multi_finish(MseTasks M, const float* __restrict__ part, float* __restrict__ loss) {
    __shared__ float sS[4], sW[4];
    if (threadIdx.x < M.n) {
        float S = 0.f, W = 0.f;
        for (int b = 0; b < kMseBlocks; ++b) {
            S += part[((size_t)threadIdx.x * kMseBlocks + b) * 2];
            W += part[((size_t)threadIdx.x * kMseBlocks + b) * 2 + 1];
        }
        sS[threadIdx.x] = S;
        sW[threadIdx.x] = W * (float)M.t[threadIdx.x].T;
    }
    __syncthreads();
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        float L = 0.f;
        for (int k = 0; k < M.n; ++k) {
            const float coef = M.t[k].c * (M.t[k].scale_idx >= 0 ? M.scale_dev[M.t[k].scale_idx] : 1.f);
            L += coef * (sS[k] / sW[k]);
        }
        loss[0] = L;
    }
    const MseTask& t = M.t[blockIdx.y];
    const float coef = t.c * (t.scale_idx >= 0 ? M.scale_dev[t.scale_idx] : 1.f);
    const float scale = coef * 2.f / sW[blockIdx.y];
    const int64_t total = t.B * t.T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)kMseBlocks * 256)
        t.g[i] = scale * t.w[i / t.T] * (t.out[i] - t.y[i]);
}

__global__ void k_segment_sum_any(const float* __restrict__ src, int64_t src_ld, const int32_t* __restrict__ rowptr,
                                  const int32_t* __restrict__ perm, int32_t pos_base, float* __restrict__ out,
                                  int64_t n_seg, int64_t width) {
    const int64_t total = n_seg * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / width, c = i % width;
        const int beg = rowptr[s] - pos_base, deg = rowptr[s + 1] - rowptr[s];
        float acc = 0.f;
        for (int k = 0; k < deg; ++k) acc += src[(size_t)perm[beg + k] * src_ld + c];
        out[i] = acc;
    }
}

// out[i,:] = table[index[i],:] (+ addend[i,:] when given; addend may alias out)
__global__ void k_gather_rows4(const float* __restrict__ table, const int64_t* __restrict__ index,
                               float* out, int64_t rows, int64_t w4, const float* addend) {
    const int64_t total = rows * w4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / w4, c = i % w4;
        float4 v = ld4(table + ((size_t)index[r] * w4 + c) * 4);
        if (addend) { const float4 a = ld4(addend + i * 4); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        st4(out + i * 4, v);
    }
}

__global__ void k_gather_rows1(const float* __restrict__ table, const int64_t* __restrict__ index,
                               float* __restrict__ out, int64_t rows, int64_t width) {
    const int64_t total = rows * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = table[(size_t)index[i / width] * width + (i % width)];
}

__global__ void k_segment_softmax(const float* __restrict__ logits, const int32_t* __restrict__ rowptr,
                                  const int32_t* __restrict__ perm, int32_t pos_base, float* __restrict__ probs,
                                  int64_t n_seg, int64_t width) {
    const int64_t total = n_seg * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / width, c = i % width;
        const int beg = rowptr[s] - pos_base, deg = rowptr[s + 1] - rowptr[s];
        float mx = -INFINITY;
        for (int k = 0; k < deg; ++k) mx = fmaxf(mx, logits[(size_t)perm[beg + k] * width + c]);
        float den = 0.f;
        for (int k = 0; k < deg; ++k) den += expf(logits[(size_t)perm[beg + k] * width + c] - mx);
        for (int k = 0; k < deg; ++k) {
            const size_t o = (size_t)perm[beg + k] * width + c;
            probs[o] = expf(logits[o] - mx) / den;
        }
    }
}

__global__ void k_segment_softmax_bwd(const float* __restrict__ probs, const float* __restrict__ g_probs,
                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm,
                                      int32_t pos_base, float* __restrict__ g_logits, int64_t n_seg, int64_t width) {
    const int64_t total = n_seg * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / width, c = i % width;
        const int beg = rowptr[s] - pos_base, deg = rowptr[s + 1] - rowptr[s];
        float dot = 0.f;
        for (int k = 0; k < deg; ++k) {
            const size_t o = (size_t)perm[beg + k] * width + c;
            dot = fmaf(probs[o], g_probs[o], dot);
        }
        for (int k = 0; k < deg; ++k) {
            const size_t o = (size_t)perm[beg + k] * width + c;
            g_logits[o] = probs[o] * (g_probs[o] - dot);
        }
    }
}

// =====================================================================================
// dropout + ReLU epilogue (Philox-4x32-10)
// =====================================================================================
template <bool BWD>
__device__ __forceinline__ void dropout_act_body(const float* __restrict__ a, const float* __restrict__ y_saved, float* __restrict__ o,
                                                 int64_t numel, float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                                 int relu, int vb, int nb) {
    const int64_t n4 = (numel + 3) / 4;
    if (offset_dev) offset += *offset_dev;
    const float inv_keep = p < 1.f ? 1.f / (1.f - p) : 0.f;
    for (int64_t i = (int64_t)vb * blockDim.x + threadIdx.x; i < n4; i += (int64_t)nb * blockDim.x) {
        float m[4] = {1.f, 1.f, 1.f, 1.f};
        if (p > 0.f) {
            const uint4 r = philox4x32(offset + (uint64_t)i, seed);
            m[0] = keep_scale(r.x, p, inv_keep); m[1] = keep_scale(r.y, p, inv_keep);
            m[2] = keep_scale(r.z, p, inv_keep); m[3] = keep_scale(r.w, p, inv_keep);
        }
        const int64_t e0 = i * 4;
        if (e0 + 3 < numel) {
            const float4 v = ld4(a + e0);
            float4 res;
            if (!BWD) {
                res = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
                if (relu) { res.x = fmaxf(res.x, 0.f); res.y = fmaxf(res.y, 0.f); res.z = fmaxf(res.z, 0.f); res.w = fmaxf(res.w, 0.f); }
            } else {
                const float4 ys = relu ? ld4(y_saved + e0) : make_float4(1.f, 1.f, 1.f, 1.f);
                res = make_float4(ys.x > 0.f || !relu ? v.x * m[0] : 0.f, ys.y > 0.f || !relu ? v.y * m[1] : 0.f,
                                  ys.z > 0.f || !relu ? v.z * m[2] : 0.f, ys.w > 0.f || !relu ? v.w * m[3] : 0.f);
            }
            st4(o + e0, res);
        } else {
            for (int q = 0; q < 4 && e0 + q < numel; ++q) {
                float v = a[e0 + q] * m[q];
                if (!BWD) { if (relu) v = fmaxf(v, 0.f); }
                else if (relu && !(y_saved[e0 + q] > 0.f)) v = 0.f;
                o[e0 + q] = v;
            }
        }
    }
}
template <bool BWD>
__global__ void k_dropout_act(const float* __restrict__ a, const float* __restrict__ y_saved, float* __restrict__ o,
                              int64_t numel, float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                              int relu) {
    dropout_act_body<BWD>(a, y_saved, o, numel, p, seed, offset, offset_dev, relu, (int)blockIdx.x, (int)gridDim.x);
}

// torch.optim.Adam's update rule (no amsgrad) on one flat tensor: the reference's optimiser, finetune_gat2.py:257
// (vb, nb): this block's index / the number of blocks working on the tensor -- k_adam's own grid, or the riders' range of the
// deferred-reduction launch (AdamRide below)
__device__ __forceinline__ void adam_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                          int64_t n, float lr_over_bc1, float beta1, float beta2, float eps, float inv_sqrt_bc2, float wd,
                                          const int64_t* __restrict__ step_dev, const float* __restrict__ lr_dev, int vb, int nb) {
    if (step_dev) {      // captured in a hipGraph: step count and learning rate live in device memory, bias corrections here
        __shared__ float s2[2];
        if (threadIdx.x == 0) {
            const double st = (double)*step_dev;
            const double bc1 = 1.0 - pow((double)beta1, st), bc2 = 1.0 - pow((double)beta2, st);
            s2[0] = (float)((double)*lr_dev / bc1);
            s2[1] = (float)(1.0 / sqrt(bc2));
        }
        __syncthreads();
        lr_over_bc1 = s2[0];
        inv_sqrt_bc2 = s2[1];
    }
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)vb * blockDim.x + threadIdx.x; i < n4; i += (int64_t)nb * blockDim.x) {
        float4 pp = ld4(p + i * 4), gg = ld4(g + i * 4), mm = ld4(m + i * 4), vv = ld4(v + i * 4);
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gq = G[q] + wd * P[q];
            M[q] = M[q] + (gq - M[q]) * (1.f - beta1);
            V[q] = V[q] * beta2 + (1.f - beta2) * gq * gq;
            P[q] -= lr_over_bc1 * (M[q] / (sqrtf(V[q]) * inv_sqrt_bc2 + eps));
        }
        st4(p + i * 4, pp); st4(m + i * 4, mm); st4(v + i * 4, vv);
    }
    if (vb == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        const float gq = g[i] + wd * p[i];
        const float mq = m[i] + (gq - m[i]) * (1.f - beta1);
        const float vq = v[i] * beta2 + (1.f - beta2) * gq * gq;
        m[i] = mq; v[i] = vq;
        p[i] -= lr_over_bc1 * (mq / (sqrtf(vq) * inv_sqrt_bc2 + eps));
    }
}
// An Adam update of parameters whose gradients were final BEFORE the encoder's backward pass began (the prediction head's, 84 % of a
// FragNetFineTune) rides in one of that pass's launches: blocks [first, first + nblk).  Independent of everything the pass computes;
// the step's own Adam launch then covers the rest of the flat buffer only (fn_encoder.adam_rider).  Where: FN_TUNE_RIDER_AT.
struct AdamRide {
    fn_adam_slice a;
    int first, nblk;             // nblk == 0: none
};
__device__ __forceinline__ void adam_ride(const AdamRide& R) {
    adam_body(R.a.p, R.a.g, R.a.m, R.a.v, R.a.n, 0.f, R.a.beta1, R.a.beta2, R.a.eps, 0.f, R.a.weight_decay, R.a.step_dev, R.a.lr_dev,
              (int)blockIdx.x - R.first, R.nblk);
}
inline AdamRide make_adam_ride(const fn_adam_slice* a, int first, int threads, int pieces) {
    AdamRide R{};
    if (a && a->n > 0) {
        const int64_t per = pieces > 0 ? pieces : 1;
        const int64_t nb = (a->n / 4 + per * threads - 1) / (per * threads);          // 16-byte pieces per thread
        R.a = *a;  R.first = first;  R.nblk = (int)(nb < 1 ? 1 : nb > 4096 ? 4096 : nb);
    }
    return R;
}
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       int64_t n, float lr_over_bc1, float beta1, float beta2, float eps, float inv_sqrt_bc2, float wd,
                       const int64_t* __restrict__ step_dev, const float* __restrict__ lr_dev) {
    adam_body(p, g, m, v, n, lr_over_bc1, beta1, beta2, eps, inv_sqrt_bc2, wd, step_dev, lr_dev, (int)blockIdx.x, (int)gridDim.x);
}

__global__ void k_edge_concat(const float* __restrict__ x, const float* __restrict__ e_attr,
                              const int64_t* __restrict__ edge_index, float* __restrict__ out, int64_t E) {
    const int64_t total = E * 96;     // float4 slots per row: 3 x 32
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / 96;
        const int q = (int)(i % 96);
        float4 v;
        if (q < 32) v = ld4(x + (size_t)edge_index[e] * FN_D + q * 4);
        else if (q < 64) v = ld4(x + (size_t)edge_index[E + e] * FN_D + (q - 32) * 4);
        else v = ld4(e_attr + (size_t)e * FN_D + (q - 64) * 4);
        st4(out + i * 4, v);
    }
}

// =====================================================================================
// Prediction-head small ops (gat2.py:631-637, 745-751: Linear -> dropout -> ReLU stacks on [molecules, width]).
// The dense products stay library GEMMs; these are the launches around them.
// =====================================================================================
// g_x = (y > 0) ? g_y * scale : 0  and  colsum[c] = sum_rows g_x[:, c]   (bias gradient of the Linear below)
// y = relu(dropout(.)) is positive only where the element was kept and passed the ReLU, so no Philox replay.
// One block owns a strip of 32 columns for ALL rows: the column sums are deterministic and need no second pass.
__global__ __launch_bounds__(256) void k_gate_colsum(const float* __restrict__ g_y, const float* __restrict__ y,
                                                     float* __restrict__ g_x, float* __restrict__ sums, int64_t rows,
                                                     int cols, float scale, int64_t rows_per_chunk) {
    // blockIdx.y = row chunk (tall inputs: the pretrain towers run on every edge / atom); its column sums go to
    // sums[chunk][cols] and k_sum_chunks adds the chunks in order.  One chunk: sums is the result itself.
    __shared__ float4 sm[32][8];
    const int c4 = threadIdx.x & 7, rl = threadIdx.x >> 3;           // 8 float4 columns x 32 row lanes
    const int col = blockIdx.x * 32 + c4 * 4;
    const int64_t r_begin = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r_end = r_begin + rows_per_chunk < rows ? r_begin + rows_per_chunk : rows;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < cols) {
        int64_t r = r_begin + rl;
        for (; r + 96 < r_end; r += 128) {                          // four rows in flight per thread
            float4 g[4], v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { g[q] = ld4(g_y + (r + 32 * q) * cols + col); v[q] = ld4(y + (r + 32 * q) * cols + col); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 o = make_float4(v[q].x > 0.f ? g[q].x * scale : 0.f, v[q].y > 0.f ? g[q].y * scale : 0.f,
                                             v[q].z > 0.f ? g[q].z * scale : 0.f, v[q].w > 0.f ? g[q].w * scale : 0.f);
                st4(g_x + (r + 32 * q) * cols + col, o);
                acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            }
        }
        for (; r < r_end; r += 32) {
            const float4 g = ld4(g_y + r * cols + col), v = ld4(y + r * cols + col);
            const float4 o = make_float4(v.x > 0.f ? g.x * scale : 0.f, v.y > 0.f ? g.y * scale : 0.f,
                                         v.z > 0.f ? g.z * scale : 0.f, v.w > 0.f ? g.w * scale : 0.f);
            st4(g_x + r * cols + col, o);
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
    }
    sm[rl][c4] = acc;
    __syncthreads();
    if (rl == 0 && col < cols) {
        float4 t = sm[0][c4];
        for (int q = 1; q < 32; ++q) { const float4 u = sm[q][c4]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        st4(sums + (size_t)blockIdx.y * cols + col, t);
    }
}
// out[c] = sum over chunks of part[chunk][c] in a fixed order: 16 chunk lanes each add every 16th chunk, then the 16
// lane sums are added in lane order (a serial walk over ~110 dependent loads took 25 us)
__global__ __launch_bounds__(256) void k_sum_chunks(const float* __restrict__ part, int chunks, int64_t width, float* __restrict__ out) {
    __shared__ float sm[16][17];
    const int cl = threadIdx.x & 15, ql = threadIdx.x >> 4;
    const int64_t c = (int64_t)blockIdx.x * 16 + cl;
    float t = 0.f;
    if (c < width)
        for (int q = ql; q < chunks; q += 16) t += part[(size_t)q * width + c];
    sm[ql][cl] = t;
    __syncthreads();
    if (ql == 0 && c < width) {
        float v = sm[0][cl];
        for (int q = 1; q < 16; ++q) v += sm[q][cl];
        out[c] = v;
    }
}

// g_x = (y > 0) ? g_y * scale : 0 for up to four tensors (numel % 4 == 0, 16-byte aligned) in one launch
struct GateTask {
    const float *g, *y;
    float* o;
    int64_t n4;
    int first, nblk;
};
struct GateTasks {
    GateTask t[4];
    int n, blocks;
    float scale;
};
__global__ void k_gate_many(GateTasks G) {
    int ti = 0;
    while (ti + 1 < G.n && (int)blockIdx.x >= G.t[ti + 1].first) ++ti;
    const GateTask& t = G.t[ti];
    const float sc = G.scale;
    for (int64_t i = (int64_t)((int)blockIdx.x - t.first) * blockDim.x + threadIdx.x; i < t.n4; i += (int64_t)t.nblk * blockDim.x) {
        const float4 g = ld4(t.g + 4 * i), v = ld4(t.y + 4 * i);
        st4(t.o + 4 * i, make_float4(v.x > 0.f ? g.x * sc : 0.f, v.y > 0.f ? g.y * sc : 0.f, v.z > 0.f ? g.z * sc : 0.f,
                                     v.w > 0.f ? g.w * sc : 0.f));
    }
}

// y[m, c] = <x[m, :], w[c, :]> + b[c] for a handful of outputs (the last Linear of a head: n_classes columns).
// One wave per row; the row stays in registers while the C weight rows stream from L1.
__global__ __launch_bounds__(256) void k_small_linear(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ y, int64_t M, int K, int C,
                                                      int64_t M_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) {                                                  // padding rows of a static-shape batch: defined, 0
        if (row < M_out && lane < C) y[row * C + lane] = 0.f;
        return;
    }
    const int k4 = K / 4;                                            // K % 4 == 0
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
        for (int j = lane; j < k4; j += 64) acc += dot4(ld4(x + row * K + 4 * j), ld4(w + (size_t)c * K + 4 * j));
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) y[row * C + c] = acc + (b ? b[c] : 0.f);
    }
}

// backward of the above in one launch: g_x[m, k] = sum_c g[m, c] w[c, k];  dW[c, k] = sum_m g[m, c] x[m, k];
// db[c] = sum_m g[m, c].  A block owns 16 columns k for all rows (deterministic sums, 64 rows in flight);
// CM >= C bounds the registers.
template <int CM>
__global__ __launch_bounds__(256) void k_small_linear_bwd(const float* __restrict__ g, const float* __restrict__ x,
                                                          const float* __restrict__ w, float* __restrict__ g_x,
                                                          float* __restrict__ dW, float* __restrict__ db, int64_t M, int K, int C,
                                                          int64_t rows_per_chunk, float gate_scale) {
    // blockIdx.y = row chunk: dW / db then point at per-chunk partials [chunk][C*K] / [chunk][C] (k_sum_chunks finishes)
    __shared__ float4 sm[64][4];
    const int c4 = threadIdx.x & 3, rl = threadIdx.x >> 2;           // 4 float4 columns x 64 row lanes
    const int col = blockIdx.x * 16 + c4 * 4;
    const bool live = col < K;
    const int64_t m_begin = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t m_end = m_begin + rows_per_chunk < M ? m_begin + rows_per_chunk : M;
    dW += (size_t)blockIdx.y * C * K;
    db += (size_t)blockIdx.y * C;
    float4 wv[CM], acc[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        wv[c] = (live && c < C) ? ld4(w + (size_t)c * K + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (live) {
        // U rows per trip (four; two for the widest class: registers), every load issued before the first use (one row per trip made the 512-row head a chain of
        // eight dependent round trips: 7.4 us for 1 MB)
        constexpr int U = CM > 4 ? 2 : 4;
        for (int64_t m0 = m_begin + rl; m0 < m_end; m0 += U * 64) {
            float4 xv[U];
            float gv[U][CM];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t m = m0 + 64 * u < m_end ? m0 + 64 * u : m0;
                xv[u] = ld4(x + m * K + col);
#pragma unroll
                for (int c = 0; c < CM; ++c) gv[u][c] = c < C ? g[m * C + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t m = m0 + 64 * u;
                if (m >= m_end) continue;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (c < C) {
                        o.x += gv[u][c] * wv[c].x; o.y += gv[u][c] * wv[c].y; o.z += gv[u][c] * wv[c].z; o.w += gv[u][c] * wv[c].w;
                        acc[c].x += gv[u][c] * xv[u].x; acc[c].y += gv[u][c] * xv[u].y; acc[c].z += gv[u][c] * xv[u].z; acc[c].w += gv[u][c] * xv[u].w;
                    }
                }
                if (gate_scale > 0.f) {                             // x = relu(dropout(.)) of the layer below: its backward, fused
                    o.x = xv[u].x > 0.f ? o.x * gate_scale : 0.f;  o.y = xv[u].y > 0.f ? o.y * gate_scale : 0.f;
                    o.z = xv[u].z > 0.f ? o.z * gate_scale : 0.f;  o.w = xv[u].w > 0.f ? o.w * gate_scale : 0.f;
                }
                st4(g_x + m * K + col, o);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        if (c < C) {                                                // uniform
            sm[rl][c4] = acc[c];
            __syncthreads();
            if (rl == 0 && live) {
                float4 t = sm[0][c4];
                for (int q = 1; q < 64; ++q) { const float4 u = sm[q][c4]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
                st4(dW + (size_t)c * K + col, t);
            }
            __syncthreads();
        }
    }
    if (blockIdx.x == 0) {                                          // bias gradient: one wave per class, fixed order
        const int lane = threadIdx.x & 63;
        for (int c = threadIdx.x >> 6; c < C; c += 4) {
            float t = 0.f;
            for (int64_t m = m_begin + lane; m < m_end; m += 64) t += g[m * C + c];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if (lane == 0) db[c] = t;
        }
    }
}

// The last Linear of a head, the loss on its outputs and that Linear's INPUT gradient in one launch (round 4; VERDICT r3 item 4: three
// dependent launches of 4-6 us each were one row-local computation apart from two sums).  Once the loss's denominator is known -- it
// depends on the row weights / label mask only, never on the predictions, so every block recomputes it from L2 (a few KB) -- a row's
// prediction, its loss gradient g = dL/dy and its input gradient g_x = g w (gated by x > 0) need nothing from other rows: one wave per
// row, the row in registers throughout.  What does cross rows is left as data for the next launch (the head's first fn_dense_bwd
// call carries the blocks, dense_head.inc small_dw_*): dW = g^T x, db = colsum(g), and the loss value as per-block partial sums
// (already divided by the denominator).  No ticket, no atomic: fixed-order sums only.  Numbers: y, g, g_x are bit-identical to
// fn_small_linear_f32 -> fn_masked_mse/bce_f32 -> fn_small_linear_bwd_f32 (same operation order).
template <int CM>
__global__ __launch_bounds__(256) void k_small_linear_loss(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                           const float* __restrict__ target, const float* __restrict__ row_w, const int kind,
                                                           float* __restrict__ y, float* __restrict__ g, float* __restrict__ g_x,
                                                           const float gate_scale, float* __restrict__ loss_part, const int64_t M, const int K,
                                                           const int C, const int64_t M_out) {
    __shared__ float s4[4];
    __shared__ float sden;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float cnt = 0.f;
    if (kind == FN_LOSS_MSE) {
        for (int64_t i = threadIdx.x; i < M_out; i += 256) cnt += row_w[i];
    } else {
        for (int64_t i = threadIdx.x; i < M_out * C; i += 256) cnt += (target[i] > -0.5f && row_w[i / C] > 0.f) ? 1.f : 0.f;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) s4[wv] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) sden = ((s4[0] + s4[1]) + (s4[2] + s4[3])) * (kind == FN_LOSS_MSE ? (float)C : 1.f);
    __syncthreads();
    const float den = sden;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    const int k4 = K / 4;                                            // K % 4 == 0, K <= 1024: four 16-byte pieces per lane at most
    float lsum = 0.f;
    if (row < M) {
        float4 xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q;
            xv[q] = j < k4 ? ld4(x + row * K + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float wm = row_w[row];
        float gc[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            gc[c] = 0.f;
            if (c < C) {                                             // uniform
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = lane + 64 * q;
                    if (j < k4) acc += dot4(xv[q], ld4(w + (size_t)c * K + 4 * j));
                }
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
                const float yv = acc + (b ? b[c] : 0.f), t = target[row * C + c];
                if (kind == FN_LOSS_MSE) {
                    const float d = yv - t, scale = 2.f / den;
                    gc[c] = scale * wm * d;
                    lsum = fmaf(wm * d, d, lsum);
                } else if (t > -0.5f && wm > 0.f) {
                    const float tt = fmaxf(t, 0.f);
                    lsum += fmaxf(yv, 0.f) - yv * tt + log1pf(expf(-fabsf(yv)));
                    gc[c] = (1.f / (1.f + expf(-yv)) - tt) * (1.f / den);
                }
                if (lane == 0) { y[row * C + c] = yv;  g[row * C + c] = gc[c]; }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q;
            if (j < k4) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (c < C) {
                        const float4 wc = ld4(w + (size_t)c * K + 4 * j);
                        o.x += gc[c] * wc.x; o.y += gc[c] * wc.y; o.z += gc[c] * wc.z; o.w += gc[c] * wc.w;
                    }
                }
                if (gate_scale > 0.f) {                              // x = relu(dropout(.)) of the layer below: its backward, fused
                    o.x = xv[q].x > 0.f ? o.x * gate_scale : 0.f;  o.y = xv[q].y > 0.f ? o.y * gate_scale : 0.f;
                    o.z = xv[q].z > 0.f ? o.z * gate_scale : 0.f;  o.w = xv[q].w > 0.f ? o.w * gate_scale : 0.f;
                }
                st4(g_x + row * K + 4 * j, o);
            }
        }
    } else if (row < M_out && lane < C) {
        y[row * C + lane] = 0.f;                                     // padding rows of a static-shape batch: defined, 0
    }
    if (lane == 0) s4[wv] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) loss_part[blockIdx.x] = ((s4[0] + s4[1]) + (s4[2] + s4[3])) / den;
}

#include "linear128.inc"

template <int KQ, bool VEC, bool PF>
__global__ __launch_bounds__(kLinThreads) void k_linear128(const float* __restrict__ X, int K, const float* __restrict__ Bt,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int64_t M,
                                                   fn_act_epilogue mk, NodeScalarEpi ns) {
    extern __shared__ __attribute__((aligned(16))) float sBt[];
    linear128_body<KQ, VEC, PF>(sBt, X, K, Bt, bias, Y, M, mk, ns, (int)blockIdx.x, (int)gridDim.x);
}

template <int KQ, bool VEC, bool PF>
__global__ __launch_bounds__(kLinThreads) void k_linear128_multi(LinTasks T) {
    extern __shared__ __attribute__((aligned(16))) float sBt[];
    int ti = 0;
    while (ti + 1 < T.n && (int)blockIdx.x >= T.t[ti + 1].first) ++ti;
    const LinTask& t = T.t[ti];
    linear128_body<KQ, VEC, PF>(sBt, t.X, t.K ? t.K : T.K, t.Bt, t.bias, t.Y, t.M, t.mk, t.ns, (int)blockIdx.x - t.first, t.nblk,
                                RowAdd{nullptr, nullptr, 0}, CuEpi{nullptr, nullptr, nullptr, nullptr, nullptr, 0}, t.n_real);
}

// the grouped launch when a task carries a RowAdd term and cannot ride in an attention launch
__global__ __launch_bounds__(kLinThreads) void k_linear128_multi_ra(LinTasks T) {
    extern __shared__ __attribute__((aligned(16))) float sBt[];
    int ti = 0;
    while (ti + 1 < T.n && (int)blockIdx.x >= T.t[ti + 1].first) ++ti;
    const LinTask& t = T.t[ti];
    linear128_body<32, true, false, true>(sBt, t.X, 128, t.Bt, t.bias, t.Y, t.M, t.mk, t.ns, (int)blockIdx.x - t.first, t.nblk, t.ra,
                                          CuEpi{nullptr, nullptr, nullptr, nullptr, nullptr, 0}, t.n_real);
}

// layer 0: all three projections read raw features only (K = 17 bond, 6 connection, 167 atom features at the reference's sizes), so
// they share a launch although their reduction lengths need two instantiations of the body: tasks with K <= 20 run the short one,
// the others the K <= 168 one (whose operand tile sets the launch's LDS size)
__global__ __launch_bounds__(kLinThreads) void k_linear128_layer0(LinTasks T) {
    extern __shared__ __attribute__((aligned(16))) float sBt[];
    int ti = 0;
    while (ti + 1 < T.n && (int)blockIdx.x >= T.t[ti + 1].first) ++ti;
    const LinTask& t = T.t[ti];
    const RowAdd no_ra{nullptr, nullptr, 0};
    const CuEpi no_cu{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (t.K > 20) linear128_body<44, false, false>(sBt, t.X, t.K, t.Bt, t.bias, t.Y, t.M, t.mk, t.ns, (int)blockIdx.x - t.first, t.nblk, no_ra, no_cu, t.n_real);
    else linear128_body<5, false, false>(sBt, t.X, t.K, t.Bt, t.bias, t.Y, t.M, t.mk, t.ns, (int)blockIdx.x - t.first, t.nblk, no_ra, no_cu, t.n_real);
}

// the one-pass backward's second launch of a layer: input-gradient products (one 64 x 64 tile per workgroup; RowAdd epilogue where a
// task carries one; the epilogue also writes the dot c = <g, out> of the rows it finishes, CuEpi)  ||  the parameter-gradient partials of
// the atom graph's edge term
// (GS: the deferred form -- every lane sums one dz segment into g_s_dst, one more MFMA step adds the rank-4 term, GsdEpi; c is the only dot left)
// (GO: the mixed form's boundary launches -- deferred rows in, rows of a layer with a second forward output out: both epilogues)
template <bool GS = false, bool GO = false>
__global__ __launch_bounds__(kBlock, 3) void k_lin_rd_cu(LinTasks T, RowDotsBwdArgs R) {
    extern __shared__ __attribute__((aligned(16))) float sBt[];
    __shared__ float sR[kRows][FN_D];
    const int b = (int)blockIdx.x;
    if (b < T.total) { lin_side_block<true, GS, GO>(sBt, T, b);  return; }
    row_dots_sorted_bwd_body(R, sR, b - T.total, R.nblk);
}

// Bt[k][n] = W[n][k]  (W is nn.Linear.weight [128, K])
__global__ void k_transpose_w(const float* __restrict__ W, int K, float* __restrict__ Bt) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 thr = 32x8
    for (int r = ty; r < 32; r += 8) tile[r][tx] = (k0 + tx < K) ? W[(size_t)(n0 + r) * K + k0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < K) Bt[(size_t)(k0 + r) * 128 + n0 + tx] = tile[tx][r];
}

// all projection weights of the encoder in one launch: matrix z -> Bt base + z * 192 * 128
struct TransposeMany {
    const float* W[3 * FN_MAX_LAYERS];
    int K[3 * FN_MAX_LAYERS];
};
__device__ __forceinline__ void transpose_many_body(const TransposeMany& tm, float* __restrict__ bt_base, float (*tile)[33],
                                                    int z, int by, int bx) {
    const int K = tm.K[z];
    const int k0 = bx * 32, n0 = by * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (k0 >= K) return;                                            // whole block
    const float* W = tm.W[z];
    float* Bt = bt_base + (size_t)z * 192 * FN_D;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = (k0 + tx < K) ? W[(size_t)(n0 + r) * K + k0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < K) Bt[(size_t)(k0 + r) * FN_D + n0 + tx] = tile[tx][r];
}
__global__ void k_transpose_many(TransposeMany tm, float* __restrict__ bt_base) {
    __shared__ float tile[32][33];
    transpose_many_body(tm, bt_base, tile, (int)blockIdx.z, (int)blockIdx.y, (int)blockIdx.x);
}

// extents of every molecule in the index spaces of a collated batch (fn_internal.h: MolExt), from the molecule CSRs and the level plans
using fni::MolExt;
struct MolExtArgs {
    const int32_t *mol_atoms, *mol_frags;       // molecule CSRs over atoms / fragments (global positions)
    int32_t base_atoms, base_frags;
    fn_gat_plan bond, atom, fbond, frag;
    int n_mol;
    MolExt* out;
    const int32_t* counts_dev;                  // nullable: device count of the real molecules (the rest is padding)
    int32_t* real_rows;                         // nullable: [4] real atoms / bonds / fragments / connections = where the last real molecule ends
};
__device__ __forceinline__ void mol_extents_body(const MolExtArgs& A, int vb) {
    const int mol = vb * blockDim.x + threadIdx.x;
    if (mol >= A.n_mol) return;
    MolExt x;
    const int a0 = A.mol_atoms[mol] - A.base_atoms, a1 = A.mol_atoms[mol + 1] - A.base_atoms;
    const int f0 = A.mol_frags[mol] - A.base_frags, f1 = A.mol_frags[mol + 1] - A.base_frags;
    // the by-source CSR of the atom graph counts, before atom a, the bonds leaving atoms < a (+ one loop item per atom)
    const int la = A.atom.m > A.atom.m_real ? 1 : 0, lf = A.frag.m > A.frag.m_real ? 1 : 0;
    const int b0 = A.atom.rowptr_s[a0] - A.atom.pos_base_s - la * a0, b1 = A.atom.rowptr_s[a1] - A.atom.pos_base_s - la * a1;
    const int c0 = A.frag.rowptr_s[f0] - A.frag.pos_base_s - lf * f0, c1 = A.frag.rowptr_s[f1] - A.frag.pos_base_s - lf * f1;
    x.a0 = a0;  x.na = a1 - a0;  x.b0 = b0;  x.nb = b1 - b0;  x.f0 = f0;  x.nf = f1 - f0;  x.c0 = c0;  x.nc = c1 - c0;
    x.eb0 = A.bond.rowptr_d[b0] - A.bond.pos_base_d;   x.meb = A.bond.rowptr_d[b1] - A.bond.pos_base_d - x.eb0;
    x.ea0 = A.atom.rowptr_d[a0] - A.atom.pos_base_d;   x.mea = A.atom.rowptr_d[a1] - A.atom.pos_base_d - x.ea0;
    if (A.fbond.rowptr_d) {
        x.ef0 = A.fbond.rowptr_d[c0] - A.fbond.pos_base_d;  x.mef = A.fbond.rowptr_d[c1] - A.fbond.pos_base_d - x.ef0;
    } else { x.ef0 = 0;  x.mef = 0; }
    x.ec0 = A.frag.rowptr_d[f0] - A.frag.pos_base_d;   x.mec = A.frag.rowptr_d[f1] - A.frag.pos_base_d - x.ec0;
    A.out[mol] = x;
    if (A.real_rows) {
        int n_real = A.counts_dev ? *A.counts_dev : A.n_mol;
        n_real = n_real < A.n_mol ? n_real : A.n_mol;
        if (mol == n_real - 1) { A.real_rows[0] = a1;  A.real_rows[1] = b1;  A.real_rows[2] = f1;  A.real_rows[3] = c1; }
        if (n_real <= 0 && mol == 0) { A.real_rows[0] = 0;  A.real_rows[1] = 0;  A.real_rows[2] = 0;  A.real_rows[3] = 0; }
    }
}
#include "dense_head.inc"
#include "mol_tail.inc"

// Everything the encoder's forward pass needs before its first projection, none of which depends on the other: W^T of
// every projection, dropout of the atom features, and the permutation of the two raw edge-attribute tensors into
// destination order.  One launch of four block ranges instead of four launches (each was 5 us of latency).
struct EncPrologue {
    TransposeMany tm;
    float* bt_base;
    int n_t;                                                        // 24 blocks per matrix
    const float* dx;  float* dy;  int64_t dnumel;  float p;  uint64_t seed, offset;  const uint64_t* offset_dev;  int n_d;
    const float* sx[2];  float* so[2];  int sK[2];  fn_gat_plan spl[2];  int n_s[2];
    const float* ssr[2];  const float* sss[2];  float* sso[2];  int n_ss[2];   // the same two attributes in SOURCE order (one-pass backward): from raw, else from sorted
    float* zp;  int64_t zn;  int n_z;                               // buffer zeroed once per forward (edge-term scratch: loop positions stay 0)
    MolExtArgs mx;  int n_x;                                        // molecule extents for the molecule-resident backward (256 molecules per block)
    // deferred one-pass backward (GsdEpi): R[z][h][k] = sum_{c < 32} att[z][h * att_w[z] + c] * W[z][(32 h + c) * 128 + k] for the K = 128
    // projections z (four heads; two blocks per matrix)
    const float* rW[3 * FN_MAX_LAYERS];  const float* rA[3 * FN_MAX_LAYERS];  int rAw[3 * FN_MAX_LAYERS];  float* rOut;  int n_r;
};
__global__ __launch_bounds__(256) void k_enc_prologue(EncPrologue A) {
    __shared__ float tile[32][33];
    int b = blockIdx.x;
    if (b < A.n_t) {
        const int z = b / 24, rem = b % 24;
        transpose_many_body(A.tm, A.bt_base, tile, z, rem / 6, rem % 6);
        return;
    }
    b -= A.n_t;
    if (b < A.n_d) {
        dropout_act_body<false>(A.dx, nullptr, A.dy, A.dnumel, A.p, A.seed, A.offset, A.offset_dev, 0, b, A.n_d);
        return;
    }
    b -= A.n_d;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (b < A.n_s[q]) {
            sort_edge_attr_body(A.sx[q], A.sK[q], A.spl[q], A.so[q], b, A.n_s[q]);
            return;
        }
        b -= A.n_s[q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (b < A.n_ss[q]) {
            sort_edge_attr_src_body(A.ssr[q], A.sss[q], A.sK[q], A.spl[q], A.sso[q], b, A.n_ss[q]);
            return;
        }
        b -= A.n_ss[q];
    }
    if (b < A.n_z) {
        for (int64_t i = (int64_t)b * blockDim.x + threadIdx.x; i < A.zn; i += (int64_t)A.n_z * blockDim.x) A.zp[i] = 0.f;
        return;
    }
    b -= A.n_z;
    if (b < A.n_x) { mol_extents_body(A.mx, b);  return; }
    b -= A.n_x;
    if (b < A.n_r) {
        const int z = b >> 1, o = (b & 1) * 256 + (int)threadIdx.x, hh = o >> 7, k = o & 127;
        const float* W = A.rW[z];
        const float* a = A.rA[z] + hh * A.rAw[z];
        float acc = 0.f;
        if (W) {
#pragma unroll 8
            for (int c = 0; c < 32; ++c) acc = fmaf(a[c], W[(size_t)(32 * hh + c) * 128 + k], acc);
        }
        A.rOut[(size_t)z * 512 + o] = acc;
    }
}

// Weight gradient: block = `rows_per_block` rows in chunks of 32 staged through double-buffered LDS.  Wave w owns
// output rows o in [32(w&3), +32) and the (w>>2)-th group of CTW 16-column tiles of X, so NH = 2 column groups
// put two waves on every SIMD.  part [grid][128*K + 128]: dW partial followed by the db partial.
constexpr int kWgChunk = 32;
// gsd != null (four heads): the deferred form of the one-pass attention backward, see wgrad128.inc -- the dY rows get their missing
// term g_s_dst[row] a_dst as they are fetched, and the block also leaves U[h][k] = sum_rows g_s_dst[row, h] X[row, k], S[h] =
// sum_rows g_s_dst[row, h] in upart [grid][4 K + 4] (the chunk's g_s_dst rows travel in the padding columns of the X tile)
// (GD: compile-time, like DF in wgrad128.inc -- as a run-time test inside fetch / stash it slowed the plain products down)
template <int CTW, int NH, bool GD = false>
__device__ __forceinline__ void wgrad_body(float* smem, const float* __restrict__ dY, const float* __restrict__ X, int K,
                                           int64_t M, int rows_per_block, float* __restrict__ part, int bid,
                                           const float* __restrict__ gsd_ = nullptr, const float* __restrict__ a_dst = nullptr,
                                           int att_w = 0, float* __restrict__ upart_ = nullptr) {
    const float* __restrict__ gsd = GD ? gsd_ : nullptr;
    float* __restrict__ upart = GD ? upart_ : nullptr;
    constexpr int NT = 256 * NH;
    constexpr int XW = 16 * CTW * NH;            // padded X width held in LDS
    constexpr int XLD = XW + 16;                 // XW is a multiple of 32 for every instantiation but <1,1>
    float* sY = smem;                                   // [2][32][144]
    float* sX = smem + 2 * kWgChunk * kBtLd;            // [2][32][XLD]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int wo = w & 3, wc = w >> 2;
    const int64_t m_begin = (int64_t)bid * rows_per_block;
    const int64_t m_end = m_begin + rows_per_block < M ? m_begin + rows_per_block : M;
    const int n_chunks = (int)((m_end - m_begin + kWgChunk - 1) / kWgChunk);

    constexpr int YPT = 1024 / NT;                            // dY float4 per thread per chunk
    constexpr int XPT = (kWgChunk * XW + NT - 1) / NT;        // X scalars per thread per chunk
    float4 ry[YPT];
    float rx[XPT];
    float4 rg = make_float4(0.f, 0.f, 0.f, 0.f);           // threads 0..31: the chunk row's g_s_dst (deferred term)
    // the term is added where the dY operand leaves LDS: the lane's two columns 32 wo + i, + 16 belong to head wo, so a step costs one
    // more LDS read (the row's g_s_dst[wo], staged in the X tile's padding columns) and two FMAs.  (Per-thread scalar loads of
    // g_s_dst in the fetch made the layer-0 workgroups -- latency-bound, one round trip per chunk, the tail of the launch -- 60 % longer.)
    const int wo_ = (tid >> 6) & 3;
    const float ad0 = GD ? a_dst[wo_ * att_w + (tid & 15)] : 0.f, ad1 = GD ? a_dst[wo_ * att_w + 16 + (tid & 15)] : 0.f;
    constexpr int UP = (XW + NT / 4 - 1) / (NT / 4);        // U columns per thread: thread = (head tid & 3, column tid >> 2 [+ NT / 4])
    float uacc[UP];
#pragma unroll
    for (int q = 0; q < UP; ++q) uacc[q] = 0.f;
    float sacc = 0.f;
    auto fetch = [&](int c) {
        const int64_t base = m_begin + (int64_t)c * kWgChunk;
#pragma unroll
        for (int q = 0; q < YPT; ++q) {
            const int idx = tid + q * NT, r = idx >> 5, c4 = idx & 31;
            ry[q] = (base + r < m_end) ? ld4(dY + (base + r) * 128 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (GD && tid < kWgChunk) rg = (base + tid < m_end) ? ld4(gsd + (base + tid) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const int idx = tid + q * NT, r = idx / XW, cc = idx % XW;
            rx[q] = (r < kWgChunk && base + r < m_end && cc < K) ? X[(base + r) * K + cc] : 0.f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < YPT; ++q) {
            const int idx = tid + q * NT, r = idx >> 5, c4 = idx & 31;
            st4(sY + (buf * kWgChunk + r) * kBtLd + c4 * 4, ry[q]);
        }
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const int idx = tid + q * NT, r = idx / XW, cc = idx % XW;
            if (r < kWgChunk) sX[(buf * kWgChunk + r) * XLD + cc] = rx[q];
        }
        if (GD && tid < kWgChunk) st4(sX + (buf * kWgChunk + tid) * XLD + XW, rg);       // the row's padding columns XW .. XW + 3
    };

    f32x4 acc[2][CTW];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[u][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};

    if (n_chunks > 0) { fetch(0); stash(0); }
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) fetch(c + 1);
#pragma unroll
        for (int s = 0; s < kWgChunk / 4; ++s) {
            const float* yrow = sY + (buf * kWgChunk + 4 * s + kq) * kBtLd + 32 * wo + i;
            const float* xrow = sX + (buf * kWgChunk + 4 * s + kq) * XLD + 16 * CTW * wc + i;
            float a0 = yrow[0], a1 = yrow[16];
            if constexpr (GD) {
                const float gq = sX[(buf * kWgChunk + 4 * s + kq) * XLD + XW + wo];
                a0 = fmaf(gq, ad0, a0);
                a1 = fmaf(gq, ad1, a1);
            }
            bsum[0] += a0;
            bsum[1] += a1;
            float bv[CTW];
#pragma unroll
            for (int cc = 0; cc < CTW; ++cc) bv[cc] = xrow[16 * cc];
#pragma unroll
            for (int cc = 0; cc < CTW; ++cc) {
                acc[0][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[cc], acc[0][cc], 0, 0, 0);
                acc[1][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[cc], acc[1][cc], 0, 0, 0);
            }
        }
        if (GD && upart) {  // the side product U, S of this chunk (rows past the end were stashed as zeros)
            const int hh = tid & 3;
#pragma unroll 8
            for (int r = 0; r < kWgChunk; ++r) {
                const float* xr = sX + (buf * kWgChunk + r) * XLD;
                const float g = xr[XW + hh];
#pragma unroll
                for (int q = 0; q < UP; ++q) {
                    const int k = (tid >> 2) + q * (NT / 4);
                    if (k < XW) uacc[q] = fmaf(g, xr[k], uacc[q]);
                }
                sacc += g;
            }
        }
        if (c + 1 < n_chunks) stash(buf ^ 1);
        __syncthreads();
    }
    if (GD && upart) {
        float* up = upart + (size_t)bid * (4 * K + 4);
#pragma unroll
        for (int q = 0; q < UP; ++q) {
            const int k = (tid >> 2) + q * (NT / 4);
            if (k < K) up[(tid & 3) * K + k] = uacc[q];
        }
        if (tid < 4) up[4 * K + tid] = sacc;
    }
    // partials are written in the accumulators' native layout: one coalesced 16-byte store per lane and tile;
    // k_wgrad_reduce maps them back to dW[o][col] while summing over blocks
    constexpr int PW = 128 * XW + 128;
    float* pw = part + (size_t)bid * PW;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int cc = 0; cc < CTW; ++cc)
            st4(pw + ((size_t)((w * 2 + u) * CTW + cc) * 64 + lane) * 4,
                make_float4(acc[u][cc][0], acc[u][cc][1], acc[u][cc][2], acc[u][cc][3]));
    if (wc == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float v = bsum[u];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) pw[(size_t)128 * XW + 32 * wo + 16 * u + i] = v;
        }
    }
}

template <int CTW, int NH>
__global__ __launch_bounds__(256 * NH) void k_linear128_wgrad(const float* __restrict__ dY, const float* __restrict__ X,
                                                              int K, int64_t M, int rows_per_block,
                                                              float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    wgrad_body<CTW, NH>(smem, dY, X, K, M, rows_per_block, part, (int)blockIdx.x);
}

// all weight-gradient partial products of a backward pass that share K (every projection beyond layer 0): one launch
struct WgradTask {
    const float *dY, *X;
    float* part;
    int64_t M;
    int rpb, first;
    int K;                    // k_linear128_wgrad_mixed only: this product's reduction length (0 elsewhere: WgradTasks::K)
    const int32_t* n_real;    // nullable device word: rows at or behind *n_real are padding (zero gradient rows) and are not read
    // the deferred form of the one-pass attention backward (wgrad128.inc): dY lacks g_s_dst[row] a_dst; gsd == null: nothing to add
    const float* gsd;         // [M][4]
    const float* a_dst;       // att + dst_off: head h's 32 floats at a_dst + h * att_w
    int att_w;
    float* upart;             // [blocks][4 K + 4]: U[h][k] = sum_rows gsd[row, h] X[row, k], then S[h] = sum_rows gsd[row, h]
};
constexpr int kMaxWgradTasks = 3 * FN_MAX_LAYERS;
struct WgradTasks {
    WgradTask t[kMaxWgradTasks];
    int n, K;
};
#include "wgrad128.inc"
template <int CTW, int NH>
__global__ __launch_bounds__(256 * NH) void k_linear128_wgrad_multi(WgradTasks T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int ti = 0;
    while (ti + 1 < T.n && (int)blockIdx.x >= T.t[ti + 1].first) ++ti;
    const WgradTask& t = T.t[ti];
    wgrad_body<CTW, NH>(smem, t.dY, t.X, T.K, t.M, t.rpb, t.part, (int)blockIdx.x - t.first);
}

// the weight-gradient partials of layer 0 (raw-feature widths: 17 / 6 / 167 at the reference's sizes) in one launch: every
// product picks the instantiation its K needs (all with two column groups, i.e. 512 threads); the launch's LDS size is the
// largest product's
__global__ __launch_bounds__(512) void k_linear128_wgrad_mixed(WgradTasks T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int ti = 0;
    while (ti + 1 < T.n && (int)blockIdx.x >= T.t[ti + 1].first) ++ti;
    const WgradTask& t = T.t[ti];
    const int bid = (int)blockIdx.x - t.first;
    const int64_t M = t.n_real && *t.n_real < t.M ? (int64_t)*t.n_real : t.M;
    if (t.gsd) {
        if (t.K <= 32) wgrad_body<1, 2, true>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid, t.gsd, t.a_dst, t.att_w, t.upart);
        else if (t.K <= 128) wgrad_body<4, 2, true>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid, t.gsd, t.a_dst, t.att_w, t.upart);
        else wgrad_body<6, 2, true>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid, t.gsd, t.a_dst, t.att_w, t.upart);
    } else if (t.K <= 32) wgrad_body<1, 2>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid);
    else if (t.K <= 128) wgrad_body<4, 2>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid);
    else wgrad_body<6, 2>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid);
}

// every weight-gradient partial product of a backward pass in ONE launch: blocks [0, n128) run the direct K = 128 kernel
// (csrc/wgrad128.inc, two row slices), the others layer 0's products; the launch's LDS size is the larger of the two needs.
// Neither group waits for the other, and the short layer-0 workgroups fill the CUs the long K = 128 ones leave towards the end.
__global__ __launch_bounds__(512) void k_wgrad_all(const WgradTasks W, const WgradTasks W0, int n128) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < n128) {
        int ti = 0;
        while (ti + 1 < W.n && (int)blockIdx.x >= W.t[ti + 1].first) ++ti;
        if (W.t[ti].gsd) wgrad128_block<2, true>(W.t[ti], (int)blockIdx.x - W.t[ti].first, smem);
        else wgrad128_block<2, false>(W.t[ti], (int)blockIdx.x - W.t[ti].first, smem);
        return;
    }
    const int b = (int)blockIdx.x - n128;
    int ti = 0;
    while (ti + 1 < W0.n && b >= W0.t[ti + 1].first) ++ti;
    const WgradTask& t = W0.t[ti];
    const int bid = b - t.first;
    const int64_t M = t.n_real && *t.n_real < t.M ? (int64_t)*t.n_real : t.M;
    if (t.gsd) {
        if (t.K <= 32) wgrad_body<1, 2, true>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid, t.gsd, t.a_dst, t.att_w, t.upart);
        else if (t.K <= 128) wgrad_body<4, 2, true>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid, t.gsd, t.a_dst, t.att_w, t.upart);
        else wgrad_body<6, 2, true>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid, t.gsd, t.a_dst, t.att_w, t.upart);
    } else if (t.K <= 32) wgrad_body<1, 2>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid);
    else if (t.K <= 128) wgrad_body<4, 2>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid);
    else wgrad_body<6, 2>(smem, t.dY, t.X, t.K, M, t.rpb, t.part, bid);
}

// sums the native-layout partials over blocks and scatters them to dW [128][K] / db [128]
template <int CTW, int NH>
__device__ __forceinline__ void wgrad_reduce_body(int vb, float* sm, const float* __restrict__ part, int n_rows, int K,
                                                  float* __restrict__ dW, float* __restrict__ db) {
    constexpr int XW = 16 * CTW * NH;
    constexpr int PW = 128 * XW + 128;
    float(*red)[33] = reinterpret_cast<float(*)[33]>(sm);
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int col = vb * 32 + c;
    float acc = 0.f;
    if (col < PW)
        for (int r = rg; r < n_rows; r += 32) acc += part[(size_t)r * PW + col];
    red[rg][c] = acc;
    __syncthreads();
    if (threadIdx.x < 32 && col < PW) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) v += red[g][threadIdx.x];
        if (col >= 128 * XW) {
            db[col - 128 * XW] = v;
        } else {
            const int r = col & 3, lane = (col >> 2) & 63, tile = col >> 8;        // tile = (w*2+u)*CTW + cc
            const int cc = tile % CTW, wu = tile / CTW, u = wu & 1, w = wu >> 1;
            const int o = 32 * (w & 3) + 16 * u + 4 * (lane >> 4) + r;
            const int xc = 16 * (CTW * (w >> 2) + cc) + (lane & 15);
            if (xc < K) dW[(size_t)o * K + xc] = v;
        }
    }
}

template <int CTW, int NH>
__global__ __launch_bounds__(1024) void k_wgrad_reduce(const float* __restrict__ part, int n_rows, int K,
                                                       float* __restrict__ dW, float* __restrict__ db) {
    __shared__ float sm[32 * 33];
    wgrad_reduce_body<CTW, NH>(blockIdx.x, sm, part, n_rows, K, dW, db);
}

// ---- deferred reductions.  The backward pass leaves every per-block partial (attention-vector and edge-embedding
// partials of each level, the edge-term column sums, the weight-gradient partials) in its own buffer and records a
// task; ONE launch then runs them all (a dependent kernel costs >= 4.6 us of launch-to-launch latency on this
// machine however small it is, and there were 27 of these per step).
enum { RT_FINALIZE = 0, RT_COLSUM = 1, RT_WGRAD = 2 };
struct ReduceTask {
    int kind, first, nblk, H;
    const float *p0, *p1;
    int n0, n1;
    fn_edge_term et;
    const float* att;
    int att_w, dst_off, src_off, K;
    float *o0, *o1, *o2;
    int ld, off, cls, pad_;
    // RT_FINALIZE, deferred form of the one-pass backward (gat_bwd_one.inc DF, four heads): the level's pass left no dL/da_dst partials;
    // four extra blocks (one per head) form it from the weight-gradient kernels' side product: dL/da_dst[c] = sum_k W[c, k] U[h(c), k] + b[c] S[h(c)]
    const float *up, *upW, *upb;      // up [n_up][4 upK + 4] per-block partials of U | S; the projection's weight [128][upK] and bias; up == null: none
    int n_up, upK;
};
constexpr int kMaxReduceTasks = 36;      // one launch for all 30 tasks of a 4-layer backward pass (5.5 KB of kernel arguments)
struct ReduceTasks {
    ReduceTask t[kMaxReduceTasks];
    int first[kMaxReduceTasks + 1];      // first block of every task, packed: a block finds its task by walking THIS array (three
                                         // cache lines of the argument block) -- walking t[].first was one dependent scalar load
                                         // per 152-byte struct, up to 30 in a row from the kernel-argument segment: 10 of the
                                         // launch's 22 us before the first partial was read
    int n;
};
// "fat" bodies for the task-table kernel: a block reduces 8 columns of a column-major [cols][FN_MAX_PART] partial
// array (128 threads per column, contiguous reads), or a 256-column strip of the row-major weight-gradient partials
// (64 float4 columns x 16 row groups: 1 KiB contiguous per wave and row) -- ~1.5 k blocks per backward pass instead
// of ~10 k one-column blocks.
// the same sum with 32 threads per column (a 1024-thread block takes 32 columns: a quarter of the blocks of the 128-thread form --
// the deferred-reduction launch is mostly block scheduling, §6); no LDS, no barrier: the half-wave's butterfly finishes it
__device__ __forceinline__ float colmajor_sum_32(const float* __restrict__ col, int n_rows) {
    const int lane = threadIdx.x & 31;
    float v = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int r = lane;
    for (; r + 96 < n_rows; r += 128) { v += col[r];  v1 += col[r + 32];  v2 += col[r + 64];  v3 += col[r + 96]; }
    for (; r < n_rows; r += 32) v += col[r];
    v = (v + v1) + (v2 + v3);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
template <int CTW, int NH, bool PLAIN = false>
__device__ __forceinline__ void wgrad_reduce_strip(int vb, float* sm, const float* __restrict__ part, int n_rows, int K,
                                                   float* __restrict__ dW, float* __restrict__ db) {
    // a block sums a 1024-column strip of the partial rows: 256 float4 columns x 4 row groups (4 KiB contiguous per wave and
    // row), four loads per thread in flight.  (Round 3: a strip used to be 256 columns x 16 row groups -- 65 blocks of 16
    // waves per product that loaded two float4 each; a quarter of the waves now.)
    constexpr int XW = 16 * CTW * NH;
    constexpr int PW = 128 * XW + 128;
    const int c4 = threadIdx.x & 255, rg = threadIdx.x >> 8;
    const int col0 = vb * 1024 + c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col0 < PW) {
        float4 a1 = acc, a2 = acc, a3 = acc;
        int r = rg;
        for (; r + 12 < n_rows; r += 16) {
            const float4 v0 = ld4(part + (size_t)r * PW + col0), v1 = ld4(part + (size_t)(r + 4) * PW + col0);
            const float4 v2 = ld4(part + (size_t)(r + 8) * PW + col0), v3 = ld4(part + (size_t)(r + 12) * PW + col0);
            acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
            a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; r < n_rows; r += 4) {
            const float4 v = ld4(part + (size_t)r * PW + col0);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        acc.x = (acc.x + a1.x) + (a2.x + a3.x);  acc.y = (acc.y + a1.y) + (a2.y + a3.y);
        acc.z = (acc.z + a1.z) + (a2.z + a3.z);  acc.w = (acc.w + a1.w) + (a2.w + a3.w);
    }
    st4(sm + rg * 1024 + c4 * 4, acc);
    __syncthreads();
    const int col = vb * 1024 + threadIdx.x;
    if (col < PW) {
        const float v = (sm[threadIdx.x] + sm[1024 + threadIdx.x]) + (sm[2048 + threadIdx.x] + sm[3072 + threadIdx.x]);
        if (col >= 128 * XW) {
            db[col - 128 * XW] = v;
        } else if (PLAIN) {                                                          // k_wgrad128_multi: partials are [o][k] already
            dW[col] = v;
        } else {
            const int r = col & 3, lane = (col >> 2) & 63, tile = col >> 8;        // tile = (w*2+u)*CTW + cc
            const int cc = tile % CTW, wu = tile / CTW, u = wu & 1, w = wu >> 1;
            const int o = 32 * (w & 3) + 16 * u + 4 * (lane >> 4) + r;
            const int xc = 16 * (CTW * (w >> 2) + cc) + (lane & 15);
            if (xc < K) dW[(size_t)o * K + xc] = v;
        }
    }
}

// the four extra blocks (one per head) of a deferred level's RT_FINALIZE task (ReduceTask::up); sm: 4096 floats
__device__ __forceinline__ void adst_from_u_body(const ReduceTask& t, float* sm, int hh) {
    const int K = t.upK, UW = 4 * K + 4, tid = threadIdx.x;
    // column sums of this head's K columns of the partial rows (+ its S): 256 columns x 4 row groups, four loads in flight
    const int col = tid & 255, rg = tid >> 8;
    const float* src = t.up + (col < K ? hh * K + col : 4 * K + hh);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col <= K) {
        int r = rg;
        for (; r + 12 < t.n_up; r += 16) {
            a0 += src[(size_t)r * UW];        a1 += src[(size_t)(r + 4) * UW];
            a2 += src[(size_t)(r + 8) * UW];  a3 += src[(size_t)(r + 12) * UW];
        }
        for (; r < t.n_up; r += 4) a0 += src[(size_t)r * UW];
    }
    sm[rg * 256 + col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tid < 256) sm[1024 + tid] = (sm[tid] + sm[256 + tid]) + (sm[512 + tid] + sm[768 + tid]);       // U[hh][0..K), then S[hh] at K
    __syncthreads();
    const float* U = sm + 1024;
    const int c = hh * 32 + (tid >> 5), part = tid & 31;     // 32 columns of the head x 32 lanes
    float a = 0.f;
    for (int k = part; k < K; k += 32) a = fmaf(t.upW[(size_t)c * K + k], U[k], a);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (part == 0) t.o0[hh * t.att_w + t.dst_off + (c & 31)] = fmaf(t.upb[c], U[K], a);
}

__global__ __launch_bounds__(1024) void k_reduce_tasks(ReduceTasks T, AdamRide R) {
    __shared__ float sm[16 * 256];
    if (R.nblk && (int)blockIdx.x >= R.first) { adam_ride(R);  return; }
    int ti = 0;
    while (ti + 1 < T.n && (int)blockIdx.x >= T.first[ti + 1]) ++ti;
    const ReduceTask& t = T.t[ti];
    const int vb = (int)blockIdx.x - T.first[ti];
    if (t.kind == RT_FINALIZE) {
        if (vb < 2 * FN_D / 32) {
            if (t.up && vb < FN_D / 32) return;             // deferred form: dL/da_dst comes from the extra block below, not from partials
            const int col = vb * 32 + (threadIdx.x >> 5);
            const float v = colmajor_sum_32(t.p0 + (size_t)col * FN_MAX_PART, t.n0);
            if ((threadIdx.x & 31) == 0) {
                const int DH = FN_D / t.H, cc = col & 127, part = col >> 7;
                t.o0[(cc / DH) * t.att_w + (part ? t.src_off : t.dst_off) + (cc % DH)] = v;
            }
        } else if (vb == 2 * FN_D / 32 && t.et.mode == 2) {
            gat_finalize_body(2 * FN_D, sm, t.p0, t.n0, t.p1, t.n1, t.et, t.att, t.att_w, t.dst_off, t.src_off, t.o0, t.o1, t.o2, t.H);
        } else {
            adst_from_u_body(t, sm, vb - 2 * FN_D / 32 - (t.et.mode == 2 ? 1 : 0));
        }
    } else if (t.kind == RT_COLSUM) {
        const int col = vb * 32 + (threadIdx.x >> 5);
        const float v = colmajor_sum_32(t.p0 + (size_t)col * FN_MAX_PART, t.n0);
        if ((threadIdx.x & 31) == 0) t.o0[(col / FN_D) * t.ld + t.off + (col % FN_D)] = v;
    } else {
        switch (t.cls) {
            case 0: wgrad_reduce_strip<1, 1>(vb, sm, t.p0, t.n0, t.K, t.o0, t.o1); break;
            case 1: wgrad_reduce_strip<1, 2>(vb, sm, t.p0, t.n0, t.K, t.o0, t.o1); break;
            case 2: wgrad_reduce_strip<4, 2>(vb, sm, t.p0, t.n0, t.K, t.o0, t.o1); break;
            case 4: wgrad_reduce_strip<4, 2, true>(vb, sm, t.p0, t.n0, t.K, t.o0, t.o1); break;
            default: wgrad_reduce_strip<6, 2>(vb, sm, t.p0, t.n0, t.K, t.o0, t.o1); break;
        }
    }
}

// column sums of part [n_rows][cols]: columns < split go to out0, the rest to out1.  1024 threads = 32 columns x 32 row groups
__global__ __launch_bounds__(1024) void k_reduce_rows(const float* __restrict__ part, int n_rows, int64_t cols,
                                                      float* __restrict__ out0, float* __restrict__ out1, int64_t split) {
    __shared__ float red[32][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int64_t col = (int64_t)blockIdx.x * 32 + c;
    float acc = 0.f;
    if (col < cols)
        for (int r = rg; r < n_rows; r += 32) acc += part[(size_t)r * cols + col];
    red[rg][c] = acc;
    __syncthreads();
    if (threadIdx.x < 32 && col < cols) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) v += red[g][threadIdx.x];
        if (col < split) out0[col] = v;
        else out1[col - split] = v;
    }
}


// m: the level's edge count (a level without edges has an empty attribute table, whose pointer may be null: nothing reads it)
bool bad_edge_term(const fn_edge_term* et, int64_t m = 1) {      // (the other translation units: fni::bad_edge_term)
    if (!et) return true;
    if (et->mode == 0) return false;
    if (et->mode != 2) return true;
    return et->K < 1 || et->K > FN_MAX_EDGE_K || et->d_e < 1 || et->d_e > 128 || (!et->x_sorted && m > 0) || !et->embW || !et->embb;
}

}  // namespace

// =====================================================================================
// C-ABI
// =====================================================================================
namespace {
unsigned long long* g_mol_stamps = nullptr;     // fn_debug_set_stamps
int64_t g_mol_stamps_n = 0;
hipEvent_t g_prof_ev[4] = {nullptr, nullptr, nullptr, nullptr};      // fn_debug_set_profile_events: forward begin / end, backward begin / end
// records profiling event `i` (if set) on the stream: as an EXTERNAL event node while the stream is being captured, so that the
// time between two of them can be read after a replay of the graph
int prof_event(int i, hipStream_t st) {
    if (!g_prof_ev[i]) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError();  cs = hipStreamCaptureStatusNone; }
    hipError_t rc;
    if (cs != hipStreamCaptureStatusActive) rc = hipEventRecord(g_prof_ev[i], st);
    else {
        rc = hipEventRecordWithFlags(g_prof_ev[i], st, hipEventRecordExternal);
        if (rc != hipSuccess) {      // the same node by hand: an event-record node behind the stream's current capture dependencies
            (void)hipGetLastError();
            hipGraph_t graph = nullptr;
            const hipGraphNode_t* deps = nullptr;
            size_t n_deps = 0;
            unsigned long long id = 0;
            rc = hipStreamGetCaptureInfo_v2(st, &cs, &id, &graph, &deps, &n_deps);
            hipGraphNode_t node = nullptr;
            if (rc == hipSuccess) rc = hipGraphAddEventRecordNode(&node, graph, deps, n_deps, g_prof_ev[i]);
            if (rc == hipSuccess) rc = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
        }
    }
    if (rc != hipSuccess) {
        (void)hipGetLastError();
        static thread_local char msg[160];
        snprintf(msg, sizeof msg, "fn_debug_set_profile_events: recording the event failed (%s)", hipGetErrorString(rc));
        return fail(FN_EUNSUPPORTED, msg);
    }
    return 0;
}
int g_tune[FN_TUNE_COUNT] = {1024, 0, 0, 192, 0, 0, 0, 1, 1, 0, 1792, 1536, 512, 512, 2, -1, 0, 1, 0, 0, 1, 23, 1, 768, 1, 0, 1, 1, 0, 0, 6144, 1, 1, 1};   // in the order of the FN_TUNE_* keys
}  // namespace
namespace fni {      // hooks for the other translation units (fn_internal.h)
int fail(int code, const char* what) { return ::fail(code, what); }
int launch_status(const char* where) { return ::launch_status(where); }
int tune(int key) { return key >= 0 && key < FN_TUNE_COUNT ? g_tune[key] : 0; }
unsigned long long* stamps(int64_t* n_u64) { *n_u64 = g_mol_stamps_n;  return g_mol_stamps; }
bool bad_edge_term(const fn_edge_term* et, int64_t m) { return ::bad_edge_term(et, m); }
}  // namespace fni
namespace {
template <typename Kern> int allow_lds(Kern kern, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail((int)e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); }
    return 0;
}
// blocks of a product with `tiles` row tiles when every block walks `iters` of them (two column halves per tile)
inline int lin_iters(int64_t total_tiles) {     // row tiles per block so that the launch is resident at once (four blocks per CU)
    const int64_t slots = g_tune[FN_TUNE_GEMM_SLOTS];
    if (slots <= 0) return 1;                   // default: one output tile per workgroup (measured best, tools/probe/gemm_probe.hip)
    const int64_t it = (2 * total_tiles + slots - 1) / slots;
    return (int)(it < 1 ? 1 : it);
}
template <int KQ>
int launch_linear128(const float* X, int K, const float* Bt, const float* bias, float* Y, int64_t M, fn_act_epilogue mk,
                     NodeScalarEpi ns, hipStream_t st) {
    const size_t lds = (size_t)(4 * KQ * kLinLd) * sizeof(float);
    const int64_t tiles = (M + kLinRows - 1) / kLinRows;
    const int iters = lin_iters(tiles), grid = lin_blocks(tiles, iters);
    constexpr bool VEC = KQ % 4 == 0;
    if (VEC && K == 4 * KQ) {
        if (iters > 1) {
            if (int rc = allow_lds(k_linear128<KQ, VEC, VEC>, lds)) return rc;
            hipLaunchKernelGGL((k_linear128<KQ, VEC, VEC>), dim3(grid), dim3(kLinThreads), lds, st, X, K, Bt, bias, Y, M, mk, ns);
        } else {
            if (int rc = allow_lds(k_linear128<KQ, VEC, false>, lds)) return rc;
            hipLaunchKernelGGL((k_linear128<KQ, VEC, false>), dim3(grid), dim3(kLinThreads), lds, st, X, K, Bt, bias, Y, M, mk, ns);
        }
    } else {
        if (int rc = allow_lds(k_linear128<KQ, false, false>, lds)) return rc;
        hipLaunchKernelGGL((k_linear128<KQ, false, false>), dim3(grid), dim3(kLinThreads), lds, st, X, K, Bt, bias, Y, M, mk, ns);
    }
    return 0;
}
int linear128_group_impl(LinTasks& T, hipStream_t st) {
    constexpr int KQ = 32;
    const size_t lds = (size_t)(4 * KQ * kLinLd) * sizeof(float);
    int64_t total = 0;
    for (int i = 0; i < T.n; ++i) total += T.t[i].M > 0 ? (T.t[i].M + kLinRows - 1) / kLinRows : 0;
    const int iters = lin_iters(total);
    int blocks = 0, live = 0;
    for (int i = 0; i < T.n; ++i) {
        if (T.t[i].M <= 0) continue;
        LinTask t = T.t[i];
        t.first = blocks;
        t.nblk = lin_blocks((t.M + kLinRows - 1) / kLinRows, iters);
        blocks += t.nblk;
        T.t[live++] = t;
    }
    T.n = live;
    T.K = 128;
    if (!live) return 0;
    bool any_ra = false;
    for (int i = 0; i < T.n; ++i) any_ra |= T.t[i].ra.z != nullptr;
    if (any_ra) {                          // a RowAdd term must not get lost: the kernel variant that applies it (one tile per workgroup)
        blocks = 0;
        for (int i = 0; i < T.n; ++i) {
            T.t[i].first = blocks;
            T.t[i].nblk = lin_blocks((T.t[i].M + kLinRows - 1) / kLinRows, 1);
            blocks += T.t[i].nblk;
        }
        hipLaunchKernelGGL(k_linear128_multi_ra, dim3(blocks), dim3(kLinThreads), lds, st, T);
        return launch_status("grouped projection GEMM (+ row term)");
    }
    if (iters > 1) {
        if (int rc = allow_lds(k_linear128_multi<KQ, true, true>, lds)) return rc;
        hipLaunchKernelGGL((k_linear128_multi<KQ, true, true>), dim3(blocks), dim3(kLinThreads), lds, st, T);
    } else {
        if (int rc = allow_lds(k_linear128_multi<KQ, true, false>, lds)) return rc;
        hipLaunchKernelGGL((k_linear128_multi<KQ, true, false>), dim3(blocks), dim3(kLinThreads), lds, st, T);
    }
    return launch_status("grouped projection GEMM");
}

// layer 0: the bond (K = 17) and connection (K = 6) projections in one launch of the K <= 20 variant, each task with its own K
int launch_linear128_small_group(LinTasks& T, hipStream_t st) {
    constexpr int KQ = 5;
    bool mixed = false;                          // a task with 20 < K <= 168 rides along (k_linear128_layer0)
    for (int i = 0; i < T.n; ++i) mixed |= T.t[i].M > 0 && T.t[i].K > 4 * KQ;
    const size_t lds = (size_t)(4 * (mixed ? 44 : KQ) * kLinLd) * sizeof(float);
    int blocks = 0, live = 0;
    for (int i = 0; i < T.n; ++i) {
        if (T.t[i].M <= 0) continue;
        LinTask t = T.t[i];
        if (t.K < 1 || t.K > 168) return fail(FN_EINVAL, "layer-0 projection group: K must be 1..168");
        t.first = blocks;
        t.nblk = lin_blocks((t.M + kLinRows - 1) / kLinRows, 1);
        blocks += t.nblk;
        T.t[live++] = t;
    }
    T.n = live;
    T.K = 4 * KQ;
    if (!live) return 0;
    if (mixed) hipLaunchKernelGGL(k_linear128_layer0, dim3(blocks), dim3(kLinThreads), lds, st, T);
    else hipLaunchKernelGGL((k_linear128_multi<KQ, false, false>), dim3(blocks), dim3(kLinThreads), lds, st, T);
    return launch_status("grouped projection GEMM (layer 0)");
}

template <int CTW, int NH>
int launch_wgrad(const float* dY, const float* X, int K, int64_t M, int rpb, int grid, float* part, float* dW, float* db,
                 hipStream_t st) {
    constexpr int XW = 16 * CTW * NH, XLD = XW + 16, PW = 128 * XW + 128;
    const size_t lds = (size_t)2 * kWgChunk * (kBtLd + XLD) * sizeof(float);
    if (int rc = allow_lds(k_linear128_wgrad<CTW, NH>, lds)) return rc;
    hipLaunchKernelGGL((k_linear128_wgrad<CTW, NH>), dim3(grid), dim3(256 * NH), lds, st, dY, X, K, M, rpb, part);
    if (dW) hipLaunchKernelGGL((k_wgrad_reduce<CTW, NH>), dim3((PW + 31) / 32), dim3(1024), 0, st, part, grid, K, dW, db);
    return 0;
}
inline int64_t wgrad_part_width(int K) {
    const int xw = K <= 16 ? 16 : K <= 32 ? 32 : K <= 128 ? 128 : 192;
    return (int64_t)128 * xw + 128;
}
inline int wgrad_rows_per_block(int64_t M) {
    int64_t rpb = (M + 255) / 256;
    rpb = (rpb + kWgChunk - 1) / kWgChunk * kWgChunk;
    return (int)(rpb < kWgChunk ? kWgChunk : rpb);
}
}  // namespace

namespace {
__global__ void k_zero2_i32(int32_t* __restrict__ a, int64_t na, int32_t* __restrict__ b, int64_t nb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < na) a[i] = 0;
        else b[i - na] = 0;
    }
}
}  // namespace
namespace fni {
int launch_linear128_group(LinTasks& T, hipStream_t st) { return ::linear128_group_impl(T, st); }
}  // namespace fni

extern "C" {

int fn_abi_version(void) { return FN_ABI_VERSION; }

int fn_debug_set_stamps(void* buf, int64_t n_u64) {
    g_mol_stamps = static_cast<unsigned long long*>(buf);
    g_mol_stamps_n = buf ? n_u64 : 0;
    return 0;
}

int fn_debug_set_profile_events(void* const* events) {
    for (int i = 0; i < 4; ++i) g_prof_ev[i] = events ? static_cast<hipEvent_t>(events[i]) : nullptr;
    return 0;
}

int fn_set_tuning(int key, int value) {
    if (key < 0 || key >= FN_TUNE_COUNT) return fail(FN_EINVAL, "fn_set_tuning: unknown key");
    if (key == FN_TUNE_FWD_BLOCKS && value < 1) return fail(FN_EINVAL, "fn_set_tuning: block count must be positive");
    g_tune[key] = value;
    return 0;
}
const char* fn_last_error(void) { return tl_err; }



int fn_node_scalars_f32(const float* h, const float* att, int att_w, int dst_off, int src_off, float* s_dst,
                        float* s_src, int64_t n, int heads, fn_stream_t stream) {
    if (!h || !att || !s_dst || !s_src || n < 0) return fail(FN_EINVAL, "fn_node_scalars_f32: bad argument");
    if ((att_w | dst_off | src_off) & 3) return fail(FN_EINVAL, "fn_node_scalars_f32: att blocks must be 16-byte aligned");
    if (n == 0) return 0;
    FN_DISPATCH_H(heads, hipLaunchKernelGGL(k_node_scalars<HH>, dim3(row_grid(n, kGridCap)), dim3(kBlock), 0, S(stream),
                                            h, att, att_w, dst_off, src_off, s_dst, s_src, n));
    return launch_status("fn_node_scalars_f32");
}

// ---- argument validation + launch geometry of the three attention kernels (shared by the single-level C-ABI
// entry points and the engine's two-level launches); nblk == 0 means "nothing to do"


static int prep_gat_bwd_dst(const float* g_out, const float* h, const float* p_sorted, const fn_edge_term* et,
                            const fn_gat_plan* plan, float neg_slope, float* dz_sorted, float* g_s_orig, float* pz_src,
                            float* g_s_dst, float* part_e, int* n_part_e, int heads, GatBwdDstArgs* A) {
    if (!g_out || !h || !plan || !g_s_dst || !n_part_e || !et) return fail(FN_EINVAL, "fn_gat_bwd_dst_f32: bad argument");
    if (et->mode != 0 && bad_edge_term(et, plan ? plan->m : 1)) return fail(FN_EINVAL, "fn_gat_bwd_dst_f32: bad edge term");
    if (plan->m > 0 && (!p_sorted || !pz_src || !plan->spos_d || (et->mode == 0 && !dz_sorted && !g_s_orig)))
        return fail(FN_EINVAL, "fn_gat_bwd_dst_f32: null edge buffer");
    if (et->mode == 2 && !part_e) return fail(FN_EINVAL, "fn_gat_bwd_dst_f32: null part_e");
    if (heads != 1 && heads != 2 && heads != 4 && heads != 8) return fail(FN_EUNSUPPORTED, "heads must be 1, 2, 4 or 8 (128 = heads * head_dim)");
    *n_part_e = 0;
    *A = GatBwdDstArgs{g_out, h, p_sorted, *et, *plan, neg_slope, dz_sorted, g_s_orig, pz_src, g_s_dst, part_e, 1, 0};
    if (plan->n == 0) return 0;
    if (plan->n > (1 << 23) || plan->m * heads > (1 << 29))
        return fail(FN_EUNSUPPORTED, "fn_gat_bwd_dst_f32: level too large for 32-bit byte offsets (n <= 2^23 rows, m*heads <= 2^29)");
    // R rows per half-wave; one edge-parameter partial row per block, hence at most FN_MAX_PART blocks.
    // Unlike the forward kernel this one wants R SMALL: its (p, dz) stores are scattered 8-byte writes into source
    // order, vmcnt counts loads and stores in one in-order queue, so every extra row per wave waits behind the
    // previous row's slow stores (B=2048: R=4 56 us, R=8 84 us; B=512: 16 us at any R)
    const int64_t groups = (plan->n + kBwdRows - 1) / kBwdRows;
    // rows per half-wave: ~groups / 1536 (three at ESOL batch 512: 0.926 -> 0.902 ms per step against one row each), but at most
    // three -- the (p, dz) stores into source order are scattered and every further row waits behind them (B = 2048: 56 us at
    // four rows, 84 us at eight) -- unless the level is so large that the partial rows would not fit FN_MAX_PART
    const int64_t resident = g_tune[FN_TUNE_DST_BLOCKS] > 0 && g_tune[FN_TUNE_DST_BLOCKS] < FN_MAX_PART ? g_tune[FN_TUNE_DST_BLOCKS] : FN_MAX_PART;
    const int64_t need = (groups + FN_MAX_PART - 1) / FN_MAX_PART, want = (groups + resident - 1) / resident;
    A->rows_per_hw = (int)std::max<int64_t>(need, std::min<int64_t>(want, 3));
    A->nblk = (int)((plan->n + (int64_t)kBwdRows * A->rows_per_hw - 1) / ((int64_t)kBwdRows * A->rows_per_hw));
    *n_part_e = (et->mode == 2) ? A->nblk : 0;
    return 0;
}

static int launch_gat_bwd_dst(const GatBwdDstArgs& A, int heads, hipStream_t st) {
    if (A.nblk == 0) return 0;
    const int kl = edge_class(&A.et);
    FN_DISPATCH_H(heads, {
        if (kl == 0) hipLaunchKernelGGL((k_gat_bwd_dst<HH, 0, kBwdRows>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
        else if (kl == 1) hipLaunchKernelGGL((k_gat_bwd_dst<HH, 1, kBwdRows>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
        else hipLaunchKernelGGL((k_gat_bwd_dst<HH, FN_MAX_EDGE_K, kBwdRows>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
    });
    return launch_status("fn_gat_bwd_dst_f32");
}


int fn_gat_bwd_dst_f32(const float* g_out, const float* h, const float* p_sorted, const fn_edge_term* et,
                       const fn_gat_plan* plan, float neg_slope, float* dz_sorted, float* g_s_orig, float* pz_src,
                       float* g_s_dst, float* part_e, int* n_part_e, int heads, fn_stream_t stream) {
    GatBwdDstArgs A;
    if (int rc = prep_gat_bwd_dst(g_out, h, p_sorted, et, plan, neg_slope, dz_sorted, g_s_orig, pz_src, g_s_dst, part_e, n_part_e, heads, &A)) return rc;
    return launch_gat_bwd_dst(A, heads, S(stream));
}

static int prep_gat_bwd_src(const float* g_out, const float* h, const float* pz_src, const float* g_s_dst, const float* att,
                            int att_w, int dst_off, int src_off, const fn_gat_plan* plan, float* g_h, float* part_a,
                            int* n_part_a, int heads, GatBwdSrcArgs* A) {
    if (!g_out || !h || !g_s_dst || !att || !plan || !g_h || !part_a || !n_part_a) return fail(FN_EINVAL, "fn_gat_bwd_src_f32: bad argument");
    if (plan->m > 0 && !pz_src) return fail(FN_EINVAL, "fn_gat_bwd_src_f32: null edge buffer");
    if ((att_w | dst_off | src_off) & 3) return fail(FN_EINVAL, "fn_gat_bwd_src_f32: att blocks must be 16-byte aligned");
    if (heads != 1 && heads != 2 && heads != 4 && heads != 8) return fail(FN_EUNSUPPORTED, "heads must be 1, 2, 4 or 8 (128 = heads * head_dim)");
    *n_part_a = 0;
    *A = GatBwdSrcArgs{g_out, h, pz_src, g_s_dst, att, att_w, dst_off, src_off, *plan, g_h, part_a, 1, 0};
    if (plan->n == 0) return 0;
    if (plan->n > (1 << 23) || plan->m * heads > (1 << 28))
        return fail(FN_EUNSUPPORTED, "fn_gat_bwd_src_f32: level too large for 32-bit byte offsets (n <= 2^23 rows, m*heads <= 2^28)");
    // every block writes 256 partial sums column-major (scattered): at most 1024 blocks, each half-wave pipelining R rows
    const int64_t groups = (plan->n + kBwdRows - 1) / kBwdRows;
    int64_t resident = (int64_t)g_tune[FN_TUNE_SRC_BLOCKS];
    if (resident > 1024 || resident < 1) resident = 1024;
    A->rows_per_hw = (int)((groups + resident - 1) / resident);
    A->nblk = (int)((plan->n + (int64_t)kBwdRows * A->rows_per_hw - 1) / ((int64_t)kBwdRows * A->rows_per_hw));
    *n_part_a = A->nblk;
    return 0;
}

static int launch_gat_bwd_src(const GatBwdSrcArgs& A, int heads, hipStream_t st) {
    if (A.nblk == 0) return 0;
    FN_DISPATCH_H(heads, hipLaunchKernelGGL((k_gat_bwd_src<HH, kBwdRows>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A));
    return launch_status("fn_gat_bwd_src_f32");
}

int fn_gat_bwd_src_f32(const float* g_out, const float* h, const float* pz_src,
                       const float* g_s_dst, const float* att, int att_w, int dst_off, int src_off,
                       const fn_gat_plan* plan, float* g_h, float* part_a, int* n_part_a, int heads, fn_stream_t stream) {
    GatBwdSrcArgs A;
    if (int rc = prep_gat_bwd_src(g_out, h, pz_src, g_s_dst, att, att_w, dst_off, src_off, plan, g_h, part_a, n_part_a, heads, &A)) return rc;
    return launch_gat_bwd_src(A, heads, S(stream));
}

int fn_gat_bwd_finalize_f32(const float* part_a, int n_part_a, const float* part_e, int n_part_e, const fn_edge_term* et,
                            const float* att, int att_w, int dst_off, int src_off, float* g_att, float* g_embW,
                            float* g_embb, int heads, fn_stream_t stream) {
    if (!part_a || n_part_a < 0 || n_part_e < 0 || !att || !g_att || bad_edge_term(et, 0)) return fail(FN_EINVAL, "fn_gat_bwd_finalize_f32: bad argument");
    if (et->mode == 2 && (!part_e || !g_embW || !g_embb)) return fail(FN_EINVAL, "fn_gat_bwd_finalize_f32: null mode-2 buffer");
    if (heads != 1 && heads != 2 && heads != 4 && heads != 8) return fail(FN_EUNSUPPORTED, "heads must be 1, 2, 4 or 8");
    hipLaunchKernelGGL(k_gat_finalize, dim3(2 * FN_D + (et->mode == 2 ? 1 : 0)), dim3(1024), 0, S(stream), part_a, n_part_a, part_e, n_part_e, *et, att,
                       att_w, dst_off, src_off, g_att, g_embW, g_embb, heads);
    return launch_status("fn_gat_bwd_finalize_f32");
}

int fn_attn_by_src_f32(const float* p_sorted, const fn_gat_plan* plan, float* attn, int heads, fn_stream_t stream) {
    if (!plan || !attn || (plan->m > 0 && !p_sorted)) return fail(FN_EINVAL, "fn_attn_by_src_f32: bad argument");
    if (plan->n == 0) return 0;
    FN_DISPATCH_H(heads, hipLaunchKernelGGL(k_attn_by_src<HH>, dim3(flat_grid(plan->n * heads, kGridCap)), dim3(kBlock), 0,
                                            S(stream), p_sorted, *plan, attn));
    return launch_status("fn_attn_by_src_f32");
}

int fn_row_dots_sorted_f32(const float* feat, const float* A, int lda, int off, int J, const fn_gat_plan* plan,
                           float* s_sorted, fn_stream_t stream) {
    if (!A || !plan || J < 1 || J > 8 || ((lda | off) & 3)) return fail(FN_EINVAL, "fn_row_dots_sorted_f32: bad argument");
    if (plan->m == 0) return 0;
    if (!s_sorted || (plan->m_real > 0 && !feat)) return fail(FN_EINVAL, "fn_row_dots_sorted_f32: null buffer");
    hipLaunchKernelGGL(k_row_dots_sorted, dim3(row_grid(plan->m, kGridCap)), dim3(kBlock), 0, S(stream), feat, A, lda, off, J,
                       *plan, s_sorted);
    return launch_status("fn_row_dots_sorted_f32");
}

int fn_row_dots_sorted_bwd_f32(const float* g_s_sorted, const float* feat, const float* A, int lda, int off, int J,
                               const fn_gat_plan* plan, float* g_feat, float* part, int* n_part, fn_stream_t stream) {
    if (!A || !plan || !part || !n_part || J < 1 || J > 8 || ((lda | off) & 3))
        return fail(FN_EINVAL, "fn_row_dots_sorted_bwd_f32: bad argument");
    if (plan->m_real > 0 && (!g_s_sorted || !feat || !g_feat || !plan->inv_d))
        return fail(FN_EINVAL, "fn_row_dots_sorted_bwd_f32: null buffer");
    const int g = row_grid(plan->m_real, kRowDotsBwdBlocks);
    *n_part = g;
    hipLaunchKernelGGL(k_row_dots_sorted_bwd, dim3(g), dim3(kBlock), 0, S(stream),
                       RowDotsBwdArgs{g_s_sorted, feat, A, lda, off, J, *plan, g_feat, part, nullptr, 0, g});
    return launch_status("fn_row_dots_sorted_bwd_f32");
}

int fn_sort_edge_attr_f32(const float* x, int K, const fn_gat_plan* plan, float* x_sorted, fn_stream_t stream) {
    if (!plan || K < 1) return fail(FN_EINVAL, "fn_sort_edge_attr_f32: bad argument");
    if (plan->m == 0) return 0;
    if (!x_sorted || (plan->m_real > 0 && !x)) return fail(FN_EINVAL, "fn_sort_edge_attr_f32: null buffer");
    hipLaunchKernelGGL(k_sort_edge_attr, dim3(flat_grid(plan->m * K, kGridCap)), dim3(kBlock), 0, S(stream), x, K, *plan, x_sorted, 0);
    return launch_status("fn_sort_edge_attr_f32");
}
int fn_sort_edge_attr_src_f32(const float* x, int K, const fn_gat_plan* plan, float* x_src, fn_stream_t stream) {
    if (!plan || K < 1) return fail(FN_EINVAL, "fn_sort_edge_attr_src_f32: bad argument");
    if (plan->m == 0) return 0;
    if (!x_src || (plan->m_real > 0 && !x)) return fail(FN_EINVAL, "fn_sort_edge_attr_src_f32: null buffer");
    hipLaunchKernelGGL(k_sort_edge_attr, dim3(flat_grid(plan->m * K, kGridCap)), dim3(kBlock), 0, S(stream), x, K, *plan, x_src, 1);
    return launch_status("fn_sort_edge_attr_src_f32");
}

int fn_colsum_f32(const float* part, int n_rows, int cols, float* out, int ld, int off, fn_stream_t stream) {
    if (!part || !out || n_rows < 0 || n_rows > FN_MAX_PART || cols < 1) return fail(FN_EINVAL, "fn_colsum_f32: bad argument");
    hipLaunchKernelGGL(k_colsum, dim3(cols), dim3(1024), 0, S(stream), part, n_rows, cols, out, ld, off);
    return launch_status("fn_colsum_f32");
}

int fn_transpose_w_f32(const float* W, int K, float* Bt, fn_stream_t stream) {
    if (!W || !Bt || K < 1) return fail(FN_EINVAL, "fn_transpose_w_f32: bad argument");
    hipLaunchKernelGGL(k_transpose_w, dim3((K + 31) / 32, 4), dim3(256), 0, S(stream), W, K, Bt);
    return launch_status("fn_transpose_w_f32");
}

static int linear128_impl(const float* X, int K, const float* Bt, const float* bias, float* Y, int64_t M,
                          const fn_act_epilogue* act_bwd, NodeScalarEpi ns, fn_stream_t stream) {
    if (K < 1 || M < 0) return fail(FN_EINVAL, "fn_linear128_f32: bad argument");
    const fn_act_epilogue mk = act_bwd ? *act_bwd : fn_act_epilogue{nullptr, 0.f, 0, 0, 0, nullptr};
    if (M == 0) return 0;
    if (!X || !Bt || !Y || (((uintptr_t)X | (uintptr_t)Y) & 15)) return fail(FN_EINVAL, "fn_linear128_f32: null or misaligned buffer");
    int rc;
    if (K <= 8) rc = launch_linear128<2>(X, K, Bt, bias, Y, M, mk, ns, S(stream));
    else if (K <= 20) rc = launch_linear128<5>(X, K, Bt, bias, Y, M, mk, ns, S(stream));
    else if (K <= 128) rc = launch_linear128<32>(X, K, Bt, bias, Y, M, mk, ns, S(stream));
    else if (K <= 168) rc = launch_linear128<44>(X, K, Bt, bias, Y, M, mk, ns, S(stream));
    else return fail(FN_EUNSUPPORTED, "fn_linear128_f32: K > 168");
    if (rc) return rc;
    return launch_status("fn_linear128_f32");
}

int fn_linear128_f32(const float* X, int K, const float* Bt, const float* bias, float* Y, int64_t M,
                     const fn_act_epilogue* act_bwd, fn_stream_t stream) {
    return linear128_impl(X, K, Bt, bias, Y, M, act_bwd, NodeScalarEpi{nullptr, nullptr, nullptr, 0, 0, 0, 0}, stream);
}

int64_t fn_linear128_wgrad_ws(int64_t M, int K) {
    const int rpb = wgrad_rows_per_block(M);
    const int64_t grid = (M + rpb - 1) / rpb;
    return (grid < 1 ? 1 : grid) * wgrad_part_width(K);
}

int fn_linear128_wgrad_f32(const float* dY, const float* X, int K, int64_t M, float* ws, float* dW, float* db, fn_stream_t stream) {
    if (K < 1 || M < 0 || !dW || !db) return fail(FN_EINVAL, "fn_linear128_wgrad_f32: bad argument");
    if (M == 0) {
        hipLaunchKernelGGL(k_zero2_i32, dim3(flat_grid(128 * (K + 1), kGridCap)), dim3(kBlock), 0, S(stream),
                           reinterpret_cast<int32_t*>(dW), (int64_t)128 * K, reinterpret_cast<int32_t*>(db), (int64_t)128);
        return launch_status("fn_linear128_wgrad_f32");
    }
    if (!dY || !X || !ws || ((uintptr_t)dY & 15)) return fail(FN_EINVAL, "fn_linear128_wgrad_f32: null or misaligned buffer");
    const int rpb = wgrad_rows_per_block(M);
    const int grid = (int)((M + rpb - 1) / rpb);
    int rc;
    if (K <= 16) rc = launch_wgrad<1, 1>(dY, X, K, M, rpb, grid, ws, dW, db, S(stream));
    else if (K <= 32) rc = launch_wgrad<1, 2>(dY, X, K, M, rpb, grid, ws, dW, db, S(stream));
    else if (K <= 128) rc = launch_wgrad<4, 2>(dY, X, K, M, rpb, grid, ws, dW, db, S(stream));
    else if (K <= 192) rc = launch_wgrad<6, 2>(dY, X, K, M, rpb, grid, ws, dW, db, S(stream));
    else return fail(FN_EUNSUPPORTED, "fn_linear128_wgrad_f32: K > 192");
    if (rc) return rc;
    return launch_status("fn_linear128_wgrad_f32");
}

int fn_segment_sum_f32(const float* src, int64_t src_ld, const int32_t* rowptr, const int32_t* perm, int32_t pos_base,
                       float* out, int64_t n_seg, int64_t width, int64_t n_items, fn_stream_t stream) {
    if (!rowptr || !out || n_seg < 0 || width < 1 || src_ld < width) return fail(FN_EINVAL, "fn_segment_sum_f32: bad argument");
    if (n_seg == 0) return 0;
    if (!src || !perm) return fail(FN_EINVAL, "fn_segment_sum_f32: null src/perm");
    if (width == FN_D && (src_ld & 3) == 0 && (((uintptr_t)src | (uintptr_t)out) & 15) == 0) {
        if (n_items >= 4 * n_seg)
            hipLaunchKernelGGL(k_segment_sum128_wide, dim3((unsigned)(n_seg < 8 * kGridCap ? n_seg : 8 * kGridCap)), dim3(kBlock), 0,
                               S(stream), src, src_ld, rowptr, perm, pos_base, out, n_seg);
        else
            hipLaunchKernelGGL(k_segment_sum128, dim3(row_grid(n_seg, kGridCap)), dim3(kBlock), 0, S(stream), src, src_ld, rowptr,
                               perm, pos_base, out, n_seg);
    }
    else
        hipLaunchKernelGGL(k_segment_sum_any, dim3(flat_grid(n_seg * width, kGridCap)), dim3(kBlock), 0, S(stream), src, src_ld,
                           rowptr, perm, pos_base, out, n_seg, width);
    return launch_status("fn_segment_sum_f32");
}

int fn_gather_rows_f32(const float* table, const int64_t* index, float* out, int64_t rows, int64_t width, fn_stream_t stream) {
    if (rows < 0 || width < 1 || !out) return fail(FN_EINVAL, "fn_gather_rows_f32: bad argument");
    if (rows == 0) return 0;
    if (!table || !index) return fail(FN_EINVAL, "fn_gather_rows_f32: null table/index");
    if ((width & 3) == 0 && (((uintptr_t)table | (uintptr_t)out) & 15) == 0)
        hipLaunchKernelGGL(k_gather_rows4, dim3(flat_grid(rows * (width / 4), kGridCap)), dim3(kBlock), 0, S(stream), table, index,
                           out, rows, width / 4, (const float*)nullptr);
    else
        hipLaunchKernelGGL(k_gather_rows1, dim3(flat_grid(rows * width, kGridCap)), dim3(kBlock), 0, S(stream), table, index, out,
                           rows, width);
    return launch_status("fn_gather_rows_f32");
}

int fn_segment_softmax_f32(const float* logits, const int32_t* rowptr, const int32_t* perm, int32_t pos_base, float* probs,
                           int64_t n_seg, int64_t width, fn_stream_t stream) {
    if (!rowptr || n_seg < 0 || width < 1) return fail(FN_EINVAL, "fn_segment_softmax_f32: bad argument");
    if (n_seg == 0) return 0;
    if (!logits || !perm || !probs) return fail(FN_EINVAL, "fn_segment_softmax_f32: null buffer");
    hipLaunchKernelGGL(k_segment_softmax, dim3(flat_grid(n_seg * width, kGridCap)), dim3(kBlock), 0, S(stream), logits, rowptr,
                       perm, pos_base, probs, n_seg, width);
    return launch_status("fn_segment_softmax_f32");
}

int fn_segment_softmax_bwd_f32(const float* probs, const float* g_probs, const int32_t* rowptr, const int32_t* perm,
                               int32_t pos_base, float* g_logits, int64_t n_seg, int64_t width, fn_stream_t stream) {
    if (!rowptr || n_seg < 0 || width < 1) return fail(FN_EINVAL, "fn_segment_softmax_bwd_f32: bad argument");
    if (n_seg == 0) return 0;
    if (!probs || !g_probs || !perm || !g_logits) return fail(FN_EINVAL, "fn_segment_softmax_bwd_f32: null buffer");
    hipLaunchKernelGGL(k_segment_softmax_bwd, dim3(flat_grid(n_seg * width, kGridCap)), dim3(kBlock), 0, S(stream), probs,
                       g_probs, rowptr, perm, pos_base, g_logits, n_seg, width);
    return launch_status("fn_segment_softmax_bwd_f32");
}

int fn_dropout_act_f32(const float* x, float* y, int64_t numel, float p, uint64_t seed, uint64_t offset,
                       const uint64_t* offset_dev, int relu, fn_stream_t stream) {
    if (numel < 0 || p < 0.f || p > 1.f) return fail(FN_EINVAL, "fn_dropout_act_f32: bad argument");
    if (numel == 0) return 0;
    if (!x || !y || (((uintptr_t)x | (uintptr_t)y) & 15)) return fail(FN_EINVAL, "fn_dropout_act_f32: null or misaligned buffer");
    hipLaunchKernelGGL(k_dropout_act<false>, dim3(flat_grid((numel + 3) / 4, kGridCap)), dim3(kBlock), 0, S(stream), x,
                       (const float*)nullptr, y, numel, p, seed, offset, offset_dev, relu);
    return launch_status("fn_dropout_act_f32");
}

int fn_dropout_act_bwd_f32(const float* g_y, const float* y, float* g_x, int64_t numel, float p, uint64_t seed,
                           uint64_t offset, const uint64_t* offset_dev, int relu, fn_stream_t stream) {
    if (numel < 0 || p < 0.f || p > 1.f) return fail(FN_EINVAL, "fn_dropout_act_bwd_f32: bad argument");
    if (numel == 0) return 0;
    if (!g_y || !g_x || (relu && !y) || (((uintptr_t)g_y | (uintptr_t)g_x | (uintptr_t)y) & 15))
        return fail(FN_EINVAL, "fn_dropout_act_bwd_f32: null or misaligned buffer");
    hipLaunchKernelGGL(k_dropout_act<true>, dim3(flat_grid((numel + 3) / 4, kGridCap)), dim3(kBlock), 0, S(stream), g_y, y, g_x,
                       numel, p, seed, offset, offset_dev, relu);
    return launch_status("fn_dropout_act_bwd_f32");
}


namespace {
constexpr int64_t kTallRows = 2048, kTallChunk = 256;      // inputs taller than kTallRows are reduced in chunks of kTallChunk rows
inline int64_t tall_chunks(int64_t rows) { return rows > kTallRows ? (rows + kTallChunk - 1) / kTallChunk : 1; }
}  // namespace

int64_t fn_gate_colsum_ws(int64_t rows, int64_t cols) {
    const int64_t ch = tall_chunks(rows);
    return ch > 1 ? ch * cols : 0;
}

int fn_gate_colsum_f32(const float* g_y, const float* y, float* g_x, float* colsum, int64_t rows, int64_t cols, float scale,
                       float* ws, fn_stream_t stream) {
    if (rows < 0 || cols < 0 || (cols & 3) || cols > INT32_MAX) return fail(FN_EINVAL, "fn_gate_colsum_f32: cols must be a multiple of 4");
    if (cols == 0) return 0;
    if (!colsum || (rows > 0 && (!g_y || !y || !g_x)) || (((uintptr_t)g_y | (uintptr_t)y | (uintptr_t)g_x | (uintptr_t)colsum | (uintptr_t)ws) & 15))
        return fail(FN_EINVAL, "fn_gate_colsum_f32: null or misaligned buffer");
    const int64_t ch = tall_chunks(rows);
    if (ch > 1 && !ws) return fail(FN_EINVAL, "fn_gate_colsum_f32: rows > 2048 need the fn_gate_colsum_ws() workspace");
    hipLaunchKernelGGL(k_gate_colsum, dim3((unsigned)((cols + 31) / 32), (unsigned)ch), dim3(256), 0, S(stream), g_y, y, g_x,
                       ch > 1 ? ws : colsum, rows, (int)cols, scale, ch > 1 ? kTallChunk : (rows > 0 ? rows : 1));
    if (ch > 1) hipLaunchKernelGGL(k_sum_chunks, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, S(stream), ws, (int)ch, cols, colsum);
    return launch_status("fn_gate_colsum_f32");
}

int fn_small_linear_f32(const float* x, const float* w, const float* b, float* y, int64_t M, int64_t K, int64_t C, int64_t M_out,
                        fn_stream_t stream) {
    if (M < 0 || K < 4 || (K & 3) || K > INT32_MAX || C < 1 || C > FN_SMALL_LINEAR_MAX)
        return fail(FN_EINVAL, "fn_small_linear_f32: K must be a multiple of 4 and 1 <= C <= FN_SMALL_LINEAR_MAX");
    if (M_out < M) M_out = M;
    if (M_out == 0) return 0;
    if (!x || !w || !y || (((uintptr_t)x | (uintptr_t)w) & 15)) return fail(FN_EINVAL, "fn_small_linear_f32: null or misaligned buffer");
    hipLaunchKernelGGL(k_small_linear, dim3((unsigned)((M_out + 3) / 4)), dim3(256), 0, S(stream), x, w, b, y, M, (int)K, (int)C, M_out);
    return launch_status("fn_small_linear_f32");
}

int64_t fn_small_linear_bwd_ws(int64_t M, int64_t K, int64_t C) {
    const int64_t ch = tall_chunks(M);
    return ch > 1 ? ch * C * (K + 1) : 0;
}

int fn_small_linear_bwd_f32(const float* g, const float* x, const float* w, float* g_x, float* dW, float* db, int64_t M, int64_t K,
                            int64_t C, float gate_scale, float* ws, fn_stream_t stream) {
    if (M < 0 || K < 4 || (K & 3) || K > INT32_MAX || C < 1 || C > FN_SMALL_LINEAR_MAX)
        return fail(FN_EINVAL, "fn_small_linear_bwd_f32: K must be a multiple of 4 and 1 <= C <= FN_SMALL_LINEAR_MAX");
    if (!w || !dW || !db || (M > 0 && (!g || !x || !g_x)) || (((uintptr_t)x | (uintptr_t)w | (uintptr_t)g_x | (uintptr_t)dW | (uintptr_t)ws) & 15))
        return fail(FN_EINVAL, "fn_small_linear_bwd_f32: null or misaligned buffer");
    const int64_t ch = tall_chunks(M);
    if (ch > 1 && !ws) return fail(FN_EINVAL, "fn_small_linear_bwd_f32: M > 2048 needs the fn_small_linear_bwd_ws() workspace");
    float* dW_o = ch > 1 ? ws : dW;
    float* db_o = ch > 1 ? ws + ch * C * K : db;
    const int64_t rpc = ch > 1 ? kTallChunk : (M > 0 ? M : 1);
    const dim3 grid((unsigned)((K + 15) / 16), (unsigned)ch);
    if (C <= 1) hipLaunchKernelGGL(k_small_linear_bwd<1>, grid, dim3(256), 0, S(stream), g, x, w, g_x, dW_o, db_o, M, (int)K, (int)C, rpc, gate_scale);
    else if (C <= 4) hipLaunchKernelGGL(k_small_linear_bwd<4>, grid, dim3(256), 0, S(stream), g, x, w, g_x, dW_o, db_o, M, (int)K, (int)C, rpc, gate_scale);
    else hipLaunchKernelGGL(k_small_linear_bwd<FN_SMALL_LINEAR_MAX>, grid, dim3(256), 0, S(stream), g, x, w, g_x, dW_o, db_o, M, (int)K, (int)C, rpc, gate_scale);
    if (ch > 1) {
        hipLaunchKernelGGL(k_sum_chunks, dim3((unsigned)((C * K + 15) / 16)), dim3(256), 0, S(stream), ws, (int)ch, C * K, dW);
        hipLaunchKernelGGL(k_sum_chunks, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, S(stream), ws + ch * C * K, (int)ch, C, db);
    }
    return launch_status("fn_small_linear_bwd_f32");
}

int64_t fn_small_linear_loss_ws(int64_t M_out) { return M_out > 0 ? (M_out + 3) / 4 : 0; }

int fn_small_linear_loss_f32(const float* x, const float* w, const float* b, const float* target, const float* row_w, int kind,
                             float* y, float* g, float* g_x, float gate_scale, float* loss_part, int64_t M, int64_t K, int64_t C,
                             int64_t M_out, fn_stream_t stream) {
    if (M < 0 || K < 4 || (K & 3) || K > FN_SMALL_LINEAR_LOSS_MAX_K || C < 1 || C > FN_SMALL_LINEAR_MAX || (kind != FN_LOSS_MSE && kind != FN_LOSS_BCE))
        return fail(FN_EINVAL, "fn_small_linear_loss_f32: K a multiple of 4 and <= FN_SMALL_LINEAR_LOSS_MAX_K, 1 <= C <= FN_SMALL_LINEAR_MAX, kind MSE or BCE");
    if (M_out < M) M_out = M;
    if (M_out == 0) return 0;
    if (!w || !target || !row_w || !y || !loss_part || gate_scale < 0.f || (M > 0 && (!x || !g || !g_x)) ||
        (((uintptr_t)x | (uintptr_t)w | (uintptr_t)g_x) & 15))
        return fail(FN_EINVAL, "fn_small_linear_loss_f32: null or misaligned buffer");
    const dim3 grid((unsigned)fn_small_linear_loss_ws(M_out));
    if (C <= 1) hipLaunchKernelGGL(k_small_linear_loss<1>, grid, dim3(256), 0, S(stream), x, w, b, target, row_w, kind, y, g, g_x, gate_scale, loss_part, M, (int)K, (int)C, M_out);
    else if (C <= 4) hipLaunchKernelGGL(k_small_linear_loss<4>, grid, dim3(256), 0, S(stream), x, w, b, target, row_w, kind, y, g, g_x, gate_scale, loss_part, M, (int)K, (int)C, M_out);
    else hipLaunchKernelGGL(k_small_linear_loss<FN_SMALL_LINEAR_MAX>, grid, dim3(256), 0, S(stream), x, w, b, target, row_w, kind, y, g, g_x, gate_scale, loss_part, M, (int)K, (int)C, M_out);
    return launch_status("fn_small_linear_loss_f32");
}

// (static, not an anonymous namespace: inside this extern "C" block clang gives a namespace-scope function C linkage and exports it)
static bool dense_shape_ok(int64_t M, int64_t K, int64_t N) {
    return M >= 0 && M <= FN_DENSE_MAX_ROWS && K >= 4 && N >= 4 && !(K & 3) && !(N & 3) && K <= 65536 && N <= 65536;
}
static int dense_tiles(int64_t n, int t) { return (int)((n + t - 1) / t); }

int fn_dense_fwd_f32(const float* X, const float* W, const float* bias, float* Y, int64_t M, int64_t K, int64_t N,
                     const fn_act_epilogue* act, fn_stream_t stream) {
    if (!dense_shape_ok(M, K, N)) return fail(FN_EINVAL, "fn_dense_fwd_f32: K and N must be multiples of 4, M <= FN_DENSE_MAX_ROWS");
    if (M == 0) return 0;
    if (!X || !W || !Y || (((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y | (uintptr_t)bias) & 15))
        return fail(FN_EINVAL, "fn_dense_fwd_f32: null or misaligned buffer");
    if (act && (act->p < 0.f || act->p > 1.f)) return fail(FN_EINVAL, "fn_dense_fwd_f32: dropout probability outside [0, 1]");
    DenseArgs T{};
    T.A = X;  T.Bsrc = W;  T.bias = bias;  T.OUT = Y;
    T.I = (int)M;  T.J = (int)N;  T.R = (int)K;  T.lda = (int)K;  T.ldb = (int)K;
    if (act) { T.act = *act;  T.act.y = Y; }
    // tall inputs: workgroup-shared 64 x 128 operand tiles (dense_head.inc, k_dense_fwd_tiles).  32-bit element offsets as elsewhere
    if (g_tune[FN_TUNE_DENSE_TILES] != 0 && K % kDtK == 0 && dense_tiles(M, kDtM) * dense_tiles(N, kDtN) >= kDtMinTiles) {
        T.tiles_i = dense_tiles(M, kDtM);  T.tiles_j = dense_tiles(N, kDtN);
        if (int rc = allow_lds(k_dense_fwd_tiles, kDtLdsBytes)) return rc;
        hipLaunchKernelGGL(k_dense_fwd_tiles, dim3((unsigned)(8 * ((T.tiles_i * T.tiles_j + 7) / 8))), dim3(kDtThreads), kDtLdsBytes, S(stream), T);
        return launch_status("fn_dense_fwd_f32 (workgroup-shared tiles)");
    }
    T.tiles_i = dense_tiles(M, 32);  T.tiles_j = dense_tiles(N, kDnCols);
    const int narrow = T.tiles_i * T.tiles_j < 192;      // too few 32 x 64 tiles to occupy the chip: 32 x 32
    if (narrow) T.tiles_j = dense_tiles(N, 32);
    hipLaunchKernelGGL(k_dense_fwd, dim3((unsigned)(T.tiles_i * T.tiles_j)), dim3(kDnThreads), kDnLdsBytes, S(stream), T, narrow);
    return launch_status("fn_dense_fwd_f32");
}

int fn_dense_bwd_f32(const float* g_y, const float* X, const float* W, float* g_x, float gate_scale, float* dW, float* db,
                     int64_t M, int64_t K, int64_t N, int64_t M_out, fn_stream_t stream) {
    return fn_dense_bwd_tail_f32(g_y, X, W, g_x, gate_scale, dW, db, M, K, N, M_out, nullptr, stream);
}

int fn_dense_bwd_tail_f32(const float* g_y, const float* X, const float* W, float* g_x, float gate_scale, float* dW, float* db,
                          int64_t M, int64_t K, int64_t N, int64_t M_out, const fn_small_dw* tail, fn_stream_t stream) {
    if (!dense_shape_ok(M, K, N)) return fail(FN_EINVAL, "fn_dense_bwd_f32: K and N must be multiples of 4, M <= FN_DENSE_MAX_ROWS");
    SmallDw sd{};
    sd.first_block = -1;
    if (tail) {
        if (tail->M < 0 || tail->M > FN_DENSE_MAX_ROWS || tail->K < 4 || (tail->K & 3) || tail->K > 65536 || tail->C < 1 ||
            tail->C > FN_SMALL_LINEAR_MAX || tail->n_part < 0 || !tail->dW || !tail->db || (tail->M > 0 && (!tail->g || !tail->x)) ||
            (tail->loss && tail->n_part > 0 && !tail->loss_part) || (((uintptr_t)tail->x | (uintptr_t)tail->dW) & 15))
            return fail(FN_EINVAL, "fn_dense_bwd_tail_f32: bad tail (M <= FN_DENSE_MAX_ROWS, K % 4 == 0, 1 <= C <= FN_SMALL_LINEAR_MAX)");
        sd.g = tail->g;  sd.x = tail->x;  sd.dW = tail->dW;  sd.db = tail->db;  sd.loss_part = tail->loss_part;  sd.loss = tail->loss;
        sd.n_part = (int)tail->n_part;  sd.M = (int)tail->M;  sd.K = (int)tail->K;  sd.C = (int)tail->C;
    }
    if (!W || !dW || (M > 0 && (!g_y || !X)) || gate_scale < 0.f ||
        (((uintptr_t)g_y | (uintptr_t)X | (uintptr_t)W | (uintptr_t)g_x | (uintptr_t)dW) & 15))
        return fail(FN_EINVAL, "fn_dense_bwd_f32: null or misaligned buffer");
    DensePair P{};
    DenseArgs& a = P.a;                                  // dW [N,K] = gy^T X, db = column sums of gy
    a.A = g_y;  a.Bsrc = X;  a.OUT = dW;  a.db = db;
    a.I = (int)N;  a.J = (int)K;  a.R = (int)M;  a.lda = (int)N;  a.ldb = (int)K;
    a.tiles_i = dense_tiles(N, 64);  a.tiles_j = dense_tiles(K, kDnCols);
    if (a.tiles_i * a.tiles_j < 192) {                   // too few 64 x 64 tiles to occupy the chip: 64 x 32 tiles (dense_dw_narrow)
        P.a_narrow = 1;
        a.tiles_j = dense_tiles(K, 32);
    }
    int blocks = a.tiles_i * a.tiles_j;
    P.b.first_block = blocks;
    if (M_out < M) M_out = M;
    if (M_out > FN_DENSE_MAX_ROWS) return fail(FN_EINVAL, "fn_dense_bwd_f32: M_out > FN_DENSE_MAX_ROWS");
    if (g_x && M_out > 0 && M == 0) {                    // no input rows: the padding rows of g_x are all there is, and they are zero
        hipLaunchKernelGGL(k_zero2_i32, dim3(flat_grid(M_out * K, kGridCap)), dim3(kBlock), 0, S(stream),
                           reinterpret_cast<int32_t*>(g_x), M_out * K, static_cast<int32_t*>(nullptr), (int64_t)0);
        if (int rc = launch_status("fn_dense_bwd_f32 (empty input)")) return rc;
    } else if (g_x && M_out > 0) {
        DenseArgs& b = P.b;                              // gX [M,K] = gy W, gated by X > 0
        b.A = g_y;  b.Bsrc = W;  b.OUT = g_x;  b.Z = gate_scale > 0.f ? X : nullptr;  b.gate_scale = gate_scale;
        b.I = (int)M;  b.I_out = (int)M_out;  b.J = (int)K;  b.R = (int)N;  b.lda = (int)N;  b.ldb = (int)K;
        b.tiles_i = dense_tiles(M_out, 32);  b.tiles_j = dense_tiles(K, kDnCols);
        if (b.tiles_i * b.tiles_j < 192) { P.b_narrow = 1;  b.tiles_j = dense_tiles(K, 32); }
        blocks += b.tiles_i * b.tiles_j;
    }
    if (tail) {
        sd.first_block = blocks;
        blocks += (sd.K + 15) / 16 + 1;
    }
    hipLaunchKernelGGL(k_dense_bwd, dim3((unsigned)blocks), dim3(kDnThreads), kDnLdsBytes, S(stream), P, sd);
    return launch_status("fn_dense_bwd_f32");
}

int fn_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                float weight_decay, int64_t step, fn_stream_t stream) {
    if (n < 0 || step < 1) return fail(FN_EINVAL, "fn_adam_f32: bad argument");
    if (n == 0) return 0;
    if (!p || !g || !m || !v || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15))
        return fail(FN_EINVAL, "fn_adam_f32: null or misaligned buffer");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(k_adam, dim3(flat_grid((n + 3) / 4, kGridCap)), dim3(kBlock), 0, S(stream), p, g, m, v, n,
                       (float)((double)lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), weight_decay,
                       (const int64_t*)nullptr, (const float*)nullptr);
    return launch_status("fn_adam_f32");
}

int fn_adam_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2,
                    float eps, float weight_decay, const int64_t* step_dev, fn_stream_t stream) {
    if (n < 0 || !lr_dev || !step_dev) return fail(FN_EINVAL, "fn_adam_dev_f32: bad argument");
    if (n == 0) return 0;
    if (!p || !g || !m || !v || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15))
        return fail(FN_EINVAL, "fn_adam_dev_f32: null or misaligned buffer");
    hipLaunchKernelGGL(k_adam, dim3(flat_grid((n + 3) / 4, kGridCap)), dim3(kBlock), 0, S(stream), p, g, m, v, n, 0.f, beta1, beta2,
                       eps, 0.f, weight_decay, step_dev, lr_dev);
    return launch_status("fn_adam_dev_f32");
}

int fn_edge_concat_f32(const float* x, const float* e_attr, const int64_t* edge_index, float* out, int64_t E,
                       fn_stream_t stream) {
    if (E < 0) return fail(FN_EINVAL, "fn_edge_concat_f32: bad argument");
    if (E == 0) return 0;
    if (!x || !e_attr || !edge_index || !out) return fail(FN_EINVAL, "fn_edge_concat_f32: null buffer");
    hipLaunchKernelGGL(k_edge_concat, dim3(flat_grid(E * 96, kGridCap)), dim3(kBlock), 0, S(stream), x, e_attr, edge_index, out, E);
    return launch_status("fn_edge_concat_f32");
}

int fn_pool_cat_f32(const float* x_atoms, const float* x_frags, const fn_seg_plan* mol_atoms, const fn_seg_plan* mol_frags,
                    float* out, fn_stream_t stream) {
    if (!mol_atoms || !mol_frags || !out || mol_atoms->n_seg != mol_frags->n_seg) return fail(FN_EINVAL, "fn_pool_cat_f32: bad argument");
    if (mol_atoms->n_seg == 0) return 0;
    if ((mol_atoms->n_items > 0 && !x_atoms) || (mol_frags->n_items > 0 && !x_frags) || !mol_atoms->rowptr || !mol_frags->rowptr)
        return fail(FN_EINVAL, "fn_pool_cat_f32: null buffer");
    hipLaunchKernelGGL(k_pool_cat, dim3((unsigned)mol_atoms->n_seg, 2), dim3(kBlock), 0, S(stream), x_atoms, x_frags, *mol_atoms,
                       *mol_frags, out);
    return launch_status("fn_pool_cat_f32");
}

int fn_pool_cat_bwd_f32(const float* g, const int64_t* batch, const int64_t* frag_batch, float* g_atoms, float* g_frags,
                        int64_t N, int64_t F, fn_stream_t stream) {
    if (N < 0 || F < 0) return fail(FN_EINVAL, "fn_pool_cat_bwd_f32: bad argument");
    if (N + F == 0) return 0;
    if (!g || (N && (!batch || !g_atoms)) || (F && (!frag_batch || !g_frags))) return fail(FN_EINVAL, "fn_pool_cat_bwd_f32: null buffer");
    hipLaunchKernelGGL(k_pool_cat_bwd, dim3(flat_grid((N + F) * 32, kGridCap)), dim3(kBlock), 0, S(stream), g, batch, frag_batch,
                       g_atoms, g_frags, N, F);
    return launch_status("fn_pool_cat_bwd_f32");
}

int fn_masked_mse_f32(const float* out, const float* y, const float* w, int64_t B, int T, float* loss, float* g_out,
                      fn_stream_t stream) {
    if (!out || !y || !w || !loss || !g_out || B < 1 || T < 1) return fail(FN_EINVAL, "fn_masked_mse_f32: bad argument");
    hipLaunchKernelGGL(k_masked_mse, dim3(1), dim3(1024), 0, S(stream), out, y, w, B, T, loss, g_out);
    return launch_status("fn_masked_mse_f32");
}

int fn_masked_bce_f32(const float* out, const float* y, const float* w, int64_t B, int T, float* loss, float* g_out,
                      fn_stream_t stream) {
    if (!out || !y || !w || !loss || !g_out || B < 1 || T < 1) return fail(FN_EINVAL, "fn_masked_bce_f32: bad argument");
    hipLaunchKernelGGL(k_masked_bce, dim3(1), dim3(1024), 0, S(stream), out, y, w, B, T, loss, g_out);
    return launch_status("fn_masked_bce_f32");
}

int64_t fn_masked_mse_multi_ws(int n_tasks) { return n_tasks > 0 ? (int64_t)n_tasks * kMseBlocks * 2 : 0; }

int fn_masked_mse_multi_f32(const fn_mse_task* tasks, int n_tasks, const float* scale_dev, float* ws, float* loss, fn_stream_t stream) {
    if (!tasks || n_tasks < 1 || n_tasks > 4 || !ws || !loss) return fail(FN_EINVAL, "fn_masked_mse_multi_f32: bad argument");
    MseTasks M{};
    M.n = n_tasks;
    M.scale_dev = scale_dev;
    for (int k = 0; k < n_tasks; ++k) {
        const fn_mse_task& t = tasks[k];
        if (!t.out || !t.y || !t.w || !t.g_out || t.B < 1 || t.T < 1 || (t.scale_idx >= 0 && !scale_dev))
            return fail(FN_EINVAL, "fn_masked_mse_multi_f32: bad task");
        M.t[k] = MseTask{t.out, t.y, t.w, t.g_out, t.B, t.T, t.scale_idx, t.coef};
    }
    hipLaunchKernelGGL(k_mse_multi_partial, dim3(kMseBlocks, n_tasks), dim3(256), 0, S(stream), M, ws);
    hipLaunchKernelGGL(k_mse_multi_finish, dim3(kMseBlocks, n_tasks), dim3(256), 0, S(stream), M, ws, loss);
    return launch_status("fn_masked_mse_multi_f32");
}



}  // extern "C"

#include "encoder.inc"
