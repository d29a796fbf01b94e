// Molecule-resident single-pass backward of an attention level (autograd of reference model/gat/gat2.py:146-169,
// 196-224, 250-272, 286-316: scatter_softmax + weighted scatter_add of one level).
// STATUS: parity-green, measured SLOWER than the two passes it was built to replace (39-42 us against 28 us, bond level at
// ESOL batch 512) and therefore off by default (FN_TUNE_BWD_MOL); HISTORY.md section 4c and profiles/r03_molbwd_phases.md have the
// phase timings and the reasons.
//
// The graphs of a collated batch are block-diagonal per molecule (dataset/data.py:877-948), so a workgroup that owns whole
// molecules owns every edge that touches their rows.  It stages the molecule's GRADIENT rows g[t] in LDS by LDS-DMA (coalesced,
// once, no registers; 64-row tile rounds for molecules beyond the tile) together with the per-edge state (probabilities, both
// CSR orders packed into one word per edge, raw attribute or original edge id), keeps the projected rows h[s] of the rows it
// owns in registers, and then needs no further global read:
//   pass A  (source-owner half-waves)    dp_e = <h[s], g[t_e]> per head  and  acc_s = sum_e p_e g[t_e]  -- ONE LDS row read per edge
//   pass B  (half-wave per destination row, lane j of a head owns in-edges 2j, 2j+1, DPP sums)
//                                       c_t = sum_e p_e dp_e,  dz_e = p_e (dp_e - c_t) LeakyReLU'_e,  g_s_dst[t] = sum dz_e,
//                                       and the edge outputs: dz in original edge order (mode 0) or the lane's share of
//                                       sum_e dz_e (x_e, 1) (mode 2)
//   pass C  (source-owner half-waves)    g_s_src[s] = sum_e dz_e,  g_h[s] = acc_s + g_s_dst[s] a_dst + g_s_src[s] a_src -> global,
//                                       dL/da partial columns
// g and h are read once, (p, dz) never leave the CU, g_h / dz are written once.  The two-pass kernels (k_gat_bwd_dst +
// k_gat_bwd_src, fragnet_hip.hip) read g and h twice and exchange (p, dz) through HBM.
//
// Units that do not fit the size class of the launch (rows > R or edges > M) run the same three passes with their rows
// gathered from global memory and the edge state in a caller-provided scratch, in extra workgroups of the launch -- correct,
// slow, rare.  Rows behind the last real molecule of a padded batch (fn_stage_padded) are zeroed by extra workgroups.
// Everything is fixed-order: no float atomics, bitwise reproducible.
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

#include "fn_internal.h"

namespace {
using fni::MolBwdLevel;
using fni::MolExt;

struct MolBwdArgs {
    MolBwdLevel lv[fni::kMolBwdMaxLevels];
    int n_lv, n_zero, force_slow;
    int64_t n_mols;
    const MolExt* ext;
    const int32_t* counts_dev;
    int32_t* status;
    unsigned long long* stamps;      // nullable profiling aid (fn_debug_set_stamps): kStamps s_memtime values per workgroup
};
constexpr int kStamps = 16;
struct Stamp {
    unsigned long long* p;
    int i;
    __device__ __forceinline__ void hit() {
        if (p) {
            if (threadIdx.x == 0 && i < kStamps) p[i] = __builtin_amdgcn_s_memtime();
            ++i;
        }
    }
};

__device__ __forceinline__ void lv_extent(const MolExt& x, int which, int& r0, int& nr, int& e0, int& me) {
    if (which == fni::LV_BOND) { r0 = x.b0;  nr = x.nb;  e0 = x.eb0;  me = x.meb; }
    else if (which == fni::LV_ATOM) { r0 = x.a0;  nr = x.na;  e0 = x.ea0;  me = x.mea; }
    else if (which == fni::LV_FBOND) { r0 = x.c0;  nr = x.nc;  e0 = x.ef0;  me = x.mef; }
    else { r0 = x.f0;  nr = x.nf;  e0 = x.ec0;  me = x.mec; }
}

template <int H, int NT, int RPH, int RT, int MCAP>
struct Cfg {
    static constexpr int NHW = NT / 32;            // half-waves = rows in flight
    static constexpr int R = NHW * RPH;            // rows of a unit (their h rows and accumulators live in registers)
    static constexpr int HS = MCAP + 16;           // head stride of the per-edge arrays (heads land 16 banks apart)
    static constexpr int EI = (MCAP + NT - 1) / NT;
    static constexpr int TI = RT * 32 / NT;        // 16-byte tile pieces per thread and round
    static constexpr int NW = NT / 64;
    static constexpr int PE = H * (FN_MAX_EDGE_K + 1);
    // floats: tile (RT gradient rows of a round) | p | z | idx | x | rs | rd | gsd | red | att | accA
    static constexpr int oP = RT * FN_D, oZ = oP + H * HS, oI = oZ + H * HS, oX = oI + MCAP, oRS = oX + MCAP, oRD = oRS + R + 4,
                         oG = oRD + R + 4, oRed = oG + R * H, oAtt = oRed + NHW * PE, oAcc = oAtt + 2 * FN_D, total = oAcc + 2 * FN_D;
    static constexpr size_t lds_bytes = (size_t)total * 4;
    static_assert(MCAP % 64 == 0, "a wave's LDS-DMA piece of a per-edge array never runs past the array");
    static_assert(RT * 32 % NT == 0 && RT % NHW == 0, "tile rounds are whole passes of the workgroup");
    static_assert(NHW * 2 * FN_D <= RT * FN_D, "the parameter-gradient partials reuse the tile");
};

template <int H, int NT, int RPH, int RT, int MCAP>
struct Smem {
    using C = Cfg<H, NT, RPH, RT, MCAP>;
    float* tile;  float* p;  float* z;  uint32_t* idx;  float* x;  int* rs;  int* rd;  float* gsd;
    float* red;      // [NHW][PE]   running sums of dz (x, 1) per half-wave and head (mode 2)
    float* att;      // [2][128]    destination | source block of the attention vector, in lane order
    float* accA;     // [2][128]    running sums of g_s_dst h | g_s_src h over the units of this workgroup
    __device__ explicit Smem(float* s)
        : tile(s), p(s + C::oP), z(s + C::oZ), idx(reinterpret_cast<uint32_t*>(s + C::oI)), x(s + C::oX),
          rs(reinterpret_cast<int*>(s + C::oRS)), rd(reinterpret_cast<int*>(s + C::oRD)), gsd(s + C::oG), red(s + C::oRed),
          att(s + C::oAtt), accA(s + C::oAcc) {}
};

// per-thread running partials of the parameter gradients (slow-path workgroups; the fast path keeps them in LDS)
struct Partials {
    float4 qd, qs;                      // sum_s g_s_dst[s] h[s], sum_s g_s_src[s] h[s]   (this lane's four columns)
    float pw[FN_MAX_EDGE_K + 1];        // mode 2: sum_e dz[e, head] (x[e, k], 1) of the edges this lane owned (slot FN_MAX_EDGE_K = bias)
};

// ================================= fast path: the unit's gradient rows and edge state live in LDS
// XM: 0 = no per-edge side array; 1 = x (mode 2, K == 1) staged in LDS; 2 = original edge ids (mode 0 with g_s_orig) staged in LDS;
// 3 = mode 2 with K > 1 (small levels): attributes straight from global memory in pass B
template <int H, int NT, int RPH, int RT, int MCAP>
__device__ __forceinline__ void unit_fast(const MolBwdLevel& L, const Smem<H, NT, RPH, RT, MCAP>& S, int r0, int nr, int e0, int me,
                                          int32_t* status, Stamp& ts) {
    using C = Cfg<H, NT, RPH, RT, MCAP>;
    constexpr int LPH = 32 / H, NHW = C::NHW, HS = C::HS, KS = FN_MAX_EDGE_K + 1;
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5, head = lane / LPH, j = lane % LPH, wv = tid >> 6;
    const fn_gat_plan& pl = L.pl;
    const int m = (int)pl.m;
    const int K = L.et.mode == 2 ? L.et.K : 0;
    const int xm = L.et.mode == 2 ? (K == 1 ? 1 : 3) : (L.g_s_orig ? 2 : 0);
    int bad = 0;

    // ---- one round trip: everything the unit needs; the addresses depend on the extents only.  Tile rows and the per-edge
    // float arrays go global -> LDS directly (no registers), the rest through a handful of registers.
    auto tile_round = [&](int round) {
        const int rows = min(nr - round * RT, RT);
#pragma unroll
        for (int i = 0; i < C::TI; ++i) {
            if (i * (NT / 32) + 2 * wv < rows) {                // wave-uniform: pieces beyond the round's rows are not requested
                const int x = tid + i * NT;
                const int row = min(x >> 5, rows - 1);
                dma_to_lds<16>(L.g_out + (size_t)(r0 + round * RT + row) * FN_D + (x & 31) * 4, S.tile + (size_t)(i * NT + wv * 64) * 4);
            }
        }
    };
    tile_round(0);
    float4 hrow[RPH];
#pragma unroll
    for (int k = 0; k < RPH; ++k) hrow[k] = ld4(L.h + (size_t)(r0 + min(hw + NHW * k, nr - 1)) * FN_D + lane * 4);
    int tq[C::EI], dq[C::EI];
    if (me > 0) {
        const int mlast = me - 1;
#pragma unroll
        for (int i = 0; i < C::EI; ++i) {
            if (i * NT + wv * 64 < me) {                        // wave-uniform: a wave's 64 positions start inside the unit's edges
                const int q = min(tid + i * NT, mlast);
#pragma unroll
                for (int hh = 0; hh < H; ++hh)
                    dma_to_lds<4>(L.p_sorted + (size_t)hh * m + e0 + q, S.p + hh * HS + i * NT + wv * 64);
                if (xm == 1) dma_to_lds<4>(L.et.x_sorted + e0 + q, S.x + i * NT + wv * 64);
                else if (xm == 2) dma_to_lds<4>(pl.eid_d + e0 + q, S.x + i * NT + wv * 64);
                tq[i] = pl.dst_s[e0 + q];
                dq[i] = pl.dpos_s[e0 + q];
            } else { tq[i] = r0;  dq[i] = e0; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < C::EI; ++i) { tq[i] = r0;  dq[i] = e0; }
    }
    const int rsv = pl.rowptr_s[r0 + min(tid, nr)] - pl.pos_base_s - e0;
    const int rdv = pl.rowptr_d[r0 + min(tid, nr)] - pl.pos_base_d - e0;
#pragma unroll
    for (int i = 0; i < C::EI; ++i) {
        const int q = tid + i * NT;
        if (q < me) {
            int t = tq[i] - r0, dp = dq[i] - e0;
            if ((unsigned)t >= (unsigned)nr || (unsigned)dp >= (unsigned)me) { bad = 1;  t = 0;  dp = 0; }
            S.idx[q] = (uint32_t)t | ((uint32_t)dp << 16);
        }
    }
    if (tid <= nr) {
        int a = rsv, b = rdv;
        if (a < 0 || a > me || (tid == 0 && a != 0) || (tid == nr && a != me)) { bad = 1;  a = a < 0 ? 0 : me; }
        if (b < 0 || b > me || (tid == 0 && b != 0) || (tid == nr && b != me)) { bad = 1;  b = b < 0 ? 0 : me; }
        S.rs[tid] = a;
        S.rd[tid] = b;
    }
    if (bad && status) atomicOr(status, 2);
    __syncthreads();
    ts.hit();                                                  // 2: loads landed, LDS image written

    // ---- pass A: dp_e and the probability-weighted sum of gradient rows, one LDS row read per edge.  Units with more rows
    // than the tile take further rounds: the next RT gradient rows replace the tile, every edge is handled in the round its
    // destination row is resident.
    float4 acc[RPH];
#pragma unroll
    for (int k = 0; k < RPH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int rounds = (nr + RT - 1) / RT;
    for (int round = 0; round < rounds; ++round) {
        if (round) {
            __syncthreads();                                   // everybody is done with the previous rows
            tile_round(round);
            __syncthreads();
        }
        const uint32_t t_lo = (uint32_t)(round * RT);
#pragma unroll
        for (int k = 0; k < RPH; ++k) {
            const int s = hw + NHW * k;
            if (s < nr) {
                int q = S.rs[s];
                const int q1 = S.rs[s + 1];
                for (; q < q1; q += 4) {                       // four edges per trip: four independent LDS round trips in flight
                    uint32_t w[4];
                    float4 g[4];
                    float pe[4], d[4];
                    bool on[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) w[u] = S.idx[min(q + u, q1 - 1)];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t t = (w[u] & 0xffffu) - t_lo;
                        on[u] = q + u < q1 && t < (uint32_t)RT;
                        g[u] = ld4(S.tile + (on[u] ? t : 0u) * FN_D + lane * 4);
                        pe[u] = fabsf(S.p[head * HS + (w[u] >> 16)]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) d[u] = head_sum<LPH>(dot4(hrow[k], g[u]));
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (on[u]) {
                            if (j == 0) S.z[head * HS + (w[u] >> 16)] = d[u];
                            fma4(acc[k], pe[u], g[u]);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    ts.hit();                                                  // 3: pass A

    // ---- pass B: softmax backward per destination row.  A half-wave per row, lane j of a head owns the in-edges 2j, 2j+1 of
    // each 2*LPH-edge chunk (consecutive positions); sums over a head's lanes are DPP row operations.  The edge outputs ride
    // along: dz in original edge order (mode 0) or this lane's share of sum_e dz (x_e, 1) (mode 2).
    float pw[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) pw[k] = 0.f;
    for (int t = hw; t < nr; t += NHW) {
        const int b = S.rd[t], deg = S.rd[t + 1] - b;
        const float* ph = S.p + head * HS + b;
        float* zh = S.z + head * HS + b;
        if (deg <= 2 * LPH) {
            const int i0 = 2 * j, i1 = 2 * j + 1;
            const bool h0 = i0 < deg, h1 = i1 < deg;
            const float ps0 = h0 ? ph[i0] : 0.f, ps1 = h1 ? ph[i1] : 0.f;
            const float d0 = h0 ? zh[i0] : 0.f, d1 = h1 ? zh[i1] : 0.f;
            const float p0 = fabsf(ps0), p1 = fabsf(ps1);
            const float c = head_sum<LPH>(fmaf(p0, d0, p1 * d1));
            const float dz0 = p0 * (d0 - c) * ((__float_as_uint(ps0) >> 31) ? L.slope : 1.f);
            const float dz1 = p1 * (d1 - c) * ((__float_as_uint(ps1) >> 31) ? L.slope : 1.f);
            if (h0) zh[i0] = dz0;
            if (h1) zh[i1] = dz1;
            const float gs = head_sum<LPH>(dz0 + dz1);
            if (j == 0) S.gsd[t * H + head] = gs;
            if (xm == 1) {
                pw[0] = fmaf(dz0, h0 ? S.x[b + i0] : 0.f, fmaf(dz1, h1 ? S.x[b + i1] : 0.f, pw[0]));
                pw[FN_MAX_EDGE_K] += dz0 + dz1;
            } else if (xm == 2) {
                const int* eidv = reinterpret_cast<const int*>(S.x) + b;
                if (h0) { const int eid = eidv[i0];  if (eid < pl.m_real) L.g_s_orig[(size_t)eid * H + head] = dz0; }
                if (h1) { const int eid = eidv[i1];  if (eid < pl.m_real) L.g_s_orig[(size_t)eid * H + head] = dz1; }
            } else if (xm == 3) {
                pw[FN_MAX_EDGE_K] += dz0 + dz1;
#pragma unroll
                for (int k = 0; k < FN_MAX_EDGE_K; ++k) {
                    if (k < K) {
                        const float* xk = L.et.x_sorted + (size_t)k * m + e0 + b;
                        pw[k] = fmaf(dz0, h0 ? xk[i0] : 0.f, fmaf(dz1, h1 ? xk[i1] : 0.f, pw[k]));
                    }
                }
            }
        } else {                                               // rare: more in-edges than a head's lanes can own at once
            float cp = 0.f;
            for (int i = j; i < deg; i += LPH) cp = fmaf(fabsf(ph[i]), zh[i], cp);
            const float c = head_sum<LPH>(cp);
            float gp = 0.f;
            for (int i = j; i < deg; i += LPH) {
                const float ps = ph[i];
                const float dz = fabsf(ps) * (zh[i] - c) * ((__float_as_uint(ps) >> 31) ? L.slope : 1.f);
                zh[i] = dz;
                gp += dz;
                if (xm == 1) { pw[0] = fmaf(dz, S.x[b + i], pw[0]);  pw[FN_MAX_EDGE_K] += dz; }
                else if (xm == 2) { const int eid = reinterpret_cast<const int*>(S.x)[b + i];  if (eid < pl.m_real) L.g_s_orig[(size_t)eid * H + head] = dz; }
                else if (xm == 3) {
                    pw[FN_MAX_EDGE_K] += dz;
#pragma unroll
                    for (int k = 0; k < FN_MAX_EDGE_K; ++k)
                        if (k < K) pw[k] = fmaf(dz, L.et.x_sorted[(size_t)k * m + e0 + b + i], pw[k]);
                }
            }
            const float gs = head_sum<LPH>(gp);
            if (j == 0) S.gsd[t * H + head] = gs;
        }
    }
    if (xm == 1 || xm == 3) {                                  // this half-wave's share, added to its own LDS slots (no race)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k < K || k == FN_MAX_EDGE_K) {
                const float v = head_sum<LPH>(pw[k]);
                if (j == 0) S.red[hw * C::PE + head * (K + 1) + (k == FN_MAX_EDGE_K ? K : k)] += v;
            }
        }
    }
    __syncthreads();
    ts.hit();                                                  // 4: pass B

    // ---- pass C: per-source sums of dz, rank-one terms of the node scalars, rows out
    const float4 ad = ld4(S.att + lane * 4), as = ld4(S.att + FN_D + lane * 4);
    float4 qd = make_float4(0.f, 0.f, 0.f, 0.f), qs = qd;
#pragma unroll
    for (int k = 0; k < RPH; ++k) {
        const int s = hw + NHW * k;
        if (s < nr) {
            const int q0 = S.rs[s], q1 = S.rs[s + 1];
            float zs = 0.f;
            for (int q = q0 + j; q < q1; q += LPH) zs += S.z[head * HS + (S.idx[q] >> 16)];
            const float gss = head_sum<LPH>(zs);
            const float gsd = S.gsd[s * H + head];
            float4 a = acc[k];
            fma4(a, gsd, ad);
            fma4(a, gss, as);
            st4(L.g_h + (size_t)(r0 + s) * FN_D + lane * 4, a);
            fma4(qd, gsd, hrow[k]);
            fma4(qs, gss, hrow[k]);
        }
    }
    ts.hit();                                                  // 5: pass C (stores issued)
    // the unit's share of dL/da_dst | dL/da_src: half-waves through the tile (dead since pass A), fixed-order sum, added to
    // the workgroup's running columns
    st4(S.tile + hw * 2 * FN_D + lane * 4, qd);
    st4(S.tile + hw * 2 * FN_D + FN_D + lane * 4, qs);
    __syncthreads();
    if (tid < 2 * FN_D) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < NHW; ++w) a += S.tile[w * 2 * FN_D + tid];
        S.accA[tid] += a;
    }
    __syncthreads();                                           // the next unit overwrites the LDS image
    ts.hit();                                                  // 6: partial sums
}

// ================================= slow path: same passes, rows from global memory, edge state in the caller's scratch.
// Units beyond the size class of the launch (rows > R or edges > MCAP): correct, not fast, rare.  They are taken by extra
// workgroups of the launch (a branch of their own at the top of the kernel), so their registers and code do not weigh on the
// fast path and a long molecule does not hold up the workgroup that owns its neighbours.
template <int H, int NT>
__device__ __forceinline__ void unit_slow(const MolBwdLevel& L, int r0, int nr, int e0, int me, Partials& P, float4 ad, float4 as,
                                       int32_t* status) {
    constexpr int LPH = 32 / H, NHW = NT / 32;
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5, head = lane / LPH, j = lane % LPH;
    const fn_gat_plan& pl = L.pl;
    const int m = (int)pl.m, n = (int)pl.n;
    const int K = L.et.mode == 2 ? L.et.K : 0;
    float* __restrict__ z = L.scr_z;
    float* __restrict__ gsd_g = L.scr_gsd;
    int bad = 0;
    for (int s = hw; s < nr; s += NHW) {
        const float4 hr = ld4(L.h + (size_t)(r0 + s) * FN_D + lane * 4);
        const int q0 = pl.rowptr_s[r0 + s] - pl.pos_base_s, q1 = pl.rowptr_s[r0 + s + 1] - pl.pos_base_s;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = q0; q < q1; ++q) {
            int t = pl.dst_s[q], dp = pl.dpos_s[q];
            if ((unsigned)(t - r0) >= (unsigned)nr || (unsigned)(dp - e0) >= (unsigned)me) bad = 1;
            t = min(max(t, 0), n - 1);  dp = min(max(dp, 0), m - 1);
            const float4 gv = ld4(L.g_out + (size_t)t * FN_D + lane * 4);
            const float pe = fabsf(L.p_sorted[(size_t)head * m + dp]);
            const float d = head_sum<LPH>(dot4(hr, gv));
            if (j == 0) z[(size_t)head * m + dp] = d;
            fma4(a, pe, gv);
        }
        st4(L.g_h + (size_t)(r0 + s) * FN_D + lane * 4, a);
    }
    if (bad && status) atomicOr(status, 2);
    __threadfence_block();
    __syncthreads();
    for (int t = hw; t < nr; t += NHW) {
        const int b = pl.rowptr_d[r0 + t] - pl.pos_base_d, deg = pl.rowptr_d[r0 + t + 1] - pl.pos_base_d - b;
        const float* ph = L.p_sorted + (size_t)head * m + b;
        float* zh = z + (size_t)head * m + b;
        float cp = 0.f;
        for (int i = j; i < deg; i += LPH) cp = fmaf(fabsf(ph[i]), zh[i], cp);
        const float c = head_sum<LPH>(cp);
        float gp = 0.f;
        for (int i = j; i < deg; i += LPH) {
            const float ps = ph[i];
            const float dz = fabsf(ps) * (zh[i] - c) * ((__float_as_uint(ps) >> 31) ? L.slope : 1.f);
            zh[i] = dz;
            gp += dz;
            if (L.et.mode == 0) {
                if (L.g_s_orig) { const int eid = pl.eid_d[b + i];  if (eid < pl.m_real) L.g_s_orig[(size_t)eid * H + head] = dz; }
            } else {
                P.pw[FN_MAX_EDGE_K] += dz;
#pragma unroll
                for (int k = 0; k < FN_MAX_EDGE_K; ++k)
                    if (k < K) P.pw[k] = fmaf(dz, L.et.x_sorted[(size_t)k * m + b + i], P.pw[k]);
            }
        }
        const float gs = head_sum<LPH>(gp);
        if (j == 0) gsd_g[(size_t)(r0 + t) * H + head] = gs;
    }
    __threadfence_block();
    __syncthreads();
    for (int s = hw; s < nr; s += NHW) {
        const float4 hr = ld4(L.h + (size_t)(r0 + s) * FN_D + lane * 4);
        const int q0 = pl.rowptr_s[r0 + s] - pl.pos_base_s, q1 = pl.rowptr_s[r0 + s + 1] - pl.pos_base_s;
        float zs = 0.f;
        for (int q = q0 + j; q < q1; q += LPH) zs += z[(size_t)head * m + min(max(pl.dpos_s[q], 0), m - 1)];
        const float gss = head_sum<LPH>(zs);
        const float gsd = gsd_g[(size_t)(r0 + s) * H + head];
        float4 a = ld4(L.g_h + (size_t)(r0 + s) * FN_D + lane * 4);
        fma4(a, gsd, ad);
        fma4(a, gss, as);
        st4(L.g_h + (size_t)(r0 + s) * FN_D + lane * 4, a);
        fma4(P.qd, gsd, hr);
        fma4(P.qs, gss, hr);
    }
    __syncthreads();
}

// ---- rows and edges behind the last real molecule of a padded batch: gradients are exactly zero there
template <int H, int NT>
__device__ __forceinline__ void zero_tail(const MolBwdArgs& A, const MolBwdLevel& L, int zb) {
    const int n_real = A.counts_dev ? min(A.counts_dev[0], (int)A.n_mols) : (int)A.n_mols;
    int r_end = 0, e_end = 0;
    if (n_real > 0) {
        int r0, nr, e0, me;
        lv_extent(A.ext[n_real - 1], L.which, r0, nr, e0, me);
        r_end = r0 + nr;  e_end = e0 + me;
    }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t n4 = (int64_t)(L.pl.n - r_end) * 32;
    float* base = L.g_h + (size_t)r_end * FN_D;
    for (int64_t i = (int64_t)zb * NT + threadIdx.x; i < n4; i += (int64_t)A.n_zero * NT) st4(base + i * 4, z4);
    if (L.et.mode == 0 && L.g_s_orig) {
        const int loops = L.pl.m > L.pl.m_real ? 1 : 0;          // the atom graph's self loops sit one per row in the sorted order
        const int64_t o0 = (int64_t)(e_end - loops * r_end) * H, o1 = (int64_t)L.pl.m_real * H;
        for (int64_t i = o0 + (int64_t)zb * NT + threadIdx.x; i < o1; i += (int64_t)A.n_zero * NT) L.g_s_orig[i] = 0.f;
    }
}

// the units a workgroup takes: fast workgroups walk units bid, bid + n_fast, ...; the level's extra (slow) workgroups walk the
// same units and take what the size class does not fit (usually nothing: they read a few extents and leave).  Each unit is
// cut into the longest runs of consecutive molecules that fit; a single molecule that does not fit is a slow chunk.
template <int R, int MCAP, typename Fn>
__device__ __forceinline__ void walk_units(const MolBwdArgs& A, const MolBwdLevel& L, int first, int stride, bool want_fit, Fn&& fn) {
    const int n_real = A.counts_dev ? min(A.counts_dev[0], (int)A.n_mols) : (int)A.n_mols;
    const int G = L.mols_per_unit;
    const int* __restrict__ ext = reinterpret_cast<const int*>(A.ext);
    const int xo = L.which == fni::LV_BOND ? 2 : L.which == fni::LV_ATOM ? 0 : L.which == fni::LV_FBOND ? 6 : 4;       // MolExt words: rows
    const int eo = L.which == fni::LV_BOND ? 8 : L.which == fni::LV_ATOM ? 10 : L.which == fni::LV_FBOND ? 12 : 14;    // ... and edges
    for (int u = first; u < L.n_units; u += stride) {
        int k0 = u * G;
        const int kend = min(k0 + G, n_real);
        while (k0 < kend) {
            const int r0 = ext[k0 * 16 + xo], e0 = ext[k0 * 16 + eo];
            int nr = ext[k0 * 16 + xo + 1], me = ext[k0 * 16 + eo + 1];
            int k1 = k0 + 1;
            while (k1 < kend) {
                const int r2 = ext[k1 * 16 + xo] + ext[k1 * 16 + xo + 1] - r0, e2 = ext[k1 * 16 + eo] + ext[k1 * 16 + eo + 1] - e0;
                if (r2 > R || e2 > MCAP) break;
                nr = r2;  me = e2;
                ++k1;
            }
            k0 = k1;
            if (nr <= 0) continue;
            const bool fits = nr <= R && me <= MCAP && !A.force_slow;
            if (fits == want_fit) fn(r0, nr, e0, me);
        }
    }
}

template <int H, int NT, int RPH, int RT, int MCAP>
__global__ __launch_bounds__(NT, 4) void k_mol_bwd(const MolBwdArgs A) {
    using C = Cfg<H, NT, RPH, RT, MCAP>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Smem<H, NT, RPH, RT, MCAP> S(smem);
    constexpr int LPH = 32 / H, KS = FN_MAX_EDGE_K + 1;
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5, head = lane / LPH, j = lane % LPH;
    int li = 0;
    while (li + 1 < A.n_lv && (int)blockIdx.x >= A.lv[li + 1].first_blk) ++li;
    const int unit_blocks = A.lv[A.n_lv - 1].first_blk + A.lv[A.n_lv - 1].n_blk;
    if ((int)blockIdx.x >= unit_blocks) {
        const int z = (int)blockIdx.x - unit_blocks;
        zero_tail<H, NT>(A, A.lv[z / A.n_zero], z % A.n_zero);
        return;
    }
    const MolBwdLevel& L = A.lv[li];
    const int bid = (int)blockIdx.x - L.first_blk;
    const int K = L.et.K, ne = H * (K + 1);
    if (bid >= L.n_fast) {
        // ---- a workgroup for the units beyond the size class: partials in registers, reduced through LDS at the end
        Partials P;
        P.qd = P.qs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < KS; ++k) P.pw[k] = 0.f;
        const float4 ad = ld4(L.att + head * L.att_w + L.dst_off + j * 4);
        const float4 as = ld4(L.att + head * L.att_w + L.src_off + j * 4);
        walk_units<C::R, MCAP>(A, L, bid - L.n_fast, L.n_blk - L.n_fast, false,
                               [&](int r0, int nr, int e0, int me) { unit_slow<H, NT>(L, r0, nr, e0, me, P, ad, as, A.status); });
        st4(S.tile + hw * 2 * FN_D + lane * 4, P.qd);
        st4(S.tile + hw * 2 * FN_D + FN_D + lane * 4, P.qs);
        if (L.et.mode == 2) {
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                if (k < K || k == FN_MAX_EDGE_K) {
                    const float v = head_sum<LPH>(P.pw[k]);
                    if (j == 0) S.red[hw * C::PE + head * (K + 1) + (k == FN_MAX_EDGE_K ? K : k)] = v;
                }
            }
        }
        __syncthreads();
        for (int c = tid; c < 2 * FN_D; c += NT) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < C::NHW; ++w) a += S.tile[w * 2 * FN_D + c];
            L.part_a[(size_t)c * FN_MAX_PART + bid] = a;
        }
        if (L.et.mode == 2 && tid < ne) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < C::NHW; ++w) a += S.red[w * C::PE + tid];
            L.part_e[(size_t)bid * ne + tid] = a;
        }
        return;
    }
    // ---- fast workgroup
    Stamp ts{A.stamps ? A.stamps + (size_t)blockIdx.x * kStamps : nullptr, 0};
    ts.hit();                                                  // 0: start
    if (ts.p && threadIdx.x == 0) ts.p[14] = wall_clock64();   // 100 MHz reference for the tick rate
    // running sums of this workgroup and the attention vector's two blocks (one 1-KiB LDS-DMA of wave 0: lanes 0-31 the
    // destination block, 32-63 the source block, each in lane order = head-major)
    for (int i = tid; i < C::NHW * C::PE + 4 * FN_D; i += NT) S.red[i] = 0.f;      // red | att | accA are contiguous
    __syncthreads();
    if (tid < 64) {
        const int ll = tid & 31;
        dma_to_lds<16>(L.att + (ll / LPH) * L.att_w + (tid < 32 ? L.dst_off : L.src_off) + (ll % LPH) * 4, S.att);
    }
    walk_units<C::R, MCAP>(A, L, bid, L.n_fast, true, [&](int r0, int nr, int e0, int me) {
        ts.hit();                                              // 1: extents known
        unit_fast<H, NT, RPH, RT, MCAP>(L, S, r0, nr, e0, me, A.status, ts);
    });
    __syncthreads();
    for (int c = tid; c < 2 * FN_D; c += NT) L.part_a[(size_t)c * FN_MAX_PART + bid] = S.accA[c];
    if (L.et.mode == 2 && tid < ne) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < C::NHW; ++w) a += S.red[w * C::PE + tid];
        L.part_e[(size_t)bid * ne + tid] = a;
    }
    ts.hit();                                                  // 7 (one unit): partials written
    if (ts.p && threadIdx.x == 0) ts.p[15] = wall_clock64();
}

template <int H, int NT, int RPH, int RT, int MCAP>
int launch_class(const MolBwdArgs& A, int grid, hipStream_t st) {
    using C = Cfg<H, NT, RPH, RT, MCAP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mol_bwd<H, NT, RPH, RT, MCAP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
        if (e != hipSuccess) { (void)hipGetLastError();  return fni::fail((int)e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); }
        attr_set = true;
        if (getenv("FN_DEBUG_OCC")) {
            for (size_t lds : {C::lds_bytes, (size_t)80 * 1024, (size_t)78 * 1024, (size_t)72 * 1024, (size_t)64 * 1024, (size_t)48 * 1024}) {
                int nb = -1;
                hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mol_bwd<H, NT, RPH, RT, MCAP>, NT, lds);
                fprintf(stderr, "[mol_bwd] NT=%d lds=%zu -> %d blocks per CU (err %d)\n", NT, lds, nb, (int)e2);
            }
        }
    }
    hipLaunchKernelGGL((k_mol_bwd<H, NT, RPH, RT, MCAP>), dim3(grid), dim3(NT), C::lds_bytes, st, A);
    return fni::launch_status("molecule-resident attention backward");
}

}  // namespace

namespace fni {

bool mol_bwd_supported(int heads) { return heads == 4; }

int launch_mol_bwd(MolBwdLevel* lv, int n_lv, const MolExt* ext, int64_t n_mols, int64_t rows_hint, const int32_t* counts_dev,
                   int32_t* status, int heads, hipStream_t st) {
    if (!mol_bwd_supported(heads)) return fail(FN_EUNSUPPORTED, "molecule-resident backward: heads must be 4");
    if (n_lv < 1 || n_lv > kMolBwdMaxLevels || !ext || n_mols < 1) return fail(FN_EINVAL, "molecule-resident backward: bad arguments");
    MolBwdArgs A{};
    A.n_lv = n_lv;  A.ext = ext;  A.n_mols = n_mols;  A.counts_dev = counts_dev;  A.status = status;
    A.n_zero = counts_dev ? 8 : 0;
    A.force_slow = tune(FN_TUNE_BWD_MOL_FORCE_SLOW) == 1;
    int64_t n_stamps = 0;
    unsigned long long* sbuf = stamps(&n_stamps);
    int grid = 0;
    for (int i = 0; i < n_lv; ++i) {
        MolBwdLevel& L = lv[i];
        if (!L.g_out || !L.h || !L.p_sorted || !L.att || !L.g_h || !L.part_a || !L.scr_z || !L.scr_gsd || (L.et.mode == 2 && !L.part_e))
            return fail(FN_EINVAL, "molecule-resident backward: null buffer");
        if (L.et.mode == 2 && (L.et.K < 1 || L.et.K > FN_MAX_EDGE_K)) return fail(FN_EUNSUPPORTED, "molecule-resident backward: edge attribute width");
        if (L.pl.n >= (1 << 24) || L.pl.m >= (1ll << 31) / 8) return fail(FN_EUNSUPPORTED, "molecule-resident backward: level too large");
        if (L.mols_per_unit < 1) L.mols_per_unit = 1;
        L.n_units = (int)((n_mols + L.mols_per_unit - 1) / L.mols_per_unit);
        const int n_slow = std::min(L.n_units, 32);                 // workgroups for units beyond the size class
        L.n_fast = std::min(L.n_units, (int)FN_MAX_PART - n_slow);
        L.n_blk = L.n_fast + n_slow;
        L.first_blk = grid;
        grid += L.n_blk;
        A.lv[i] = L;
    }
    grid += A.n_zero * n_lv;
    A.stamps = n_stamps >= (int64_t)grid * kStamps ? sbuf : nullptr;
    // size class from the mean rows per molecule of the launch's largest level: the small class keeps two workgroups per CU
    const bool large = rows_hint > 60 * n_mols || tune(FN_TUNE_BWD_MOL_FORCE_SLOW) == 2;      // (2: dev switch, the large class)
    // small: 112 rows (7 per half-wave), 64-row tile rounds, 1024 edges, 79.4 KB of LDS -> two workgroups per CU;
    // large: 192 rows, 128-row tile rounds, 1536 edges, one 1024-thread workgroup per CU
    if (tune(FN_TUNE_BWD_MOL_FORCE_SLOW) == 3) return launch_class<4, 512, 7, 64, 832>(A, grid, st);      // dev: 71.6 KB of LDS
    if (tune(FN_TUNE_BWD_MOL_FORCE_SLOW) == 4) return launch_class<4, 512, 7, 48, 704>(A, grid, st);      // dev: 58 KB of LDS
    return large ? launch_class<4, 1024, 6, 128, 1536>(A, grid, st) : launch_class<4, 512, 7, 64, 1024>(A, grid, st);
}

}  // namespace fni
