// Internal declarations shared by the translation units of libfragnet_hip.so (not part of the C-ABI: nothing here is exported).
//  * device helpers every kernel file uses (vector loads, DPP reductions inside a head's lanes, Philox);
//  * the per-molecule extents table (MolExt) the molecule-resident kernels are driven by;
//  * host-side hooks into the error string / tuning table that live in fragnet_hip.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "fragnet_hip.h"

#define FNI_HIDDEN __attribute__((visibility("hidden")))

namespace fni {
FNI_HIDDEN int fail(int code, const char* what);        // sets fn_last_error(), returns code
FNI_HIDDEN int launch_status(const char* where);        // hipGetLastError() -> 0 / error code + message
FNI_HIDDEN int tune(int key);                           // fn_set_tuning table
FNI_HIDDEN unsigned long long* stamps(int64_t* n_u64);  // fn_debug_set_stamps buffer (null: none)

// extents of one molecule in the index spaces of a collated batch (dataset/data.py:877-948 concatenates molecules, so every
// one of these is a contiguous range); filled by k_mol_extents from the molecule CSRs and the level plans
struct MolExt {
    int a0, na;            // atoms
    int b0, nb;            // directed bonds = bond-graph nodes = atom-graph edges
    int f0, nf;            // fragments
    int c0, nc;            // fragment connections = fragment-bond-graph nodes = fragment-graph edges
    int eb0, meb;          // bond-graph edges: level-local destination-sorted positions
    int ea0, mea;          // atom-graph items (edges + self loops)
    int ef0, mef;          // fragment-bond-graph edges
    int ec0, mec;          // fragment-graph items
};
static_assert(sizeof(MolExt) == 64, "MolExt is sixteen int32 (include/fragnet_hip.h documents it as int32 [n_mols][16])");

// ---- argument blocks and host-side entry points shared by the translation units (the forward attention family lives in
// gat_fwd.hip / gat_fwd_lin.hip, everything else in fragnet_hip.hip)
struct GatFwdArgs {
    const float *h, *s_dst, *s_src, *att;
    int att_w;
    fn_edge_term et;
    fn_gat_plan pl;
    float slope;
    float *out, *p_sorted, *probs_orig;
    fn_act_epilogue ep;
    int rows_per_hw, nblk;
    // optional fused "row dots" of the level that consumes this one's raw output as its edge attribute (bond graph -> atom
    // graph, gat2.py:203-208): rd_out[j * rd_m + rd_pos[t]] = <out[t, :], rd_A[j * rd_lda : +128]>, j < rd_J -- the edge term
    // of the next level, written straight into ITS destination-sorted order (rd_pos = that level's inv_d)
    const float* rd_A;
    float* rd_out;
    const int32_t* rd_pos;
    int64_t rd_m;
    int rd_lda, rd_J;
    // optional second output for the one-pass backward (gat_bwd_one.inc): out2[t] = sum_e lambda_e p_e h[src_e] and
    // sigma[t, h] = sum_e lambda_e p_e, lambda_e = 1 where z_e > 0, else the LeakyReLU slope
    float *out2, *sigma;
    int p_edge_major;     // p_sorted as [m][H] instead of [H][m]: what the one-pass backward gathers by position (one line per edge)
    const int32_t* n_real;   // nullable device word: rows >= *n_real are padding (zero outputs, nothing gathered)
    int tier6;               // gather tiers 4 / 6 / 8 (1) or 4 / 8 (0)
};
struct NodeScalarEpi {             // optional fused epilogue: s_dst/s_src[row, head] = <Y[row, head cols], att blocks>
    const float* att;
    float* s_dst;
    float* s_src;
    int att_w, dst_off, src_off, heads;     // heads in {2, 4, 8}: a head's columns must lie inside one wave's 64
};
struct RowAdd {
    const float* z;           // [M][4]: dL/d(edge term) of row e, the four heads together; null: no term
    const float* a;           // a[h * lda + column]
    int lda;
};
struct CuEpi {
    const float *out, *out2, *sigma;
    float *c, *u;             // c == null: no such epilogue
    int heads;
};
struct GsdEpi {
    const float* dz;          // [m][4]; null: no such term
    const int32_t* rowptr;    // the level's by-destination CSR (M + 1 words); positions are rowptr[.] - pos_base
    int pos_base;
    const float* R;           // [4][128], see above
    float* gsd;               // out [M][4]
};
// up to three independent [M_i,K]·[K,128] products in one launch: the three projections of a layer (forward) or
// their three input-gradient products (backward) depend only on the previous layer, never on each other
struct LinTask {
    const float* Wn;          // the same weight n-major ([128 n][K]) for k_proj128; null: only the k_linear128 form is available
    const float *X, *Bt, *bias;
    float* Y;
    int64_t M;
    fn_act_epilogue mk;
    NodeScalarEpi ns;
    int first, nblk;
    int K;                    // 0: the group's K (LinTasks::K); else this task's own reduction length (layer 0: 17 bond / 6 connection features)
    RowAdd ra;                // riding input-gradient products only (lin_side_block)
    CuEpi cu;                 // ... of the one-pass backward (lin_side_block<true>)
    GsdEpi gs;                // ... of its deferred form (lin_side_block<true, true>)
    const int32_t* n_real;    // nullable device word: row tiles that start at or behind *n_real are padding and are not computed
};
struct LinTasks {
    LinTask t[3];
    int n, K;
    int base, total;          // co-launched with an attention pass (below): the GEMM workgroups are blocks [base, base + total) of that launch
};
// gat_fwd.hip
FNI_HIDDEN int prep_gat_fwd(const float* h, const float* s_dst, const float* s_src, const float* att, int att_w, const fn_edge_term* et,
                            const fn_gat_plan* plan, float neg_slope, float* out, float* p_sorted, float* probs_orig,
                            const fn_act_epilogue* act, int heads, GatFwdArgs* A, float* out2 = nullptr, float* sigma = nullptr);
FNI_HIDDEN int launch_gat_fwd(const GatFwdArgs& A, int heads, hipStream_t st);
FNI_HIDDEN int launch_gat_fwd_pair(const GatFwdArgs& A, const GatFwdArgs& B, int heads, hipStream_t st);
// gat_fwd_lin.hip: an attention pass + the K = 128 projection tiles that do not depend on it, in one launch
FNI_HIDDEN int launch_gat_fwd_lin(const GatFwdArgs& A, LinTasks& T, int heads, hipStream_t st);
FNI_HIDDEN int launch_gat_fwd_pair_lin(const GatFwdArgs& A, const GatFwdArgs& B, LinTasks& T, int heads, hipStream_t st);
// ---- the one-pass attention backward (gat_bwd_one.hip / gat_bwd_one.inc)
struct GatBwdOneArgs {
    const float *g_out, *h, *p_sorted, *cdot, *g_s_dst, *att;
    int att_w, dst_off, src_off;
    fn_edge_term et;
    fn_gat_plan pl;
    float slope;
    float *g_h, *part_a, *part_e, *dz_sorted, *g_s_orig;
    int rows_per_hw, nblk;
    int p_edge_major;       // p_sorted is [m][H] (the engine's forward writes it so: one cache line per edge) instead of [H][m]
    const float* x_src;     // mode 2, nullable: the raw attribute [K][m] in SOURCE order (streamed) instead of gathers from et.x_sorted
    const int32_t* n_real;        // nullable device word: source rows >= *n_real are padding (zero gradient row out, nothing gathered)
    int tier6;                    // gather tiers 4 / 6 / 8 / 12 (1) or 4 / 8 / 12 (0)
    float* dz_em;                 // deferred form (DF instances, below): dL/dz of every edge at its destination-order slot, EDGE-major [m][H]
    unsigned long long* stamps;   // dev aid (fn_debug_set_stamps): 16 s_memtime values per wave -- entry, loop start, per row (loads issued,
                                  // data arrived, row done) x 4, loop end, exit; null in production (one uniform branch per stamp)
};
struct CuTask {
    const float *g, *out, *out2, *sigma;
    float scale;
    float *c, *u;
    int64_t n;
    int first, nblk;
};
constexpr int kMaxCuTasks = 4;
struct CuTasks { CuTask t[kMaxCuTasks]; int n; };
struct GsdSegTask {
    const float* dz;          // [m][4]
    const int32_t* rowptr;    // by-destination CSR, n + 1 words; positions are rowptr[.] - pos_base
    int pos_base;
    int64_t n;
    float* gsd;               // out [n][4]
    const int32_t* n_real;    // nullable: rows at or behind *n_real are padding (their segments were never written): zeros
    const float* h;           // [n][128] the level's projected rows
    float* part_a;            // column-major [2 * 128][FN_MAX_PART]: columns 0..127, rows 0..nblk-1 are written
    int first, nblk;
};
struct GsdSegTasks { GsdSegTask t[3]; int n; };
FNI_HIDDEN int prep_gat_bwd_one(const float* g_out, const float* h, const float* p_sorted, const float* cdot, const float* g_s_dst,
                                const fn_edge_term* et, const float* att, int att_w, int dst_off, int src_off, const fn_gat_plan* plan,
                                float neg_slope, float* g_h, float* dz_sorted, float* g_s_orig, float* part_a, int* n_part_a, float* part_e,
                                int* n_part_e, int heads, GatBwdOneArgs* A, int64_t share = 0);
FNI_HIDDEN int launch_gat_bwd_one(const GatBwdOneArgs& A, int heads, hipStream_t st);
FNI_HIDDEN int launch_gat_bwd_one3(const GatBwdOneArgs& A, const GatBwdOneArgs& B, const GatBwdOneArgs& C, int heads, hipStream_t st);
FNI_HIDDEN int launch_gat_cu(CuTasks& T, int heads, hipStream_t st);
FNI_HIDDEN int launch_gsd_seg(const GsdSegTasks& T, int blocks, hipStream_t st);
// fragnet_hip.hip
FNI_HIDDEN int launch_linear128_group(LinTasks& T, hipStream_t st);
FNI_HIDDEN bool bad_edge_term(const fn_edge_term* et, int64_t m);
}  // namespace fni

namespace {
inline hipStream_t S(fn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
#define FN_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
// forward kind 2 (gat_fwd.inc): this level's run-time flags are the engine's training constants
inline bool fwd_kind_tr(const fni::GatFwdArgs& A, int heads) {
    return fni::tune(FN_TUNE_ENGINE_CONST) != 0 && A.out2 != nullptr && A.p_edge_major != 0 && A.probs_orig == nullptr && A.pl.m >= 2 &&
           (A.ep.y == nullptr || (A.ep.relu != 0 && A.ep.p > 0.f)) && (A.rd_out == nullptr || A.rd_J == heads);
}
// forward kind 3: the engine's evaluation launches (no second output, head-major probabilities, ReLU epilogue without dropout)
inline bool fwd_kind_ev(const fni::GatFwdArgs& A, int heads) {
    return fni::tune(FN_TUNE_ENGINE_CONST) != 0 && A.out2 == nullptr && A.p_edge_major == 0 && A.probs_orig == nullptr && A.pl.m >= 2 &&
           (A.ep.y == nullptr || (A.ep.relu != 0 && !(A.ep.p > 0.f))) && (A.rd_out == nullptr || A.rd_J == heads);
}
#define FN_DISPATCH_H(heads, CALL)                         \
    switch (heads) {                                       \
        case 1: { constexpr int HH = 1; CALL; } break;     \
        case 2: { constexpr int HH = 2; CALL; } break;     \
        case 4: { constexpr int HH = 4; CALL; } break;     \
        case 8: { constexpr int HH = 8; CALL; } break;     \
        default: return fni::fail(FN_EUNSUPPORTED, "heads must be 1, 2, 4 or 8 (128 = heads * head_dim)"); \
    }
constexpr int kBlock = 256;
constexpr int kRows = 8;          // rows (half-waves) per block
constexpr int kGridCap = 2048;    // memory-bound kernels: ~8 blocks per CU, grid-stride the rest
constexpr int kBwdRows = 8;       // rows (half-waves) per block in the attention backward kernels
inline int row_grid(int64_t rows, int cap) {
    int64_t g = (rows + kRows - 1) / kRows;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
inline int flat_grid(int64_t work, int cap) {
    int64_t g = (work + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
// edge class of an attention level = the KL template argument of its kernels: 0 stored edge term, 1 / FN_MAX_EDGE_K raw attributes
inline int edge_class(const fn_edge_term* et) { return et->mode == 0 ? 0 : (et->K == 1 ? 1 : FN_MAX_EDGE_K); }
inline int lin_blocks(int64_t tiles, int iters) { return 2 * (int)((tiles + iters - 1) / iters); }     // 64 x 64 tiles: two column halves per row tile

constexpr int kWfLd = FN_MAX_EDGE_K + 1;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void fma4(float4& acc, float s, float4 v) {
    acc.x = fmaf(s, v.x, acc.x); acc.y = fmaf(s, v.y, acc.y);
    acc.z = fmaf(s, v.z, acc.z); acc.w = fmaf(s, v.w, acc.w);
}

// XCD-aware work split.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one, each XCD has
// its own 4 MiB L2), so logical block ids are remapped to give every XCD one CONTIGUOUS range of rows: the
// source rows a destination gathers belong to the same molecule, i.e. to neighbouring rows, and then stay
// in that XCD's L2 instead of being fetched by all eight.  Bijective for any grid size; speed only.
__device__ __forceinline__ int xcd_block(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
#ifndef FN_XCD_REAL
#define FN_XCD_REAL 1
#endif
#ifndef FN_PAD_BLOCK_EXIT
#define FN_PAD_BLOCK_EXIT 1
#endif
// The same for a static-shape batch whose rows behind logical block `nreal` are padding.  Dealing CAPACITY blocks into eight chunks
// leaves the last XCD with mostly padding and the other seven with capacity / 8 rows each instead of real / 8 -- 8 % padding made the
// attention launches 6-9 % longer (round 4).  Here the REAL blocks [0, nreal) are dealt evenly (contiguous chunks as above) and the
// padding blocks follow behind them in (XCD, turn) order.  Bijective on [0, nwg) for any 0 <= nreal <= nwg.
__device__ __forceinline__ int xcd_block_real(int b, int nwg, int nreal) {
    if (!FN_XCD_REAL || nreal >= nwg || nreal <= 0) return xcd_block(b, nwg);
    const int x = b & 7, k = b >> 3;
    const int qr = (nreal >> 3) + (x < (nreal & 7) ? 1 : 0);          // real blocks of this XCD
    if (k < qr) return xcd_block(b, nreal);
    int before = 0;                                                    // padding blocks of the XCDs in front
#pragma unroll
    for (int y = 0; y < 7; ++y)
        if (y < x) before += ((nwg >> 3) + (y < (nwg & 7) ? 1 : 0)) - ((nreal >> 3) + (y < (nreal & 7) ? 1 : 0));
    return nreal + before + (k - qr);
}
// [begin, end) of the row groups (RB rows each) owned by this block: contiguous chunks, XCD-swizzled
__device__ __forceinline__ void block_groups(int64_t n_rows, int rb, int64_t& begin, int64_t& end) {
    const int64_t groups = (n_rows + rb - 1) / rb;
    const int64_t per = (groups + gridDim.x - 1) / gridDim.x;
    begin = (int64_t)xcd_block(blockIdx.x, gridDim.x) * per;
    end = begin + per < groups ? begin + per : groups;
}

// Philox-4x32 counter-based generator (Salmon et al., SC'11): 128 random bits per (counter, key), no state.  SEVEN rounds: the
// smallest count that passes BigCrush in the paper (Table 2; ten is its safety-margin default).  The bits only draw dropout masks,
// and the attention forward kernels are VALU-bound enough for the three rounds to matter: two 32 x 32 -> 64 multiplies (quarter
// rate) per round and lane, 1.2 % of the whole training step (0.809 -> 0.799 ms, profiles/r05_*; -DFN_PHILOX_ROUNDS=10 for the A/B).
// Every consumer of a mask -- the fused epilogues, fn_dropout_act_f32, its backward, the tests that replay masks -- calls this.
#ifndef FN_PHILOX_ROUNDS
#define FN_PHILOX_ROUNDS 7
#endif
__device__ __forceinline__ uint4 philox4x32(uint64_t ctr, uint64_t seed) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < FN_PHILOX_ROUNDS; ++r) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * c0, m1 = (uint64_t)0xCD9E8D57u * c2;       // one v_mad_u64_u32 each
        const uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0, hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// keep an element iff u >= p with u = (bits >> 8) / 2^24 -- decided on the integers: k / 2^24 >= p  <=>  k >= ceil(p 2^24) (both sides
// exact in fp32), which saves the conversion and the multiply per element; the threshold is loop-invariant
__device__ __forceinline__ float keep_scale(uint32_t bits, float p, float inv_keep) {
    const uint32_t thresh = (uint32_t)ceilf(p * 16777216.0f);
    return (bits >> 8) >= thresh ? inv_keep : 0.f;
}

template <int W> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
template <int W> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}


// ---- reductions over the LPH lanes of one head group.  DPP row operations (one VALU op each) instead of
// ds_bpermute: row_half_mirror pairs lane i with 7-i inside each 8 lanes, quad_perm covers xor 1 / xor 2.
// (mov_dpp with bound_ctrl: no `old` operand to materialise -- every lane of these permutations has a source -- so the compiler folds
// the move into the add that consumes it: one v_add_f32_dpp per reduction step instead of zero + v_mov_b32_dpp + add)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // lane i <- lane 7-i  (within 8)
constexpr int kDppMirror = 0x140;     // lane i <- lane 15-i (within 16)

template <int W> __device__ __forceinline__ float head_sum(float v) {
    static_assert(W == 4 || W == 8 || W == 16 || W == 32, "head group width");
    if (W == 32) v += __shfl_xor(v, 16);
    if (W >= 16) v += dpp_mov<kDppMirror>(v);
    if (W >= 8) v += dpp_mov<kDppHalfMirror>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppXor1>(v);
    return v;
}
template <int W> __device__ __forceinline__ float head_max(float v) {
    if (W == 32) v = fmaxf(v, __shfl_xor(v, 16));
    if (W >= 16) v = fmaxf(v, dpp_mov<kDppMirror>(v));
    if (W >= 8) v = fmaxf(v, dpp_mov<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_mov<kDppXor2>(v));
    v = fmaxf(v, dpp_mov<kDppXor1>(v));
    return v;
}


struct __attribute__((packed, aligned(4))) i32x2u { int x, y; };       // 4-byte aligned pairs: one dwordx2 load
struct __attribute__((packed, aligned(4))) f32x2u { float x, y; };
struct __attribute__((packed, aligned(4))) f32x4u { float x, y, z, w; };
__device__ __forceinline__ i32x2u ldp(const int32_t* p) { return *reinterpret_cast<const i32x2u*>(p); }
__device__ __forceinline__ f32x2u ldp(const float* p) { return *reinterpret_cast<const f32x2u*>(p); }
__device__ __forceinline__ void stp(float* p, float a, float b) { f32x2u v; v.x = a; v.y = b; *reinterpret_cast<f32x2u*>(p) = v; }


__device__ __forceinline__ float4 ld4_off(const float* base, uint32_t byte_off) {      // scalar base + 32-bit offset
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float ld1_off(const float* base, uint32_t byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
// the same for stores: a wave-uniform base pointer + a 32-bit byte offset is ONE address register per lane (global_store ... saddr);
// `p + (size_t)row * 128 + lane * 4` is a 64-bit multiply-add per lane and store (v_lshl_add_u64: 72 of the 672 vector instructions
// of the forward attention loop in round 4)
__device__ __forceinline__ void st4_off(float* base, uint32_t byte_off, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
__device__ __forceinline__ void st1_off(float* base, uint32_t byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}



// ---- explicit address spaces.  Pointers read out of an argument block in memory are generic; dereferenced as such they become
// flat_load / flat_store, which tick BOTH wait counters and so serialise with the LDS traffic.  Hot accesses cast to the global
// (G()) or LDS address space.
typedef float f32x4 __attribute__((ext_vector_type(4)));       // MFMA accumulator / 16-byte vector
#define FN_LDS __attribute__((address_space(3)))
#define FN_GLB __attribute__((address_space(1)))
typedef FN_LDS float lds_f;
typedef FN_LDS int lds_i;
typedef FN_GLB float glb_f;
typedef FN_GLB int glb_i;
__device__ __forceinline__ const glb_f* G(const float* p) { return (const glb_f*)p; }
__device__ __forceinline__ glb_f* G(float* p) { return (glb_f*)p; }
__device__ __forceinline__ const glb_i* G(const int* p) { return (const glb_i*)p; }
__device__ __forceinline__ float4 ld4(const glb_f* p) {
    const f32x4 v = *reinterpret_cast<const FN_GLB f32x4*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4(glb_f* p, float4 v) { *reinterpret_cast<FN_GLB f32x4*>(p) = (f32x4){v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ float4 ld4s(const lds_f* p) {
    const f32x4 v = *reinterpret_cast<const FN_LDS f32x4*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4s(lds_f* p, float4 v) { *reinterpret_cast<FN_LDS f32x4*>(p) = (f32x4){v.x, v.y, v.z, v.w}; }

// LDS-DMA (global_load_lds_*): the wave's 64 lanes write 64 x SIZE consecutive bytes at the wave-uniform LDS address; the
// global source address is per lane.  No VGPR destination: the tile costs no registers while it is in flight.
template <int SIZE>
__device__ __forceinline__ void dma_to_lds(const void* gsrc, void* lds_wave_base) {
    static_assert(SIZE == 4 || SIZE == 16, "dword or dwordx4 pieces");
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (SIZE == 16) __builtin_amdgcn_global_load_lds((const FN_GLB void*)gsrc, (FN_LDS void*)lds_wave_base, 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((const FN_GLB void*)gsrc, (FN_LDS void*)lds_wave_base, 4, 0, 0);
#endif
}

}  // namespace
