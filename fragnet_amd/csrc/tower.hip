// PretrainTask's tall towers (reference model/gat/pretrain_heads.py:33-58, 77-88): Linear(128 -> 64) -> ReLU -> Linear(64 -> 32)
// -> ReLU -> Linear(32 -> 1) applied to every atom (bond-angle tower) and every directed bond (dihedral tower) -- 13 k / 27 k rows
// at ESOL batch 512, which the molecule-sized dense-head kernels (csrc/dense_head.inc, whole reduction in one workgroup) cannot
// take and which ran as library GEMMs plus element-wise launches (~35 launches, ~180 us per step).  Here a tower is ONE launch
// each way (several towers share it) plus one reduction launch:
//   forward   the tower's 41 KB of weights sit in LDS (transposed, so that a thread's four outputs are one 16-byte read); a
//             workgroup walks 32-row tiles: X tile -> LDS, the three layers back to back with the hidden rows in LDS, h1 / h2
//             written once for the backward pass;
//   backward  per 32-row tile: gH2 = g w3 (h2 > 0), gH1 = gH2 W2 (h1 > 0), gX = gH1 W1; the weight gradients dW1 (64 x 128),
//             dW2 (32 x 64) accumulate in REGISTERS across the workgroup's tiles (each thread owns a fixed patch) and leave as
//             one partial row per workgroup; bias gradients and dW3 likewise;
//   reduce    fixed-order column sums of the <= 256 partial rows straight into the parameter-gradient buffers.
// fp32 vector FMAs on register tiles (the fp32 matrix cores run at the vector rate on gfx950, and these shapes are 64 / 32
// wide); every operand of the inner loops is a broadcast or conflict-free 16-byte LDS read.  No atomics: bitwise reproducible.
#include <algorithm>

#include "fn_internal.h"

namespace {

constexpr int D0 = 128, D1 = 64, D2 = 32;
constexpr int kTwRows = 32;                       // rows of a tile
constexpr int XS = D0 + 4, H1S = D1 + 4, H2S = D2 + 4;      // padded LDS row strides (floats): rows 2 apart land 8 banks apart
constexpr int kPartW = D1 * D0 + D1 + D2 * D1 + D2 + D2 + 1;      // dW1 | db1 | dW2 | db2 | dW3 | db3 = 10369 floats per partial row
constexpr int oDB1 = D1 * D0, oDW2 = oDB1 + D1, oDB2 = oDW2 + D2 * D1, oDW3 = oDB2 + D2, oDB3 = oDW3 + D2;
constexpr int kTwMaxBwdBlocks = 256;

struct TowerTasks {
    fn_tower t[FN_MAX_TOWERS];
    int first[FN_MAX_TOWERS], nblk[FN_MAX_TOWERS];
    float* part[FN_MAX_TOWERS];                   // backward: [nblk][kPartW]
    int n;
};

__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ float relu(float v) { return v > 0.f ? v : 0.f; }

// ---------------------------------------------------------------- forward
constexpr int kFwdLds = D0 * D1 + D1 * D2 + D2 + D1 + D2 + kTwRows * XS + kTwRows * H1S + kTwRows * H2S;      // floats (71.6 KB)

__global__ __launch_bounds__(256, 2) void k_tower_fwd(const TowerTasks T) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* W1t = sm;                       // [128][64]  W1t[k][o] = W1[o][k]
    float* W2t = W1t + D0 * D1;            // [64][32]
    float* w3s = W2t + D1 * D2;            // [32]
    float* b1s = w3s + D2;                 // [64]
    float* b2s = b1s + D1;                 // [32]
    float* Xs = b2s + D2;                  // [32][XS]
    float* H1s = Xs + kTwRows * XS;        // [32][H1S]
    float* H2s = H1s + kTwRows * H1S;      // [32][H2S]
    const int tid = threadIdx.x;
    int ti = 0;
    while (ti + 1 < T.n && (int)blockIdx.x >= T.first[ti + 1]) ++ti;
    const fn_tower& t = T.t[ti];
    const int blk = (int)blockIdx.x - T.first[ti], nblk = T.nblk[ti];
    // weights -> LDS, transposed (consecutive threads take consecutive outputs: contiguous LDS writes)
    for (int i = tid; i < D1 * (D0 / 4); i += 256) {
        const int o = i % D1, k4 = i / D1;
        const float4 v = ld4(t.w1 + (size_t)o * D0 + 4 * k4);
        W1t[(4 * k4 + 0) * D1 + o] = v.x;  W1t[(4 * k4 + 1) * D1 + o] = v.y;  W1t[(4 * k4 + 2) * D1 + o] = v.z;  W1t[(4 * k4 + 3) * D1 + o] = v.w;
    }
    for (int i = tid; i < D2 * (D1 / 4); i += 256) {
        const int o = i % D2, k4 = i / D2;
        const float4 v = ld4(t.w2 + (size_t)o * D1 + 4 * k4);
        W2t[(4 * k4 + 0) * D2 + o] = v.x;  W2t[(4 * k4 + 1) * D2 + o] = v.y;  W2t[(4 * k4 + 2) * D2 + o] = v.z;  W2t[(4 * k4 + 3) * D2 + o] = v.w;
    }
    if (tid < D2) { w3s[tid] = t.w3[tid];  b2s[tid] = t.b2[tid]; }
    if (tid < D1) b1s[tid] = t.b1[tid];
    const float b3 = t.b3[0];
    const int64_t M = t.M;
    const int64_t tiles = (M + kTwRows - 1) / kTwRows;
    const int rg = tid >> 4, og = tid & 15;            // rows 2rg, 2rg+1; layer 1: outputs 4og..4og+3; layer 2: outputs 2og, 2og+1
    for (int64_t tile = blk; tile < tiles; tile += nblk) {
        const int64_t r0 = tile * kTwRows;
        __syncthreads();                               // weights staged / the previous tile's rows are done with
        for (int i = tid; i < kTwRows * (D0 / 4); i += 256) {
            const int r = i >> 5, c = i & 31;
            const float4 v = r0 + r < M ? ld4(t.x + (size_t)(r0 + r) * D0 + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            st4(Xs + r * XS + c * 4, v);
        }
        __syncthreads();
        {   // layer 1: 32 x 64 outputs, a 2 x 4 patch per thread
            float4 a0 = ld4(b1s + 4 * og), a1 = a0;
            const float* xa = Xs + (2 * rg) * XS;
            const float* xb = xa + XS;
#pragma unroll 4
            for (int k4 = 0; k4 < D0 / 4; ++k4) {
                const float4 va = ld4(xa + 4 * k4), vb = ld4(xb + 4 * k4);
                const float4 w0 = ld4(W1t + (4 * k4 + 0) * D1 + 4 * og), w1 = ld4(W1t + (4 * k4 + 1) * D1 + 4 * og);
                const float4 w2 = ld4(W1t + (4 * k4 + 2) * D1 + 4 * og), w3 = ld4(W1t + (4 * k4 + 3) * D1 + 4 * og);
                fma4(a0, va.x, w0);  fma4(a0, va.y, w1);  fma4(a0, va.z, w2);  fma4(a0, va.w, w3);
                fma4(a1, vb.x, w0);  fma4(a1, vb.y, w1);  fma4(a1, vb.z, w2);  fma4(a1, vb.w, w3);
            }
            a0 = make_float4(relu(a0.x), relu(a0.y), relu(a0.z), relu(a0.w));
            a1 = make_float4(relu(a1.x), relu(a1.y), relu(a1.z), relu(a1.w));
            st4(H1s + (2 * rg) * H1S + 4 * og, a0);
            st4(H1s + (2 * rg + 1) * H1S + 4 * og, a1);
            if (r0 + 2 * rg < M) st4(t.h1 + (size_t)(r0 + 2 * rg) * D1 + 4 * og, a0);
            if (r0 + 2 * rg + 1 < M) st4(t.h1 + (size_t)(r0 + 2 * rg + 1) * D1 + 4 * og, a1);
        }
        __syncthreads();
        {   // layer 2: 32 x 32 outputs, a 2 x 2 patch per thread
            float2 c0 = ld2(b2s + 2 * og), c1 = c0;
            const float* ha = H1s + (2 * rg) * H1S;
            const float* hb = ha + H1S;
#pragma unroll 4
            for (int k4 = 0; k4 < D1 / 4; ++k4) {
                const float4 va = ld4(ha + 4 * k4), vb = ld4(hb + 4 * k4);
                const float2 w0 = ld2(W2t + (4 * k4 + 0) * D2 + 2 * og), w1 = ld2(W2t + (4 * k4 + 1) * D2 + 2 * og);
                const float2 w2 = ld2(W2t + (4 * k4 + 2) * D2 + 2 * og), w3 = ld2(W2t + (4 * k4 + 3) * D2 + 2 * og);
                c0.x = fmaf(va.x, w0.x, fmaf(va.y, w1.x, fmaf(va.z, w2.x, fmaf(va.w, w3.x, c0.x))));
                c0.y = fmaf(va.x, w0.y, fmaf(va.y, w1.y, fmaf(va.z, w2.y, fmaf(va.w, w3.y, c0.y))));
                c1.x = fmaf(vb.x, w0.x, fmaf(vb.y, w1.x, fmaf(vb.z, w2.x, fmaf(vb.w, w3.x, c1.x))));
                c1.y = fmaf(vb.x, w0.y, fmaf(vb.y, w1.y, fmaf(vb.z, w2.y, fmaf(vb.w, w3.y, c1.y))));
            }
            c0 = make_float2(relu(c0.x), relu(c0.y));
            c1 = make_float2(relu(c1.x), relu(c1.y));
            st2(H2s + (2 * rg) * H2S + 2 * og, c0);
            st2(H2s + (2 * rg + 1) * H2S + 2 * og, c1);
            if (r0 + 2 * rg < M) st2(t.h2 + (size_t)(r0 + 2 * rg) * D2 + 2 * og, c0);
            if (r0 + 2 * rg + 1 < M) st2(t.h2 + (size_t)(r0 + 2 * rg + 1) * D2 + 2 * og, c1);
        }
        __syncthreads();
        {   // layer 3: one output per row, eight lanes per row
            const int row = tid >> 3, part = tid & 7;
            const float4 hv = ld4(H2s + row * H2S + 4 * part), wv = ld4(w3s + 4 * part);
            float s = dot4(hv, wv);
            s += __shfl_xor(s, 1);  s += __shfl_xor(s, 2);  s += __shfl_xor(s, 4);
            if (part == 0 && r0 + row < M) t.out[r0 + row] = s + b3;
        }
    }
}

// ---------------------------------------------------------------- backward
// floats: W1 [64][128] | W2 [32][64] | w3 [32] | gs [32] | Xs [32][XS] | H1s [32][H1S] | G1s [32][H1S] | H2s / G2s [32][H2S]
constexpr int kBwdLds = D1 * D0 + D2 * D1 + D2 + kTwRows + kTwRows * XS + 2 * kTwRows * H1S + kTwRows * H2S;      // 80.1 KB

__global__ __launch_bounds__(512, 2) void k_tower_bwd(const TowerTasks T) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* W1s = sm;                       // [64][128] as stored: gX[r][k] = sum_i gH1[r][i] W1[i][k]
    float* W2s = W1s + D1 * D0;            // [32][64]            gH1[r][i] = sum_j gH2[r][j] W2[j][i]
    float* w3s = W2s + D2 * D1;            // [32]
    float* gs = w3s + D2;                  // [32]  dL/d(out) of the tile's rows
    float* Xs = gs + kTwRows;              // [32][XS]
    float* H1s = Xs + kTwRows * XS;        // [32][H1S]
    float* G1s = H1s + kTwRows * H1S;      // [32][H1S]  gH1
    float* H2s = G1s + kTwRows * H1S;      // [32][H2S]  h2, then gH2 in place
    const int tid = threadIdx.x;
    int ti = 0;
    while (ti + 1 < T.n && (int)blockIdx.x >= T.first[ti + 1]) ++ti;
    const fn_tower& t = T.t[ti];
    const int blk = (int)blockIdx.x - T.first[ti], nblk = T.nblk[ti];
    for (int i = tid; i < D1 * D0 / 4; i += 512) st4(W1s + 4 * i, ld4(t.w1 + 4 * i));
    for (int i = tid; i < D2 * D1 / 4; i += 512) st4(W2s + 4 * i, ld4(t.w2 + 4 * i));
    if (tid < D2) w3s[tid] = t.w3[tid];
    const int64_t M = t.M;
    const int64_t tiles = (M + kTwRows - 1) / kTwRows;
    // fixed register patches of the weight gradients: dW1 [64][128]: rows 4 ig1 .. +3, columns 4 kg1 .. +3;  dW2 [32][64]: row jg2,
    // columns 4 ig2 .. +3 (threads 0..511 cover 32 x 16 patches)
    const int ig1 = tid >> 5, kg1 = tid & 31;
    const int jg2 = tid >> 4, ig2 = tid & 15;
    float4 dw1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) dw1[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dw2 = make_float4(0.f, 0.f, 0.f, 0.f);
    float small = 0.f;                                 // tid < 64: db1[tid]; 64..95: db2; 96..127: dW3; 128: db3
    const int rg = tid >> 5, cg = tid & 31;            // tile products: row rg (+16), column group cg
    for (int64_t tile = blk; tile < tiles; tile += nblk) {
        const int64_t r0 = tile * kTwRows;
        __syncthreads();                               // weights staged / the previous tile is done with
        for (int i = tid; i < kTwRows * (D0 / 4); i += 512) {
            const int r = i >> 5, c = i & 31;
            st4(Xs + r * XS + c * 4, r0 + r < M ? ld4(t.x + (size_t)(r0 + r) * D0 + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f));
        }
        {
            const int r = tid >> 4, c = tid & 15;      // 32 rows x 16 float4 of h1
            st4(H1s + r * H1S + c * 4, r0 + r < M ? ld4(t.h1 + (size_t)(r0 + r) * D1 + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (tid < kTwRows * (D2 / 4)) {
            const int r = tid >> 3, c = tid & 7;       // 32 rows x 8 float4 of h2
            st4(H2s + r * H2S + c * 4, r0 + r < M ? ld4(t.h2 + (size_t)(r0 + r) * D2 + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (tid < kTwRows) gs[tid] = r0 + tid < M ? t.g_out[r0 + tid] : 0.f;      // padding rows: zero gradient everywhere below
        __syncthreads();
        // dW3[j] = sum_r g_r h2[r][j], db3 = sum_r g_r (before h2 is overwritten)
        if (tid >= 96 && tid < 128) {
            const int j = tid - 96;
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < kTwRows; ++r) a = fmaf(gs[r], H2s[r * H2S + j], a);
            small += a;
        } else if (tid == 128) {
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < kTwRows; ++r) a += gs[r];
            small += a;
        }
        __syncthreads();
        // gH2[r][j] = g_r w3[j] (h2 > 0), in place: 32 x 32 values, two per thread
        {
            const int r = tid >> 4, j2 = (tid & 15) * 2;
            float2 h = ld2(H2s + r * H2S + j2);
            const float g = gs[r];
            h.x = h.x > 0.f ? g * w3s[j2] : 0.f;
            h.y = h.y > 0.f ? g * w3s[j2 + 1] : 0.f;
            st2(H2s + r * H2S + j2, h);
        }
        __syncthreads();
        // gH1[r][i] = (sum_j gH2[r][j] W2[j][i]) (h1 > 0): 32 x 64 values, four per thread (row rg and rg + 16, columns 2 cg, 2 cg + 1)
        {
            float2 a0 = make_float2(0.f, 0.f), a1 = a0;
            const float* ga = H2s + rg * H2S;
            const float* gb = H2s + (rg + 16) * H2S;
#pragma unroll 4
            for (int j4 = 0; j4 < D2 / 4; ++j4) {
                const float4 va = ld4(ga + 4 * j4), vb = ld4(gb + 4 * j4);
                const float2 w0 = ld2(W2s + (4 * j4 + 0) * D1 + 2 * cg), w1 = ld2(W2s + (4 * j4 + 1) * D1 + 2 * cg);
                const float2 w2 = ld2(W2s + (4 * j4 + 2) * D1 + 2 * cg), w3 = ld2(W2s + (4 * j4 + 3) * D1 + 2 * cg);
                a0.x = fmaf(va.x, w0.x, fmaf(va.y, w1.x, fmaf(va.z, w2.x, fmaf(va.w, w3.x, a0.x))));
                a0.y = fmaf(va.x, w0.y, fmaf(va.y, w1.y, fmaf(va.z, w2.y, fmaf(va.w, w3.y, a0.y))));
                a1.x = fmaf(vb.x, w0.x, fmaf(vb.y, w1.x, fmaf(vb.z, w2.x, fmaf(vb.w, w3.x, a1.x))));
                a1.y = fmaf(vb.x, w0.y, fmaf(vb.y, w1.y, fmaf(vb.z, w2.y, fmaf(vb.w, w3.y, a1.y))));
            }
            const float2 ha = ld2(H1s + rg * H1S + 2 * cg), hb = ld2(H1s + (rg + 16) * H1S + 2 * cg);
            st2(G1s + rg * H1S + 2 * cg, make_float2(ha.x > 0.f ? a0.x : 0.f, ha.y > 0.f ? a0.y : 0.f));
            st2(G1s + (rg + 16) * H1S + 2 * cg, make_float2(hb.x > 0.f ? a1.x : 0.f, hb.y > 0.f ? a1.y : 0.f));
        }
        __syncthreads();
        // gX[r][k] = sum_i gH1[r][i] W1[i][k]: 32 x 128 values, eight per thread (rows rg, rg + 16; columns 4 cg .. +3)
        if (t.g_x) {
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
            const float* ga = G1s + rg * H1S;
            const float* gb = G1s + (rg + 16) * H1S;
#pragma unroll 4
            for (int i4 = 0; i4 < D1 / 4; ++i4) {
                const float4 va = ld4(ga + 4 * i4), vb = ld4(gb + 4 * i4);
                const float4 w0 = ld4(W1s + (4 * i4 + 0) * D0 + 4 * cg), w1 = ld4(W1s + (4 * i4 + 1) * D0 + 4 * cg);
                const float4 w2 = ld4(W1s + (4 * i4 + 2) * D0 + 4 * cg), w3 = ld4(W1s + (4 * i4 + 3) * D0 + 4 * cg);
                fma4(a0, va.x, w0);  fma4(a0, va.y, w1);  fma4(a0, va.z, w2);  fma4(a0, va.w, w3);
                fma4(a1, vb.x, w0);  fma4(a1, vb.y, w1);  fma4(a1, vb.z, w2);  fma4(a1, vb.w, w3);
            }
            if (r0 + rg < M) st4(t.g_x + (size_t)(r0 + rg) * D0 + 4 * cg, a0);
            if (r0 + rg + 16 < M) st4(t.g_x + (size_t)(r0 + rg + 16) * D0 + 4 * cg, a1);
        }
        // weight gradients of this tile into the thread's patches: dW1[i][k] += sum_r gH1[r][i] x[r][k];  dW2[j][i] += sum_r gH2[r][j] h1[r][i]
#pragma unroll 4
        for (int r = 0; r < kTwRows; ++r) {
            const float4 gv = ld4(G1s + r * H1S + 4 * ig1);
            const float4 xv = ld4(Xs + r * XS + 4 * kg1);
            fma4(dw1[0], gv.x, xv);  fma4(dw1[1], gv.y, xv);  fma4(dw1[2], gv.z, xv);  fma4(dw1[3], gv.w, xv);
            fma4(dw2, H2s[r * H2S + jg2], ld4(H1s + r * H1S + 4 * ig2));
        }
        if (tid < 64) {                                // db1[i] = sum_r gH1[r][i]
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < kTwRows; ++r) a += G1s[r * H1S + tid];
            small += a;
        } else if (tid < 96) {                         // db2[j] = sum_r gH2[r][j]
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < kTwRows; ++r) a += H2s[r * H2S + (tid - 64)];
            small += a;
        }
    }
    // this workgroup's partial row
    float* part = T.part[ti] + (size_t)blk * kPartW;
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(part + (size_t)(4 * ig1 + q) * D0 + 4 * kg1, dw1[q]);
    st4(part + oDW2 + jg2 * D1 + 4 * ig2, dw2);
    if (tid < 64) part[oDB1 + tid] = small;
    else if (tid < 96) part[oDB2 + tid - 64] = small;
    else if (tid < 128) part[oDW3 + tid - 96] = small;
    else if (tid == 128) part[oDB3] = small;
}

// column sums of each tower's partial rows into its six parameter-gradient buffers (fixed order: bitwise reproducible).
// A block = 64 columns x 4 row groups: four loads in flight per thread, the groups combined through LDS in order.
constexpr int kRedCols = 64;
constexpr int kRedStrips = (kPartW + kRedCols - 1) / kRedCols;
__global__ __launch_bounds__(256) void k_tower_reduce(const TowerTasks T) {
    __shared__ float red[4][kRedCols];
    const int ti = (int)blockIdx.x / kRedStrips, strip = (int)blockIdx.x % kRedStrips;
    const fn_tower& t = T.t[ti];
    const int cl = threadIdx.x & (kRedCols - 1), rgp = threadIdx.x >> 6;
    const int c = strip * kRedCols + cl;
    const int n = T.nblk[ti];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < kPartW) {
        const float* p = T.part[ti] + c;
        int r = rgp;
        for (; r + 12 < n; r += 16) {
            a0 += p[(size_t)r * kPartW];  a1 += p[(size_t)(r + 4) * kPartW];  a2 += p[(size_t)(r + 8) * kPartW];  a3 += p[(size_t)(r + 12) * kPartW];
        }
        for (; r < n; r += 4) a0 += p[(size_t)r * kPartW];
    }
    red[rgp][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rgp == 0 && c < kPartW) {
        const float v = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        if (c < oDB1) t.g_w1[c] = v;
        else if (c < oDW2) t.g_b1[c - oDB1] = v;
        else if (c < oDB2) t.g_w2[c - oDW2] = v;
        else if (c < oDW3) t.g_b2[c - oDB2] = v;
        else if (c < oDB3) t.g_w3[c - oDW3] = v;
        else t.g_b3[0] = v;
    }
}

int set_lds(const void* kern, size_t bytes) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { (void)hipGetLastError();  return fni::fail((int)e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); }
    return 0;
}

int check_towers(const fn_tower* tw, int n, bool bwd, const char* who) {
    if (!tw || n < 1 || n > FN_MAX_TOWERS) return fni::fail(FN_EINVAL, who);
    for (int i = 0; i < n; ++i) {
        const fn_tower& t = tw[i];
        if (t.M < 0 || t.M >= (1ll << 31) / D0) return fni::fail(FN_EINVAL, who);
        if (!t.w1 || !t.b1 || !t.w2 || !t.b2 || !t.w3 || !t.b3) return fni::fail(FN_EINVAL, who);
        if (t.M > 0 && (!t.x || !t.h1 || !t.h2 || (!bwd && !t.out) || (bwd && !t.g_out))) return fni::fail(FN_EINVAL, who);
        if (bwd && (!t.g_w1 || !t.g_b1 || !t.g_w2 || !t.g_b2 || !t.g_w3 || !t.g_b3)) return fni::fail(FN_EINVAL, who);
        if (((uintptr_t)t.x | (uintptr_t)t.h1 | (uintptr_t)t.h2 | (uintptr_t)t.w1 | (uintptr_t)t.w2 | (uintptr_t)t.g_x) & 15) return fni::fail(FN_EINVAL, who);
    }
    return 0;
}

int bwd_blocks(int64_t M) {
    const int64_t tiles = (M + kTwRows - 1) / kTwRows;
    return (int)std::max<int64_t>(1, std::min<int64_t>(tiles, kTwMaxBwdBlocks));
}

}  // namespace

extern "C" {

int fn_tower_fwd_f32(const fn_tower* towers, int n, fn_stream_t stream) {
    if (int rc = check_towers(towers, n, false, "fn_tower_fwd_f32: bad argument")) return rc;
    TowerTasks T{};
    int grid = 0;
    for (int i = 0; i < n; ++i) {
        if (towers[i].M == 0) continue;
        const int64_t tiles = (towers[i].M + kTwRows - 1) / kTwRows;
        T.t[T.n] = towers[i];
        T.first[T.n] = grid;
        T.nblk[T.n] = (int)std::min<int64_t>(tiles, 512);           // two resident workgroups per CU, each walking its tiles
        grid += T.nblk[T.n++];
    }
    if (!T.n) return 0;
    static bool once = false;
    if (!once) { if (int rc = set_lds(reinterpret_cast<const void*>(k_tower_fwd), kFwdLds * sizeof(float))) return rc;  once = true; }
    hipLaunchKernelGGL(k_tower_fwd, dim3(grid), dim3(256), kFwdLds * sizeof(float), reinterpret_cast<hipStream_t>(stream), T);
    return fni::launch_status("fn_tower_fwd_f32");
}

int64_t fn_tower_bwd_ws(const fn_tower* towers, int n) {
    if (!towers || n < 1 || n > FN_MAX_TOWERS) return 0;
    int64_t f = 0;
    for (int i = 0; i < n; ++i) f += (int64_t)bwd_blocks(towers[i].M) * kPartW;
    return f;
}

int fn_tower_bwd_f32(const fn_tower* towers, int n, float* ws, fn_stream_t stream) {
    if (int rc = check_towers(towers, n, true, "fn_tower_bwd_f32: bad argument")) return rc;
    if (!ws) return fni::fail(FN_EINVAL, "fn_tower_bwd_f32: null workspace");
    TowerTasks T{};
    int grid = 0;
    for (int i = 0; i < n; ++i) {       // M = 0: one workgroup that walks no tile writes a zero partial row -> zero gradients
        T.t[T.n] = towers[i];
        T.first[T.n] = grid;
        T.nblk[T.n] = bwd_blocks(towers[i].M);
        T.part[T.n] = ws;
        ws += (size_t)T.nblk[T.n] * kPartW;
        grid += T.nblk[T.n++];
    }
    static bool once = false;
    if (!once) { if (int rc = set_lds(reinterpret_cast<const void*>(k_tower_bwd), kBwdLds * sizeof(float))) return rc;  once = true; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_tower_bwd, dim3(grid), dim3(512), kBwdLds * sizeof(float), st, T);
    if (int rc = fni::launch_status("fn_tower_bwd_f32")) return rc;
    hipLaunchKernelGGL(k_tower_reduce, dim3(T.n * kRedStrips), dim3(256), 0, st, T);
    return fni::launch_status("fn_tower_bwd_f32 (reduction)");
}

}  // extern "C"
