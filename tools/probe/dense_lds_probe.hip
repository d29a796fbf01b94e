// Probe (not product code): the prediction head's hidden Linear  Y[M,N] = X[M,K] W[N,K]^T  (FTHead3, model/gat/gat2.py:678-725;
// M = 512 molecules, K = N = 1024) on the fp32 matrix cores with WORKGROUP-SHARED operand tiles -- VERDICT r5 item 1:
//   * a 64 x 128 macro-tile per 4-wave workgroup (each wave a 32 x 64 register tile = 8 accumulators per 6 fragment reads from LDS),
//   * the operands staged ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4) into a ring of 32-k slots (24 KB each),
//   * the reduction split over S workgroups per tile (S = 4 at M = 512: 256 workgroups), partial slabs P[S][M][N], summed in fixed
//     order by a combine launch (bias + the activation would ride there) or by the consumer.
// Compared in the same process with the library's fn_dense_fwd_f32 (csrc/dense_head.inc: 32 x 64 tiles, eight waves splitting K inside
// the workgroup, every wave loading its own operand pieces from L2).  Prints back-to-back launch times (HIP events) and the error
// against an fp64 CPU product; run under rocprofv3 --kernel-trace for per-kernel durations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/probe/dense_lds_probe.hip -Lfragnet_amd/lib -lfragnet_hip \
//         -Wl,-rpath,'$ORIGIN/../../fragnet_amd/lib' -o tools/probe/dense_lds_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fragnet_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define LDS __attribute__((address_space(3)))
#define GLB __attribute__((address_space(1)))

constexpr int TM = 64, TN = 128, KC = 32;                 // macro-tile, k per ring slot
constexpr int kSlotFloats = (TM + TN) * KC;               // 24 KB
constexpr int kPieces = (TM + TN) / 8;                    // 1-KiB pieces per slot (8 rows x 128 B each): 24
constexpr int kThreads = 256;

// LDS image of a slot: row rho (0..63 the X rows, 64..191 the W rows) is 128 B = 8 pieces of 16 B (4 k each); logical piece lp sits at
// position lp ^ ((rho >> 1) & 7): the sixteen lanes of every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS) then touch sixteen
// different 16-byte bank groups.  W rows are PERMUTED on the way in: LDS row 64 + 64 c + 16 u + n holds W row j0 + 64 c + 4 n + u, so
// that MFMA tile u of a wave covers the columns {4 n + u} and a lane's four accumulators of a row are four consecutive columns
// (16-byte stores straight from registers, no transpose).
template <int NSLOT, bool BF6>
__global__ __launch_bounds__(kThreads) void k_tile_fwd(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ P,
                                                       int M, int N, int K, int S, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, l = tid & 63, n = l & 15, g = l >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware mapping: blocks b, b + 8 share an XCD (round-robin dealing).  XCD x owns the reduction slice x % S and a contiguous
    // share of the column tiles for all row tiles: its L2 holds one k-slice of X and of its W rows, nothing twice.
    const int b = (int)blockIdx.x, x = b & 7, i = b >> 3;
    int ks, tn, tm;
    if (S > 0) {
        const int cgroups = 8 / S, tn_per = tiles_n / cgroups;       // host guarantees divisibility
        ks = x % S;  tn = (x / S) * tn_per + i / tiles_m;  tm = i % tiles_m;
    } else {
        // S = 0: no split, any tile count -- XCD x takes the x-th contiguous eighth of the tiles (column tile major: the workgroups of an
        // XCD share their W rows in its L2); the grid is 8 x ceil(tiles / 8), surplus workgroups leave
        const int total = tiles_m * tiles_n, per_x = (total + 7) / 8, t = x * per_x + i;
        if (i >= per_x || t >= total) return;
        ks = 0;  tn = t / tiles_m;  tm = t % tiles_m;
    }
    if (S == 0) S = 1;
    const int i0 = tm * TM, j0 = tn * TN;
    const int kbeg = ks * (K / S), nk = (K / S) / KC;
    const int tr = wv & 1, tc = wv >> 1;                   // this wave's 32 x 64 sub-tile

    auto issue = [&](int it) {
        float* slot = smem + (it % NSLOT) * kSlotFloats;
        const int k0 = kbeg + it * KC;
#pragma unroll
        for (int q = 0; q < kPieces / 4; ++q) {
            const int piece = wv + 4 * q;                  // wave-uniform
            const int rho = piece * 8 + (l >> 3);
            const int lp = (l & 7) ^ ((rho >> 1) & 7);
            const float* src;
            if (piece < TM / 8) src = X + (size_t)min(i0 + rho, M - 1) * K + k0 + 4 * lp;
            else {
                const int r = rho - TM, c = r >> 6, u = (r >> 4) & 3, nn = r & 15;
                src = W + (size_t)min(j0 + 64 * c + 4 * nn + u, N - 1) * K + k0 + 4 * lp;
            }
            __builtin_amdgcn_global_load_lds((const GLB void*)src, (LDS void*)(slot + piece * 256), 16, 0, 0);
        }
    };
    f32x4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NSLOT - 1; ++it) if (it < nk) issue(it);
    for (int it = 0; it < nk; ++it) {
        // slot `it` has landed for this wave once at most the pieces of the NSLOT - 2 younger slots are outstanding
        if (it + NSLOT - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 2) * (kPieces / 4)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + NSLOT - 1 < nk) issue(it + NSLOT - 1);
        const LDS float* slot = (const LDS float*)(smem + (it % NSLOT) * kSlotFloats);
        auto frag = [&](int rho, int kk) {
            return *reinterpret_cast<const LDS f32x4*>(slot + rho * 32 + 4 * ((4 * kk + g) ^ ((rho >> 1) & 7)));
        };
        if constexpr (!BF6) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f32x4 a[2], bw[4];
#pragma unroll
                for (int t = 0; t < 2; ++t) a[t] = frag(32 * tr + 16 * t + n, kk);
#pragma unroll
                for (int u = 0; u < 4; ++u) bw[u] = frag(TM + 64 * tc + 16 * u + n, kk);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            acc[4 * t + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][q], bw[u][q], acc[4 * t + u], 0, 0, 0);
            }
        } else {
            // six bf16 products per fp32 product (x = hi + mid + lo, each subtraction exact).  v_mfma_f32_16x16x32_bf16 takes 8 k per
            // lane: the lane's two 16-byte pieces of the slot (k = 4 g .. and 16 + 4 g ..; any k order both operands share will do)
            auto split = [](f32x4 v0, f32x4 v1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float x = c < 4 ? v0[c & 3] : v1[c & 3];
                    const __bf16 h = (__bf16)x;  const float r1 = x - (float)h;
                    const __bf16 m = (__bf16)r1;  const float r2 = r1 - (float)m;
                    hi[c] = h;  mid[c] = m;  lo[c] = (__bf16)r2;
                }
            };
            bf16x8 ah[2], am[2], al[2], bh[4], bm[4], bl[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) split(frag(32 * tr + 16 * t + n, 0), frag(32 * tr + 16 * t + n, 1), ah[t], am[t], al[t]);
#pragma unroll
            for (int u = 0; u < 4; ++u) split(frag(TM + 64 * tc + 16 * u + n, 0), frag(TM + 64 * tc + 16 * u + n, 1), bh[u], bm[u], bl[u]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f32x4 c = acc[4 * t + u];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bh[u], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bl[u], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[t], bm[u], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[t], bh[u], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bm[u], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bh[u], c, 0, 0, 0);
                    acc[4 * t + u] = c;
                }
        }
    }
    // acc[4 t + u][e] = (row 32 tr + 16 t + 4 g + e, column 64 tc + 4 n + u) of the tile
    float* out = P + (size_t)ks * M * N;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = i0 + 32 * tr + 16 * t + 4 * g + e, col = j0 + 64 * tc + 4 * n;
            if (row < M && col < N)
                *reinterpret_cast<f32x4*>(out + (size_t)row * N + col) = (f32x4){acc[4 * t][e], acc[4 * t + 1][e], acc[4 * t + 2][e], acc[4 * t + 3][e]};
        }
}

// Y = relu(sum_s P[s] + bias): the fixed-order combine as its own launch
__global__ __launch_bounds__(256) void k_combine(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ Y, int MN4, int N, int S) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= MN4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(P)[i];
    for (int s = 1; s < S; ++s) v += reinterpret_cast<const f32x4*>(P)[(size_t)s * MN4 + i];
    const f32x4 bb = reinterpret_cast<const f32x4*>(bias)[i % (N / 4)];
    v += bb;
    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
    reinterpret_cast<f32x4*>(Y)[i] = v;
}

// something that evicts the caches between two measured launches (512 MB written: larger than L2 + Infinity Cache)
__global__ void k_flush(float4* p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename F> float time_us(F f, int iters = 50) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));  CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

template <int NSLOT> void run(bool cold);
int main(int argc, char** argv) {
    const bool cold = argc > 1 && !strcmp(argv[1], "cold");
    printf("== ring of 4 slots (96 KB: one workgroup per CU)\n");
    run<4>(cold);
    printf("== ring of 3 slots (72 KB: two workgroups per CU)\n");
    run<3>(cold);
    return 0;
}
template <int NSLOT> void run(bool cold) {
    struct Shape { int M, K, N, S; };
    // S = 0: no split and the general tile mapping (what a production kernel for tall inputs would run)
    const Shape shapes[] = {{512, 1024, 1024, 4}, {512, 1024, 1024, 2}, {512, 1024, 1024, 8}, {512, 1024, 512, 8}, {512, 128, 1024, 4}, {2048, 1024, 1024, 1},
                            {2048, 1024, 1024, 0}, {2048, 1024, 512, 0}, {2048, 128, 1024, 0}, {1024, 1024, 1024, 0}, {1100, 1024, 1024, 0},
                            {8192, 1024, 1024, 0}, {8192, 1024, 512, 0}, {8192, 128, 1024, 0}, {8192, 256, 128, 0}};
    float4* junk = nullptr;
    const size_t junk_n4 = (512u << 20) / 16;
    if (cold) CK(hipMalloc(&junk, junk_n4 * 16));
    for (const Shape& sh : shapes) {
        const int M = sh.M, K = sh.K, N = sh.N, S = sh.S;
        const int tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
        if (S > 0 && ((K / S) % KC || 8 % S || tiles_n % (8 / S) || (tiles_m * tiles_n * S) % 8)) { printf("shape %d %d %d S=%d: skipped (mapping)\n", M, K, N, S); continue; }
        if (S == 0 && K % KC) { printf("shape %d %d %d: skipped (K %% 32)\n", M, K, N); continue; }
        std::vector<float> hX((size_t)M * K), hW((size_t)N * K), hb(N);
        uint32_t st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
        for (auto& v : hX) v = rnd();
        for (auto& v : hW) v = rnd() * 0.1f;
        for (auto& v : hb) v = rnd();
        float *X, *W, *bias, *P, *Y, *Yl;
        CK(hipMalloc(&X, hX.size() * 4));  CK(hipMalloc(&W, hW.size() * 4));  CK(hipMalloc(&bias, N * 4));
        CK(hipMalloc(&P, (size_t)(S ? S : 1) * M * N * 4));  CK(hipMalloc(&Y, (size_t)M * N * 4));  CK(hipMalloc(&Yl, (size_t)M * N * 4));
        CK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
        const int grid = S ? tiles_m * tiles_n * S : 8 * ((tiles_m * tiles_n + 7) / 8), MN4 = M * N / 4;
        const size_t lds = (size_t)NSLOT * kSlotFloats * 4;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_fwd<NSLOT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_fwd<NSLOT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fn_act_epilogue act{};
        act.y = Yl;  act.p = 0.f;  act.relu = 1;
        auto flush = [&]() { if (cold) hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, junk, junk_n4); };
        auto lib = [&]() { flush();  if (fn_dense_fwd_f32(X, W, bias, Yl, M, K, N, &act, nullptr)) { printf("lib: %s\n", fn_last_error()); exit(1); } };
        auto mainf = [&]() { flush();  hipLaunchKernelGGL((k_tile_fwd<NSLOT, false>), dim3(grid), dim3(kThreads), lds, 0, X, W, P, M, N, K, S, tiles_m, tiles_n); };
        auto main6 = [&]() { flush();  hipLaunchKernelGGL((k_tile_fwd<NSLOT, true>), dim3(grid), dim3(kThreads), lds, 0, X, W, P, M, N, K, S, tiles_m, tiles_n); };
        auto comb = [&]() { hipLaunchKernelGGL(k_combine, dim3((MN4 + 255) / 256), dim3(256), 0, 0, P, bias, Y, MN4, N, S ? S : 1); };
        // ---- errors against fp64 on a sample of entries
        auto check = [&](const char* what, const float* dev) {
            std::vector<float> h((size_t)M * N);
            CK(hipMemcpy(h.data(), dev, h.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0, big = 0;
            for (int s = 0; s < 4096; ++s) {
                const int r = (s * 7919) % M, c = (s * 104729) % N;
                double ref = hb[c];
                for (int k = 0; k < K; ++k) ref += (double)hX[(size_t)r * K + k] * hW[(size_t)c * K + k];
                ref = ref > 0 ? ref : 0;
                worst = std::max(worst, std::fabs(ref - h[(size_t)r * N + c]));
                big = std::max(big, std::fabs(ref));
            }
            printf("   %-34s max |err| vs fp64 = %.3e (largest |y| %.2f)\n", what, worst, big);
        };
        lib();  CK(hipDeviceSynchronize());  check("library fn_dense_fwd_f32", Yl);
        mainf();  comb();  CK(hipDeviceSynchronize());  check("LDS tiles, fp32 MFMA + combine", Y);
        main6();  comb();  CK(hipDeviceSynchronize());  check("LDS tiles, six bf16 products + combine", Y);
        CK(hipDeviceSynchronize());
        const float t_fl = cold ? time_us([&]() { flush(); }) : 0.f;
        const float t_lib = time_us(lib) - t_fl, t_main = time_us(mainf) - t_fl, t_6 = time_us(main6) - t_fl;
        const float t_both = time_us([&]() { mainf();  comb(); }) - t_fl, t_comb = time_us(comb);
        printf("M=%d K=%d N=%d S=%d (%d workgroups, %s): library %.2f us | tiles fp32 %.2f us | tiles bf16x6 %.2f us | combine %.2f us | tiles + combine %.2f us   [%.1f GFLOP]\n",
               M, K, N, S, grid, cold ? "caches flushed before every launch" : "back to back, hot", t_lib, t_main, t_6, t_comb, t_both, 2.0 * M * K * N * 1e-9);
        CK(hipFree(X));  CK(hipFree(W));  CK(hipFree(bias));  CK(hipFree(P));  CK(hipFree(Y));  CK(hipFree(Yl));
    }
    if (junk) CK(hipFree(junk));
}
