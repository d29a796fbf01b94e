// Probe (not product code): Y[M,128] = X[M,128] * Bt[128,128] + bias on the matrix cores of gfx950, two ways:
//   (a) v_mfma_f32_16x16x4_f32 -- the shape of fragnet_hip.hip's k_linear128 (4-wave blocks of 64 rows x 64 columns, the operand
//       tile in LDS, interleaved-k A rows straight from global memory);
//   (b) the same product from bf16 pieces: every fp32 operand split into three bf16 terms (x = hi + mid + lo, each subtraction
//       exact), six v_mfma_f32_16x16x32_bf16 per fp32 product (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) into an fp32
//       accumulator.  The weights are split once on the host (per step in a real engine), the rows in registers.
// Prints the time per launch (back-to-back launches, HIP events) and the error of both against an fp64 CPU product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/gemm_bf16x6_probe.hip -o tools/probe/gemm_bf16x6_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256, kRows = 64;

// ---------------- (a) fp32 MFMA ----------------
__global__ __launch_bounds__(kThreads) void k_f32(const float* __restrict__ X, const float* __restrict__ Bt, const float* __restrict__ bias,
                                                  float* __restrict__ Y, int M) {
    extern __shared__ __attribute__((aligned(16))) float sB[];      // [128 k][64 columns of this half]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int wc = blockIdx.x & 1, tile = blockIdx.x >> 1;
    float4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = tid + q * kThreads, k = idx >> 4, n4 = idx & 15;
        v[q] = *reinterpret_cast<const float4*>(Bt + (size_t)k * 128 + 64 * wc + n4 * 4);
    }
    int row = tile * kRows + w * 16 + i;
    row = row < M ? row : M - 1;
    float a[32];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float4 x = *reinterpret_cast<const float4*>(X + (size_t)row * 128 + 16 * s + 4 * kq);
        a[4 * s] = x.x; a[4 * s + 1] = x.y; a[4 * s + 2] = x.z; a[4 * s + 3] = x.w;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = tid + q * kThreads, k = idx >> 4, n4 = idx & 15;
        *reinterpret_cast<float4*>(sB + k * 64 + n4 * 4) = v[q];
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int k = 16 * (s >> 2) + 4 * kq + (s & 3);
        const float4 b = *reinterpret_cast<const float4*>(sB + k * 64 + 4 * i);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b.w, acc[3], 0, 0, 0);
    }
    const int col = 64 * wc + 4 * i;
    const float4 bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = tile * kRows + w * 16 + kq * 4 + r;
        if (orow < M)
            *reinterpret_cast<float4*>(Y + (size_t)orow * 128 + col) =
                make_float4(acc[0][r] + bv.x, acc[1][r] + bv.y, acc[2][r] + bv.z, acc[3][r] + bv.w);
    }
}

// ---------------- (a') fp32 MFMA, both column halves per workgroup: the rows are loaded ONCE and stay in registers while the
// operand tile is restaged for the second half (half the workgroups, each twice as long, one more barrier pair)
__global__ __launch_bounds__(kThreads) void k_f32_both(const float* __restrict__ X, const float* __restrict__ Bt, const float* __restrict__ bias,
                                                       float* __restrict__ Y, int M) {
    extern __shared__ __attribute__((aligned(16))) float sB[];      // [128 k][64 columns of the current half]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x;
    float4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = tid + q * kThreads, k = idx >> 4, n4 = idx & 15;
        v[q] = *reinterpret_cast<const float4*>(Bt + (size_t)k * 128 + n4 * 4);
    }
    int row = tile * kRows + w * 16 + i;
    row = row < M ? row : M - 1;
    float a[32];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float4 x = *reinterpret_cast<const float4*>(X + (size_t)row * 128 + 16 * s + 4 * kq);
        a[4 * s] = x.x; a[4 * s + 1] = x.y; a[4 * s + 2] = x.z; a[4 * s + 3] = x.w;
    }
#pragma unroll
    for (int wc = 0; wc < 2; ++wc) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = tid + q * kThreads, k = idx >> 4, n4 = idx & 15;
            *reinterpret_cast<float4*>(sB + k * 64 + n4 * 4) = v[q];
        }
        __syncthreads();
        if (wc == 0) {      // the second half's tile is requested while the first half multiplies
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = tid + q * kThreads, k = idx >> 4, n4 = idx & 15;
                v[q] = *reinterpret_cast<const float4*>(Bt + (size_t)k * 128 + 64 + n4 * 4);
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int k = 16 * (s >> 2) + 4 * kq + (s & 3);
            const float4 b = *reinterpret_cast<const float4*>(sB + k * 64 + 4 * i);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b.w, acc[3], 0, 0, 0);
        }
        const int col = 64 * wc + 4 * i;
        const float4 bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int orow = tile * kRows + w * 16 + kq * 4 + r;
            if (orow < M)
                *reinterpret_cast<float4*>(Y + (size_t)orow * 128 + col) =
                    make_float4(acc[0][r] + bv.x, acc[1][r] + bv.y, acc[2][r] + bv.z, acc[3][r] + bv.w);
        }
        if (wc == 0) __syncthreads();       // everyone is done reading the first tile
    }
}

// ---------------- (b) three bf16 terms per operand ----------------
// Bs: [2 column halves][3 terms][4 k-steps of 32][4 column tiles][64 lanes][8 bf16]; lane (i, kb) of tile t holds Bt[32 s + 8 kb + c][64 wc + 4 i + t]
constexpr int kBsHalf = 3 * 4 * 4 * 64 * 8;     // bf16 elements per column half (48 KB)

__device__ __forceinline__ void split3(const f32x8 x, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    hi = __builtin_convertvector(x, bf16x8);
    const f32x8 r1 = x - __builtin_convertvector(hi, f32x8);          // exact
    mid = __builtin_convertvector(r1, bf16x8);
    const f32x8 r2 = r1 - __builtin_convertvector(mid, f32x8);        // exact
    lo = __builtin_convertvector(r2, bf16x8);
}

template <int TERMS>      // 6: fp32-grade; 3: hi*hi + hi*mid + mid*hi (the classic "bf16x3", ~2^-16)
__global__ __launch_bounds__(kThreads) void k_b6(const float* __restrict__ X, const __bf16* __restrict__ Bs, const float* __restrict__ bias,
                                                 float* __restrict__ Y, int M) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bf16x8* sB = reinterpret_cast<bf16x8*>(smem);                   // [3][4][4][64] vectors of 8
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kb = lane >> 4;
    const int wc = blockIdx.x & 1, tile = blockIdx.x >> 1;
    const bf16x8* src = reinterpret_cast<const bf16x8*>(Bs + (size_t)wc * kBsHalf);
    constexpr int NV = kBsHalf / 8;              // 3072 vectors of 16 bytes
    bf16x8 v[NV / kThreads];
#pragma unroll
    for (int q = 0; q < NV / kThreads; ++q) v[q] = src[tid + q * kThreads];
    int row = tile * kRows + w * 16 + i;
    row = row < M ? row : M - 1;
    f32x8 a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float4 x0 = *reinterpret_cast<const float4*>(X + (size_t)row * 128 + 32 * s + 8 * kb);
        const float4 x1 = *reinterpret_cast<const float4*>(X + (size_t)row * 128 + 32 * s + 8 * kb + 4);
        a[s] = (f32x8){x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    }
#pragma unroll
    for (int q = 0; q < NV / kThreads; ++q) sB[tid + q * kThreads] = v[q];
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        bf16x8 ah, am, al;
        split3(a[s], ah, am, al);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 bh = sB[((0 * 4 + s) * 4 + t) * 64 + lane];
            const bf16x8 bm = sB[((1 * 4 + s) * 4 + t) * 64 + lane];
            if (TERMS == 6) {
                const bf16x8 bl = sB[((2 * 4 + s) * 4 + t) * 64 + lane];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[t], 0, 0, 0);
            }
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[t], 0, 0, 0);
        }
    }
    const int col = 64 * wc + 4 * i;
    const float4 bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = tile * kRows + w * 16 + kb * 4 + r;
        if (orow < M)
            *reinterpret_cast<float4*>(Y + (size_t)orow * 128 + col) =
                make_float4(acc[0][r] + bv.x, acc[1][r] + bv.y, acc[2][r] + bv.z, acc[3][r] + bv.w);
    }
}

static uint16_t bf16_rn(float x) {       // round to nearest even, as v_cvt_pk_bf16_f32
    uint32_t u;  memcpy(&u, &x, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16;  float f;  memcpy(&f, &u, 4);  return f; }

template <typename F> static float time_us(F launch, int iters) {
    for (int i = 0; i < 10; ++i) launch();
    hipEvent_t a, b;
    CK(hipEventCreate(&a));  CK(hipEventCreate(&b));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

int main() {
    const int Ms[] = {8192, 16384, 32768, 45056, 65536, 131072};
    const int Mmax = 131072;
    std::vector<float> hX((size_t)Mmax * 128), hB(128 * 128), hbias(128);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u;  return ((st >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    for (auto& x : hX) x = 4.f * rnd() * (1.f + 3.f * rnd());       // a few binades of magnitudes
    for (auto& x : hB) x = 0.3f * rnd();
    for (auto& x : hbias) x = rnd();
    // split the weights: Bs[wc][term][s][t][lane][c]
    std::vector<uint16_t> hBs((size_t)2 * kBsHalf);
    for (int wc = 0; wc < 2; ++wc)
        for (int s = 0; s < 4; ++s)
            for (int t = 0; t < 4; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int c = 0; c < 8; ++c) {
                        const int i = lane & 15, kb = lane >> 4;
                        const float x = hB[(size_t)(32 * s + 8 * kb + c) * 128 + 64 * wc + 4 * i + t];
                        const uint16_t h = bf16_rn(x);
                        const float r1 = x - bf16_f(h);
                        const uint16_t m = bf16_rn(r1);
                        const float r2 = r1 - bf16_f(m);
                        const uint16_t l = bf16_rn(r2);
                        const size_t base = (size_t)wc * kBsHalf + (((size_t)(s) * 4 + t) * 64 + lane) * 8 + c;
                        hBs[base + 0 * (kBsHalf / 3)] = h;
                        hBs[base + 1 * (kBsHalf / 3)] = m;
                        hBs[base + 2 * (kBsHalf / 3)] = l;
                    }
    float *dX, *dB, *dbias, *dY;
    __bf16* dBs;
    CK(hipMalloc(&dX, hX.size() * 4));  CK(hipMalloc(&dB, hB.size() * 4));  CK(hipMalloc(&dbias, 512));
    CK(hipMalloc(&dY, (size_t)Mmax * 128 * 4));  CK(hipMalloc(&dBs, hBs.size() * 2));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBs, hBs.data(), hBs.size() * 2, hipMemcpyHostToDevice));
    const size_t lds_f32 = 128 * 64 * 4, lds_b6 = (size_t)kBsHalf * 2;

    // accuracy on the first 512 rows against fp64
    const int R = 512;
    std::vector<double> ref((size_t)R * 128);
    for (int r = 0; r < R; ++r)
        for (int n = 0; n < 128; ++n) {
            double acc = hbias[n];
            for (int k = 0; k < 128; ++k) acc += (double)hX[(size_t)r * 128 + k] * (double)hB[(size_t)k * 128 + n];
            ref[(size_t)r * 128 + n] = acc;
        }
    std::vector<float> out((size_t)R * 128);
    auto report = [&](const char* name) {
        CK(hipMemcpy(out.data(), dY, out.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0.0, scale = 0.0;
        for (size_t j = 0; j < out.size(); ++j) { worst = fmax(worst, fabs(out[j] - ref[j]));  scale = fmax(scale, fabs(ref[j])); }
        printf("%-28s max |err| vs fp64 = %.3e  (largest |y| %.2f, relative %.2e)\n", name, worst, scale, worst / scale);
    };
    const int gridR = 2 * ((R + kRows - 1) / kRows);
    hipLaunchKernelGGL(k_f32, dim3(gridR), dim3(kThreads), lds_f32, 0, dX, dB, dbias, dY, R);
    CK(hipDeviceSynchronize());  report("fp32 MFMA");
    CK(hipMemset(dY, 0, (size_t)R * 512));
    hipLaunchKernelGGL(k_f32_both, dim3(gridR / 2), dim3(kThreads), lds_f32, 0, dX, dB, dbias, dY, R);
    CK(hipDeviceSynchronize());  report("fp32 MFMA, both halves");
    CK(hipMemset(dY, 0, (size_t)R * 512));
    hipLaunchKernelGGL(k_b6<6>, dim3(gridR), dim3(kThreads), lds_b6, 0, dX, dBs, dbias, dY, R);
    CK(hipDeviceSynchronize());  report("3 x 3 bf16 terms, 6 products");
    CK(hipMemset(dY, 0, (size_t)R * 512));
    hipLaunchKernelGGL(k_b6<3>, dim3(gridR), dim3(kThreads), lds_b6, 0, dX, dBs, dbias, dY, R);
    CK(hipDeviceSynchronize());  report("bf16x3 (3 products)");

    for (int M : Ms) {
        const int grid = 2 * ((M + kRows - 1) / kRows);
        const float t0 = time_us([&]() { hipLaunchKernelGGL(k_f32, dim3(grid), dim3(kThreads), lds_f32, 0, dX, dB, dbias, dY, M); }, 100);
        const float t6 = time_us([&]() { hipLaunchKernelGGL(k_b6<6>, dim3(grid), dim3(kThreads), lds_b6, 0, dX, dBs, dbias, dY, M); }, 100);
        const float t3 = time_us([&]() { hipLaunchKernelGGL(k_b6<3>, dim3(grid), dim3(kThreads), lds_b6, 0, dX, dBs, dbias, dY, M); }, 100);
        const float tb = time_us([&]() { hipLaunchKernelGGL(k_f32_both, dim3(grid / 2), dim3(kThreads), lds_f32, 0, dX, dB, dbias, dY, M); }, 100);
        printf("M=%7d workgroups=%5d   fp32 MFMA %7.2f us   six bf16 products %7.2f us   three %7.2f us   fp32, both halves per workgroup %7.2f us\n",
               M, grid, t0, t6, t3, tb);
    }
    return 0;
}
