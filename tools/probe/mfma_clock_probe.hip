// Dev probe (round 6): what clock does the chip hold under a back-to-back fp32-MFMA loop, in bursts as short as the step's launches?
//
// profiles/r06_wgrad_register_form_ab.txt: two unrelated K = 128 weight-gradient kernels take the same time, 58 % of the nominal fp32
// matrix peak (157.3 TF = 256 FLOP per clock and CU at 2.4 GHz).  If the chip lowers its clock under that load (MI355X_MICROARCH.md,
// "DVFS give-back") the pipe may already be full.  This probe runs v_mfma_f32_16x16x4_f32 back to back on random operands -- 16
// independent accumulators per wave, W waves per SIMD, every CU -- for a chosen number of MFMAs per wave and reports, per launch:
//   wall time from device real-time stamps (s_memrealtime, 100 MHz), core cycles from s_memtime, their quotient (the clock held), TF/s.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_clock_probe.bin tools/probe/mfma_clock_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long c0, c1, r0, r1; };

__global__ __launch_bounds__(512) void k_mfma(const float* __restrict__ seed, float* __restrict__ out, int iters, Stamp* st) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = seed[(threadIdx.x * 4 + i) & 1023];  b[i] = seed[(threadIdx.x * 4 + i + 512) & 1023]; }
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime();  r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[4 * q + p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[p], acc[4 * q + p], 0, 0, 0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Stamp s;
        s.c0 = c0;  s.r0 = r0;  s.c1 = __builtin_amdgcn_s_memtime();  s.r1 = __builtin_amdgcn_s_memrealtime();
        st[blockIdx.x] = s;
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (v == 12345.678f) out[blockIdx.x * 64 + lane] = v;
}

int main() {
    const int blocks = 256;
    float *seed, *out;
    Stamp* st;
    CK(hipMalloc(&seed, 4096));
    CK(hipMalloc(&out, blocks * 64 * 4));
    CK(hipMalloc(&st, sizeof(Stamp) * blocks));
    std::vector<float> h(1024);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice));
    printf("v_mfma_f32_16x16x4_f32 back to back, random operands, 256 workgroups (one per CU), 16 independent accumulators per wave\n");
    printf("| waves per SIMD | MFMAs per wave | wall us (device real-time) | core cycles | clock held GHz | TF/s | of the 2.4-GHz peak |\n|---|---|---|---|---|---|---|\n");
    for (int threads : {256, 512}) {
        for (int iters : {64, 256, 1024, 16384, 131072}) {
            std::vector<double> us, ghz;
            for (int rep = 0; rep < 7; ++rep) {
                hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, seed, out, iters, st);
                std::vector<Stamp> hs(blocks);
                CK(hipMemcpy(hs.data(), st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost));
                unsigned long long r0 = ~0ull, r1 = 0;
                std::vector<double> g;
                for (auto& s : hs) {
                    r0 = std::min(r0, s.r0);  r1 = std::max(r1, s.r1);
                    if (s.r1 > s.r0) g.push_back((double)(s.c1 - s.c0) / ((double)(s.r1 - s.r0) * 10.0));       // cycles per ns
                }
                std::sort(g.begin(), g.end());
                if (rep >= 2) { us.push_back((double)(r1 - r0) * 0.01);  ghz.push_back(g[g.size() / 2]); }
            }
            std::sort(us.begin(), us.end());  std::sort(ghz.begin(), ghz.end());
            const double t = us[us.size() / 2], c = ghz[ghz.size() / 2];
            const double flop = (double)blocks * (threads / 64) * iters * 16.0 * 2048.0;
            printf("| %d | %d | %.1f | %.0f | %.2f | %.1f | %.2f |\n", threads / 256, iters * 16, t, t * c * 1e3, c, flop / t * 1e-6, flop / t * 1e-6 / 157.3);
        }
    }
    return 0;
}
