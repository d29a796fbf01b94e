// Dev probe (round 6): what bounds a SHORT streaming launch on operands that are not cache-resident?
//
// profiles/r06_hbm_cold_stream.md: a torch element-wise pass over 64 MB in + 64 MB out takes 19.6 us when its operands sit in the
// memory-side cache and 34.8 us behind a 1-GB evicting pass (3.85 TB/s) -- and the step's large launches all sit at 3 - 3.9 TB/s of
// moved bytes.  If that ceiling is Little's law (bytes in flight per CU x HBM latency) rather than the memory system, more
// independent loads per lane must lift it.  This probe times, with device timestamps (first workgroup in -> last workgroup out),
//   copy<U>   : U independent 16-byte loads per lane, then U stores, one pass per workgroup (grid = n / (256 U))
//   copyP<U>  : the same body as a persistent grid-stride loop (256 CUs x 8 workgroups)
//   read<U>   : loads only (a sum per lane, one store per workgroup)
// hot (same buffers again), cold behind a READ of 1 GB, cold behind a FILL of 1 GB (dirty lines in the memory-side cache).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/cold_stream_probe.bin tools/probe/cold_stream_probe.hip
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

struct Stamp { unsigned long long t0, t1; };
// one pair per workgroup, reduced on the host (atomics on one address serialise the workgroups: 11 ns each)
__device__ __forceinline__ void stamp_in(Stamp* s) {
    if (threadIdx.x == 0) s[blockIdx.x].t0 = wall_clock64();
}
__device__ __forceinline__ void stamp_out(Stamp* s) {
    __syncthreads();
    if (threadIdx.x == 0) s[blockIdx.x].t1 = wall_clock64();
}

template <int U> __global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n16, Stamp* s) {
    stamp_in(s);
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[base + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) b[base + (size_t)u * 256] = v[u];
    stamp_out(s);
}
template <int U> __global__ __launch_bounds__(256) void k_copy_p(const float4* __restrict__ a, float4* __restrict__ b, size_t n16, Stamp* s) {
    stamp_in(s);
    const size_t step = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n16; base += step) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[base + (size_t)u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) b[base + (size_t)u * 256] = v[u];
    }
    stamp_out(s);
}
template <int U> __global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n16, Stamp* s) {
    stamp_in(s);
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[base + (size_t)u * 256];
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    if (acc == 12345.678f) out[blockIdx.x] = acc;
    stamp_out(s);
}
__global__ __launch_bounds__(256) void k_evict_read(const float4* __restrict__ a, float* out, size_t n16) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const float4 v = a[i];  acc += v.x + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_evict_fill(float4* __restrict__ a, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

static float4 *g_big, *g_a, *g_b;
static float* g_out;
static Stamp* g_s;
static size_t g_big16;

enum Mode { HOT, COLD_R, COLD_W };
constexpr unsigned kMaxBlocks = 1u << 16;
template <typename F> static double run(Mode m, unsigned blocks, F launch) {
    std::vector<double> us;
    static std::vector<Stamp> h(kMaxBlocks);
    for (int it = 0; it < 9; ++it) {
        if (m == COLD_R) hipLaunchKernelGGL(k_evict_read, dim3(4096), dim3(256), 0, 0, g_big, g_out, g_big16);
        if (m == COLD_W) hipLaunchKernelGGL(k_evict_fill, dim3(4096), dim3(256), 0, 0, g_big, g_big16);
        launch();
        CK(hipMemcpy(h.data(), g_s, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (unsigned b = 0; b < blocks; ++b) { t0 = std::min(t0, h[b].t0);  t1 = std::max(t1, h[b].t1); }
        if (it >= 2) us.push_back((double)(t1 - t0) * 0.01);          // wall_clock64: 100 MHz
    }
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
}

template <int U> static void row(size_t mb) {
    const size_t n16 = mb * 1024 * 1024 / 16;
    const unsigned grid = (unsigned)(n16 / (256 * U));
    double c[3], p[3], r[3];
    for (int m = 0; m < 3; ++m) {
        c[m] = run((Mode)m, grid, [&] { hipLaunchKernelGGL(k_copy<U>, dim3(grid), dim3(256), 0, 0, g_a, g_b, n16, g_s); });
        p[m] = run((Mode)m, 2048, [&] { hipLaunchKernelGGL(k_copy_p<U>, dim3(2048), dim3(256), 0, 0, g_a, g_b, n16, g_s); });
        r[m] = run((Mode)m, grid, [&] { hipLaunchKernelGGL(k_read<U>, dim3(grid), dim3(256), 0, 0, g_a, g_out, n16, g_s); });
    }
    const double mbs = mb * 1.048576;
    auto f = [&](double us, double mult) { static char buf[16][48]; static int k = 0; char* o = buf[k++ & 15]; snprintf(o, 48, "%6.2f (%4.2f)", us, mult * mbs / us); return o; };
    printf("| %3zu MB | %2d | %s | %s | %s | %s | %s | %s | %s | %s | %s |\n", mb, U, f(c[0], 2), f(c[1], 2), f(c[2], 2), f(p[0], 2), f(p[1], 2), f(p[2], 2),
           f(r[0], 1), f(r[1], 1), f(r[2], 1));
}

int main() {
    g_big16 = (size_t)1024 * 1024 * 1024 / 16;
    CK(hipMalloc(&g_big, g_big16 * 16));
    CK(hipMalloc(&g_a, (size_t)128 << 20));
    CK(hipMalloc(&g_b, (size_t)128 << 20));
    CK(hipMalloc(&g_out, 1 << 22));
    CK(hipMalloc(&g_s, sizeof(Stamp) * kMaxBlocks));
    CK(hipMemset(g_big, 0, g_big16 * 16));
    CK(hipMemset(g_a, 0, (size_t)128 << 20));
    CK(hipMemset(g_b, 0, (size_t)128 << 20));
    printf("us (TB/s moved), device timestamps first workgroup in -> last out, median of 7; U = independent 16-byte loads per lane\n");
    printf("| size (in; copy: + out) | U | copy hot | copy cold(read-evicted) | copy cold(fill-evicted) | persistent copy hot | cold(r) | cold(w) | read hot | cold(r) | cold(w) |\n");
    printf("|---|---|---|---|---|---|---|---|---|---|---|\n");
    for (size_t mb : {32, 64, 128}) {
        row<1>(mb);
        row<2>(mb);
        row<4>(mb);
        row<8>(mb);
        row<16>(mb);
    }
    return 0;
}
