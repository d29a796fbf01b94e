// How much does a device-side grid barrier cost on MI355X, against a kernel boundary inside a hipGraph?
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/grid_barrier_probe.hip -o tools/probe/grid_barrier_probe.bin
// Persistent grid (blocks <= resident capacity), monotonically increasing arrival counter, device-scope release/acquire
// around it (cross-XCD visibility of the data the phases exchange).  Each phase writes a value per block and reads its
// neighbour's value of the previous phase, so a broken barrier or missing visibility shows up as a wrong checksum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                       // device scope by default for global memory
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

__global__ void k_phases(unsigned* counter, float* buf, int phases, float* out) {
    const int nb = gridDim.x, b = blockIdx.x;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        if (threadIdx.x == 0) buf[(p & 1) * nb + b] = (float)(p + b);
        grid_barrier(counter, (unsigned)(p + 1) * nb);
        acc += __builtin_nontemporal_load(&buf[(p & 1) * nb + (b + 97) % nb]);      // another block's value (other XCD: block ids round-robin over XCDs)
    }
    if (threadIdx.x == 0) out[b] = acc;
}
__global__ void k_one(float* buf, int p, int nb, float* out) {                      // the same phase as its own kernel
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        if (p > 0) out[b] += buf[((p - 1) & 1) * nb + (b + 97) % nb];
        buf[(p & 1) * nb + b] = (float)(p + b);
    }
}

int main() {
    const int phases = 64;
    for (int nb : {256, 512, 1024, 2048}) {
        unsigned* counter;  float *buf, *out;
        CK(hipMalloc(&counter, 4));  CK(hipMalloc(&buf, 2 * nb * 4));  CK(hipMalloc(&out, nb * 4));
        hipStream_t st;  CK(hipStreamCreate(&st));
        hipEvent_t e0, e1;  CK(hipEventCreate(&e0));  CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(counter, 0, 4, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_phases, dim3(nb), dim3(256), 0, st, counter, buf, phases, out);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;  CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<float> h(nb);
        CK(hipMemcpy(h.data(), out, nb * 4, hipMemcpyDeviceToHost));
        double want = 0, got = 0;
        for (int b = 0; b < nb; ++b) { got += h[b];  for (int p = 0; p < phases; ++p) want += p + (b + 97) % nb; }
        // same phases as a hipGraph of kernels
        hipGraph_t g;  hipGraphExec_t ge;
        CK(hipMemsetAsync(out, 0, nb * 4, st));
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_one, dim3(nb), dim3(256), 0, st, buf, p, nb, out);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float gbest = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;  CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < gbest) gbest = ms;
        }
        printf("blocks %5d: grid barrier %.2f us/phase (checksum %s), hipGraph kernel boundary %.2f us/phase\n", nb,
               best * 1e3f / phases, got == want ? "ok" : "WRONG", gbest * 1e3f / phases);
    }
    return 0;
}
