// Feasibility probe for the molecule-resident layer kernel (HISTORY.md section 9.1): the bond-graph attention level of an
// ESOL-shape batch with one workgroup per molecule and the molecule's node rows staged in LDS, against the per-level
// kernel's 15.4 us for the same work.  Synthetic: 512 molecules x 55 directed bonds, 7 in-edges per bond from the same
// molecule, H = 4 heads, D = 128.  Not part of the library.
// build: hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probe/molecule_probe.hip -o tools/probe/molecule_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NB = 55, DEG = 7, H = 4, D = 128, MAXB = 64;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// one block = one molecule: rows [r0, r0 + nb), all sources inside the same range.  STAGE_EDGES: the molecule's edge lists
// (sources, edge terms) and row extents go to LDS with the node rows in ONE round trip; the rows then run from LDS only.
template <bool STAGE_EDGES>
__global__ __launch_bounds__(256) void k_mol_level(const float* __restrict__ h, const float* __restrict__ s_dst,
                                                   const float* __restrict__ s_src, const float* __restrict__ s_edge,
                                                   const int* __restrict__ rowptr, const int* __restrict__ src,
                                                   float* __restrict__ out, float* __restrict__ p_out, int nb) {
    __shared__ float4 sH[MAXB * 32];
    __shared__ float sS[MAXB * H], sD[MAXB * H];
    __shared__ int sRp[MAXB + 1];
    __shared__ int sSrc[MAXB * (DEG + 1)];
    __shared__ float4 sE[MAXB * (DEG + 1)];
    const int r0 = blockIdx.x * nb, tid = threadIdx.x;
    const int e0 = rowptr[r0], e1 = rowptr[r0 + nb];                 // the only dependent round trip
    for (int i = tid; i < nb * 32; i += 256) sH[i] = ld4(h + (size_t)r0 * D + 4 * i);
    for (int i = tid; i < nb * H; i += 256) { sS[i] = s_src[(size_t)r0 * H + i]; sD[i] = s_dst[(size_t)r0 * H + i]; }
    if (STAGE_EDGES) {
        for (int i = tid; i <= nb; i += 256) sRp[i] = rowptr[r0 + i] - e0;
        for (int i = tid; i < e1 - e0; i += 256) { sSrc[i] = src[e0 + i] - r0; sE[i] = ld4(s_edge + (size_t)(e0 + i) * H); }
    }
    __syncthreads();
    const int hw = tid >> 5, l = tid & 31, head = l >> 3;
    for (int r = hw; r < nb; r += 8) {
        const int row = r0 + r;
        const int beg = STAGE_EDGES ? sRp[r] : rowptr[row] - e0, deg = (STAGE_EDGES ? sRp[r + 1] : rowptr[row + 1] - e0) - beg;
        const float sd = sD[r * H + head];
        float mx = -1e30f;
        float lg[DEG + 1];
        int js[DEG + 1];
#pragma unroll
        for (int e = 0; e < DEG + 1; ++e) {
            if (e < deg) {
                const int j = STAGE_EDGES ? sSrc[beg + e] : src[e0 + beg + e] - r0;
                const float se = STAGE_EDGES ? reinterpret_cast<const float*>(&sE[beg + e])[head] : s_edge[(size_t)(e0 + beg + e) * H + head];
                float z = sd + sS[j * H + head] + se;
                z = z > 0.f ? z : 0.2f * z;
                lg[e] = z;
                js[e] = j;
                mx = fmaxf(mx, z);
            }
        }
        float den = 0.f;
#pragma unroll
        for (int e = 0; e < DEG + 1; ++e)
            if (e < deg) { lg[e] = __expf(lg[e] - mx); den += lg[e]; }
        const float inv = 1.f / den;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < DEG + 1; ++e) {
            if (e < deg) {
                const float p = lg[e] * inv;
                const float4 v = sH[js[e] * 32 + l];
                acc.x += p * v.x; acc.y += p * v.y; acc.z += p * v.z; acc.w += p * v.w;
                if ((l & 7) == 0) p_out[(size_t)(e0 + beg + e) * H + head] = p;
            }
        }
        st4(out + (size_t)row * D + 4 * l, acc);
    }
}

int main() {
    const int B = 512, N = B * NB, M = N * DEG;
    std::vector<float> h((size_t)N * D), sd((size_t)N * H), ss((size_t)N * H), se((size_t)M * H);
    std::vector<int> rp(N + 1), sr(M);
    srand(1);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : sd) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : ss) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : se) v = (rand() % 2001 - 1000) * 1e-3f;
    for (int i = 0; i <= N; ++i) rp[i] = i * DEG;
    for (int i = 0; i < N; ++i)
        for (int e = 0; e < DEG; ++e) sr[i * DEG + e] = (i / NB) * NB + rand() % NB;
    float *dh, *dsd, *dss, *dse, *dout, *dp;
    int *drp, *dsr;
    hipMalloc(&dh, h.size() * 4); hipMalloc(&dsd, sd.size() * 4); hipMalloc(&dss, ss.size() * 4); hipMalloc(&dse, se.size() * 4);
    hipMalloc(&dout, h.size() * 4); hipMalloc(&dp, se.size() * 4); hipMalloc(&drp, rp.size() * 4); hipMalloc(&dsr, sr.size() * 4);
    hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsd, sd.data(), sd.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dss, ss.data(), ss.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dse, se.data(), se.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(drp, rp.data(), rp.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsr, sr.data(), sr.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0.f;
    const int iters = 50;
    for (int variant = 0; variant < 2; ++variant) {
        auto launch = [&] {
            if (variant) hipLaunchKernelGGL(k_mol_level<true>, dim3(B), dim3(256), 0, 0, dh, dsd, dss, dse, drp, dsr, dout, dp, NB);
            else hipLaunchKernelGGL(k_mol_level<false>, dim3(B), dim3(256), 0, 0, dh, dsd, dss, dse, drp, dsr, dout, dp, NB);
        };
        for (int it = 0; it < 5; ++it) launch();
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int it = 0; it < iters; ++it) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("variant %d (%s): %.2f us per launch\n", variant, variant ? "edge lists staged in LDS" : "edge lists from global", ms * 1000 / iters);
    }
    // checksum against a host evaluation of a few rows
    std::vector<float> o((size_t)N * D);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    double err = 0;
    for (int row : {0, 777, N - 1}) {
        for (int c = 0; c < D; ++c) {
            const int hd = c / 32;
            double mx = -1e30, den = 0, acc = 0, z[DEG];
            for (int e = 0; e < DEG; ++e) {
                double v = sd[(size_t)row * H + hd] + ss[(size_t)sr[row * DEG + e] * H + hd] + se[(size_t)(row * DEG + e) * H + hd];
                z[e] = v > 0 ? v : 0.2 * v;
                mx = z[e] > mx ? z[e] : mx;
            }
            for (int e = 0; e < DEG; ++e) { z[e] = exp(z[e] - mx); den += z[e]; }
            for (int e = 0; e < DEG; ++e) acc += z[e] / den * h[(size_t)sr[row * DEG + e] * D + c];
            err = fmax(err, fabs(acc - o[(size_t)row * D + c]));
        }
    }
    const double bytes = 4.0 * ((N + 1) + M + (double)M * H + 2.0 * N * H + 2.0 * N * D + (double)M * H);
    printf("molecule-resident bond level: %.2f us per launch (%d rows, %d edges), %.0f GB/s algorithmic, max err %.2e\n",
           ms * 1000 / iters, N, M, bytes / (ms * 1e-3 / iters) / 1e9, err);
    return 0;
}
