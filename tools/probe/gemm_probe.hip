// Standalone timing probe for the projection GEMM kernels (dev tool, not part of the library).
// build: hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/probe/gemm_probe.hip -o /tmp/gemm_probe
#include "../../fragnet_amd/csrc/fragnet_hip.hip"

template <typename F> float time_us(F f, int iters = 30) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / iters;
}

int main() {
  for (int slots : {0, 1024}) {
    fn_set_tuning(FN_TUNE_GEMM_SLOTS, slots);
    printf("-- resident-block budget %d\n", slots);
    for (int K : {128, 17, 167}) {
        for (int64_t M : {2500, 13334, 26492, 47000}) {
            float *X, *Bt, *bias, *Y, *ws, *dW, *db;
            (void)hipMalloc(&X, M * K * 4); (void)hipMalloc(&Bt, K * 128 * 4); (void)hipMalloc(&bias, 512); (void)hipMalloc(&Y, M * 128 * 4);
            (void)hipMalloc(&ws, fn_linear128_wgrad_ws(M, K) * 4); (void)hipMalloc(&dW, 128 * K * 4); (void)hipMalloc(&db, 512);
            (void)hipMemset(X, 0, M * K * 4); (void)hipMemset(Bt, 0, K * 128 * 4); (void)hipMemset(bias, 0, 512); (void)hipMemset(Y, 0, M * 128 * 4);
            float t4 = time_us([&] { fn_linear128_f32(X, K, Bt, bias, Y, M, nullptr, nullptr); });
            float t5 = time_us([&] { fn_linear128_wgrad_f32(Y, X, K, M, ws, dW, db, nullptr); });
            printf("K=%d M=%ld: fwd %.1f us   wgrad(+2 reduces) %.1f us   [%s]\n", K, (long)M, t4, t5, fn_last_error());
            (void)hipFree(X); (void)hipFree(Bt); (void)hipFree(bias); (void)hipFree(Y); (void)hipFree(ws); (void)hipFree(dW); (void)hipFree(db);
        }
    }
  }
    return 0;
}
