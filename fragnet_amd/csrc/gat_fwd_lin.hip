// gat_fwd_lin.hip -- an attention forward pass and the projection GEMM tiles that do not depend on it, in ONE launch
// (k_gat_fwd_lin, k_gat_fwd_pair_lin): the atom projection of layer l beside the bond + fragment-bond levels, the next layer's
// bond / fragment-bond projections beside the atom level.  A translation unit of its own since round 5 (compile time).
#include "fn_internal.h"

namespace {
using fni::fail;
using fni::launch_status;
using fni::tune;
#include "gat_fwd.inc"
#include "linear128.inc"

// gat_base: block id of the first attention workgroup (= T.total: the GEMM blocks come first).
// __launch_bounds__(.., 4): four waves per SIMD as for the plain attention kernels -- without it the accumulators of the GEMM
// branch go to AGPRs ON TOP of the attention branch's VGPRs and the launch drops to three (the two-level destination pass
// with the 8-attribute edge class is at three either way and would spill, so it keeps the default).
template <int H, int KL, int O2 = 0>
__global__ __launch_bounds__(kBlock, 4) void k_gat_fwd_lin(GatFwdArgs A, LinTasks T, int gat_base) {
    extern __shared__ __attribute__((aligned(16))) float sBt[];
    __shared__ float sWf[8][kWfLd];
    int g;
    if (lin_side_role(T, gat_base, &g)) { if (g < T.total) lin_side_block(sBt, T, g);  return; }
    if (g < A.nblk) gat_fwd_body<H, KL, false, O2>(A, sWf, g, A.nblk);
}
template <int H, int KLA, int KLB, bool RDA, int O2 = 0>
__global__ __launch_bounds__(kBlock, 4) void k_gat_fwd_pair_lin(GatFwdArgs A, GatFwdArgs B, LinTasks T, int gat_base) {
    extern __shared__ __attribute__((aligned(16))) float sBt[];
    __shared__ float sWf[8][kWfLd];
    int g;
    if (lin_side_role(T, gat_base, &g)) { if (g < T.total) lin_side_block(sBt, T, g);  return; }
    if (g < A.nblk) gat_fwd_body<H, KLA, RDA, O2>(A, sWf, g, A.nblk);
    else if (g < A.nblk + B.nblk) gat_fwd_body<H, KLB, false, O2>(B, sWf, g - A.nblk, B.nblk);
}

}  // namespace

namespace fni {
// ---- co-launches: an attention pass + independent K = 128 projection tasks (k_gat_*_lin above).  Each returns through the
// plain launches (attention, then launch_linear128_group) whenever the combination has no kernel: the caller never needs to know.
static_assert(kBlock == kLinThreads && kBwdRows * 32 == kLinThreads, "co-launched attention and GEMM workgroups share a block size");
// lays the tasks' workgroups out (one 64 x 64 output tile each); false: cannot ride along (empty, co-launch off, misaligned, or the
// register-resident / wave-independent GEMM variants are selected, which have their own launch shapes)
static bool lin_side_prepare(LinTasks& T, int gat_blocks, int* gat_base, int* grid) {
    if (!tune(FN_TUNE_GEMM_COLAUNCH) || tune(FN_TUNE_GEMM_SLOTS) > 0) return false;
    for (int i = 0; i < T.n; ++i) {
        const LinTask& t = T.t[i];
        if (t.M <= 0) continue;
        if (t.K && t.K != FN_D) return false;
        if (((uintptr_t)t.X | (uintptr_t)t.Bt | (uintptr_t)t.Y | (uintptr_t)t.bias | (uintptr_t)t.mk.y) & 15) return false;
    }
    // one 64 x 64 tile per GEMM workgroup, all of them in front of the attention workgroups: the launch then takes what both
    // parts take back to back minus one kernel boundary (interleaving the two kinds, GEMM workgroups last, persistent GEMM
    // workgroups walking several tiles and raised wave priority all measured slower or equal: DESIGN.md section 4)
    int blocks = 0, live = 0;
    for (int i = 0; i < T.n; ++i) {
        if (T.t[i].M <= 0) continue;
        LinTask t = T.t[i];
        t.first = blocks;
        t.nblk = lin_blocks((t.M + kLinRows - 1) / kLinRows, 1);
        blocks += t.nblk;
        T.t[live++] = t;
    }
    T.n = live;
    T.K = FN_D;
    if (!live) return false;
    T.total = blocks;
    T.base = 0;
    *gat_base = blocks;
    *grid = gat_blocks + blocks;
    return true;
}

int launch_gat_fwd_lin(const GatFwdArgs& A, LinTasks& T, int heads, hipStream_t st) {
    int gb = 0, grid = 0;
    if (A.nblk == 0 || A.rd_out || edge_class(&A.et) != 0 || !lin_side_prepare(T, A.nblk, &gb, &grid)) {
        if (int rc = launch_gat_fwd(A, heads, st)) return rc;
        return T.n ? launch_linear128_group(T, st) : 0;
    }
    const bool tr = fwd_kind_tr(A, heads), ev = fwd_kind_ev(A, heads);
    FN_DISPATCH_H(heads, {
        if constexpr (HH == 4) { if (A.out2 && tr) { hipLaunchKernelGGL((k_gat_fwd_lin<HH, 0, 2>), dim3(grid), dim3(kBlock), kLinSideLds, st, A, T, gb);  break; } }
        if constexpr (HH == 4) { if (!A.out2 && ev) { hipLaunchKernelGGL((k_gat_fwd_lin<HH, 0, 3>), dim3(grid), dim3(kBlock), kLinSideLds, st, A, T, gb);  break; } }
        if (A.out2) hipLaunchKernelGGL((k_gat_fwd_lin<HH, 0, 1>), dim3(grid), dim3(kBlock), kLinSideLds, st, A, T, gb);
        else hipLaunchKernelGGL((k_gat_fwd_lin<HH, 0>), dim3(grid), dim3(kBlock), kLinSideLds, st, A, T, gb);
    });
    return launch_status("attention forward + projections of the next level");
}
int launch_gat_fwd_pair_lin(const GatFwdArgs& A, const GatFwdArgs& B, LinTasks& T, int heads, hipStream_t st) {
    const int ka = edge_class(&A.et), kb = edge_class(&B.et);
    int gb = 0, nwg = 0;
    const bool o2 = A.out2 != nullptr;
    if (A.nblk == 0 || B.nblk == 0 || ka != 1 || (kb != 1 && kb != FN_MAX_EDGE_K) || o2 != (B.out2 != nullptr) ||
        !lin_side_prepare(T, A.nblk + B.nblk, &gb, &nwg)) {
        if (int rc = launch_gat_fwd_pair(A, B, heads, st)) return rc;
        return T.n ? launch_linear128_group(T, st) : 0;
    }
    const dim3 grid(nwg);
#define FN_PAIR_LIN(KB, RD)                                                                                                  \
    do {                                                                                                                     \
        if constexpr (HH == 4) { if (o2 && tr) { hipLaunchKernelGGL((k_gat_fwd_pair_lin<HH, 1, KB, RD, 2>), grid, dim3(kBlock), kLinSideLds, st, A, B, T, gb);  break; } } \
        if constexpr (HH == 4) { if (!o2 && ev) { hipLaunchKernelGGL((k_gat_fwd_pair_lin<HH, 1, KB, RD, 3>), grid, dim3(kBlock), kLinSideLds, st, A, B, T, gb);  break; } } \
        if (o2) hipLaunchKernelGGL((k_gat_fwd_pair_lin<HH, 1, KB, RD, 1>), grid, dim3(kBlock), kLinSideLds, st, A, B, T, gb);   \
        else hipLaunchKernelGGL((k_gat_fwd_pair_lin<HH, 1, KB, RD>), grid, dim3(kBlock), kLinSideLds, st, A, B, T, gb);      \
    } while (0)
    const bool tr = fwd_kind_tr(A, heads) && fwd_kind_tr(B, heads), ev = fwd_kind_ev(A, heads) && fwd_kind_ev(B, heads);
    FN_DISPATCH_H(heads, {
        if (A.rd_out) { if (kb == 1) FN_PAIR_LIN(1, true); else FN_PAIR_LIN(FN_MAX_EDGE_K, true); }
        else { if (kb == 1) FN_PAIR_LIN(1, false); else FN_PAIR_LIN(FN_MAX_EDGE_K, false); }
    });
#undef FN_PAIR_LIN
    return launch_status("attention forward (two levels) + atom projection");
}
}  // namespace fni
