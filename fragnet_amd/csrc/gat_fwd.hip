// gat_fwd.hip -- the attention forward of one level / of two independent levels (k_gat_fwd, k_gat_fwd_pair, k_gat_fwd_rd) and their
// host side (argument validation, launch geometry, fn_gat_fwd_f32).  A translation unit of its own since round 5: the family's
// instantiations (heads x edge classes x second output x fused row dots) were 60 % of the library's compile time.
#include "fn_internal.h"

namespace {
using fni::fail;
using fni::launch_status;
using fni::tune;
using fni::bad_edge_term;
#include "gat_fwd.inc"

// (O2 instances -- the training forward of the one-pass backward, gat_bwd_one.inc -- ask for four waves per SIMD explicitly: their
// second accumulator would otherwise tip the allocation over 128 registers)
template <int H, int KL, int O2 = 0>
__global__ __launch_bounds__(kBlock, 4) void k_gat_fwd(GatFwdArgs A) {
    __shared__ float sWf[8][kWfLd];
    gat_fwd_body<H, KL, false, O2>(A, sWf, (int)blockIdx.x, (int)gridDim.x);
}
// two independent levels in one launch (bond graph + fragment-bond graph: neither reads the other's output)
template <int H, int KLA, int KLB, bool RDA = false, int O2 = 0>
__global__ __launch_bounds__(kBlock, 4) void k_gat_fwd_pair(GatFwdArgs A, GatFwdArgs B) {
    __shared__ float sWf[8][kWfLd];
    if ((int)blockIdx.x < A.nblk) gat_fwd_body<H, KLA, RDA, O2>(A, sWf, (int)blockIdx.x, A.nblk);
    else gat_fwd_body<H, KLB, false, O2>(B, sWf, (int)blockIdx.x - A.nblk, B.nblk);
}
template <int H, int O2 = 0>
__global__ __launch_bounds__(kBlock, 4) void k_gat_fwd_rd(GatFwdArgs A) {          // single bond-graph level with the row-dots epilogue
    __shared__ float sWf[8][kWfLd];
    gat_fwd_body<H, 1, true, O2>(A, sWf, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace

namespace fni {
int prep_gat_fwd(const float* h, const float* s_dst, const float* s_src, const float* att, int att_w,
                        const fn_edge_term* et, const fn_gat_plan* plan, float neg_slope, float* out, float* p_sorted,
                        float* probs_orig, const fn_act_epilogue* act, int heads, GatFwdArgs* A, float* out2,
                        float* sigma) {
    if (!h || !s_dst || !s_src || !att || !plan || bad_edge_term(et, plan->m)) return fail(FN_EINVAL, "fn_gat_fwd_f32: bad argument");
    if (!out && !(act && act->y)) return fail(FN_EINVAL, "fn_gat_fwd_f32: no output buffer");
    if (act && (act->p < 0.f || act->p > 1.f)) return fail(FN_EINVAL, "fn_gat_fwd_f32: dropout probability");
    if (plan->m > 0 && !p_sorted) return fail(FN_EINVAL, "fn_gat_fwd_f32: null p_sorted");
    if (et->mode == 0 && plan->m > 0 && !et->s_sorted) return fail(FN_EINVAL, "fn_gat_fwd_f32: null s_sorted");
    if (heads != 1 && heads != 2 && heads != 4 && heads != 8) return fail(FN_EUNSUPPORTED, "heads must be 1, 2, 4 or 8 (128 = heads * head_dim)");
    *A = GatFwdArgs{h, s_dst, s_src, att, att_w, *et, *plan, neg_slope, out, p_sorted, probs_orig,
                    act ? *act : fn_act_epilogue{nullptr, 0.f, 0, 0, 0, nullptr}, 1, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, 0, nullptr, tune(FN_TUNE_ONE_TIER6) != 0 ? 1 : 0};
    if ((out2 == nullptr) != (sigma == nullptr)) return fail(FN_EINVAL, "fn_gat_fwd_f32: out2 and sigma come together");
    A->out2 = out2;  A->sigma = sigma;
    if (plan->n == 0) return 0;
    if (!(neg_slope >= 0.f && neg_slope <= 1.f)) return fail(FN_EUNSUPPORTED, "fn_gat_fwd_f32: LeakyReLU slope must be in [0, 1]");
    if (plan->n > (1 << 23) || plan->m * heads > (1 << 29))
        return fail(FN_EUNSUPPORTED, "fn_gat_fwd_f32: level too large for 32-bit byte offsets (n <= 2^23 rows, m*heads <= 2^29)");
    // persistent half-waves: as many as fit on the chip at once, each pipelining R rows
    const int64_t groups = (plan->n + kRows - 1) / kRows;
    // with the dropout epilogue (training) fewer, longer-lived half-waves win (5 rows each at B = 512: 27.7 -> 22.4 us for the
    // bond + fragment-bond launch); the plain forward (inference) wants the chip full of them
    const bool training = act && act->y && act->p > 0.f;
    int64_t resident = (int64_t)tune(training ? FN_TUNE_FWD_BLOCKS : FN_TUNE_FWD_BLOCKS_EVAL);
    // ... but not arbitrarily long-lived: a large level (2048+ molecules per batch) made every half-wave of the plain forward walk
    // 12-46 rows and the launch wait for its slowest workgroups -- it gets FN_TUNE_FWD_BLOCKS_EVAL_LARGE workgroups instead (round 5)
    const int64_t large = (int64_t)tune(FN_TUNE_FWD_BLOCKS_EVAL_LARGE);
    if (!training && large > resident && groups > 4 * resident) resident = large;
    A->rows_per_hw = (int)((groups + resident - 1) / resident);
    A->nblk = (int)((plan->n + (int64_t)kRows * A->rows_per_hw - 1) / ((int64_t)kRows * A->rows_per_hw));
    return 0;
}

int launch_gat_fwd(const GatFwdArgs& A, int heads, hipStream_t st) {
    if (A.nblk == 0) return 0;
    const int kl = edge_class(&A.et);
    const bool o2 = A.out2 != nullptr;
    if (A.rd_out) {
        if (kl != 1) return fail(FN_EUNSUPPORTED, "attention forward: the row-dots epilogue exists for the single-attribute (bond graph) level");
        FN_DISPATCH_H(heads, {
            if (o2) hipLaunchKernelGGL((k_gat_fwd_rd<HH, 1>), dim3(A.nblk), dim3(kBlock), 0, st, A);
            else hipLaunchKernelGGL((k_gat_fwd_rd<HH>), dim3(A.nblk), dim3(kBlock), 0, st, A);
        });
        return launch_status("fn_gat_fwd_f32 (+ row dots)");
    }
#define FN_FWD1(KLV)                                                                                          \
    do {                                                                                                      \
        if constexpr (HH == 4) { if (o2 && tr) { hipLaunchKernelGGL((k_gat_fwd<HH, KLV, 2>), dim3(A.nblk), dim3(kBlock), 0, st, A);  break; } } \
        if constexpr (HH == 4) { if (!o2 && ev) { hipLaunchKernelGGL((k_gat_fwd<HH, KLV, 3>), dim3(A.nblk), dim3(kBlock), 0, st, A);  break; } } \
        if (o2) hipLaunchKernelGGL((k_gat_fwd<HH, KLV, 1>), dim3(A.nblk), dim3(kBlock), 0, st, A);            \
        else hipLaunchKernelGGL((k_gat_fwd<HH, KLV>), dim3(A.nblk), dim3(kBlock), 0, st, A);                  \
    } while (0)
    const bool tr = fwd_kind_tr(A, heads), ev = fwd_kind_ev(A, heads);      // (four heads: the engine's launches take the kinds whose uniform flags are compile-time)
    FN_DISPATCH_H(heads, {
        if (kl == 0) FN_FWD1(0);
        else if (kl == 1) FN_FWD1(1);
        else FN_FWD1(FN_MAX_EDGE_K);
    });
#undef FN_FWD1
    return launch_status("fn_gat_fwd_f32");
}
// two levels, one launch, when their edge classes are (1, FN_MAX_EDGE_K) or (1, 1); two launches otherwise
int launch_gat_fwd_pair(const GatFwdArgs& A, const GatFwdArgs& B, int heads, hipStream_t st) {
    const int ka = edge_class(&A.et), kb = edge_class(&B.et);
    if (A.nblk == 0 || B.nblk == 0 || ka != 1 || (kb != 1 && kb != FN_MAX_EDGE_K)) {
        if (int rc = launch_gat_fwd(A, heads, st)) return rc;
        return launch_gat_fwd(B, heads, st);
    }
    const bool o2 = A.out2 != nullptr;
    if (o2 != (B.out2 != nullptr)) {
        if (int rc = launch_gat_fwd(A, heads, st)) return rc;
        return launch_gat_fwd(B, heads, st);
    }
#define FN_FWD2(KB, RD)                                                                                                            \
    do {                                                                                                                           \
        if constexpr (HH == 4) { if (o2 && tr) { hipLaunchKernelGGL((k_gat_fwd_pair<HH, 1, KB, RD, 2>), dim3(A.nblk + B.nblk), dim3(kBlock), 0, st, A, B);  break; } } \
        if constexpr (HH == 4) { if (!o2 && ev) { hipLaunchKernelGGL((k_gat_fwd_pair<HH, 1, KB, RD, 3>), dim3(A.nblk + B.nblk), dim3(kBlock), 0, st, A, B);  break; } } \
        if (o2) hipLaunchKernelGGL((k_gat_fwd_pair<HH, 1, KB, RD, 1>), dim3(A.nblk + B.nblk), dim3(kBlock), 0, st, A, B);          \
        else hipLaunchKernelGGL((k_gat_fwd_pair<HH, 1, KB, RD>), dim3(A.nblk + B.nblk), dim3(kBlock), 0, st, A, B);                \
    } while (0)
    const bool tr = fwd_kind_tr(A, heads) && fwd_kind_tr(B, heads), ev = fwd_kind_ev(A, heads) && fwd_kind_ev(B, heads);
    FN_DISPATCH_H(heads, {
        if (A.rd_out) { if (kb == 1) FN_FWD2(1, true); else FN_FWD2(FN_MAX_EDGE_K, true); }
        else { if (kb == 1) FN_FWD2(1, false); else FN_FWD2(FN_MAX_EDGE_K, false); }
    });
#undef FN_FWD2
    return launch_status("attention forward (two levels)");
}

}  // namespace fni

using fni::prep_gat_fwd;
using fni::launch_gat_fwd;
using fni::GatFwdArgs;
extern "C" {
int fn_gat_fwd_f32(const float* h, const float* s_dst, const float* s_src, const float* att, int att_w,
                   const fn_edge_term* et, const fn_gat_plan* plan, float neg_slope, float* out, float* p_sorted,
                   float* probs_orig, float* out2, float* sigma, int p_edge_major, const fn_act_epilogue* act, int heads,
                   fn_stream_t stream) {
    GatFwdArgs A;
    if (int rc = prep_gat_fwd(h, s_dst, s_src, att, att_w, et, plan, neg_slope, out, p_sorted, probs_orig, act, heads, &A, out2, sigma)) return rc;
    A.p_edge_major = p_edge_major ? 1 : 0;
    return launch_gat_fwd(A, heads, S(stream));
}
}  // extern "C"
