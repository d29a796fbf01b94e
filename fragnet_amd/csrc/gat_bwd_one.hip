// gat_bwd_one.hip -- the backward of an attention level as ONE source-owner pass (k_gat_bwd_one, k_gat_bwd_one3), the node-local dots
// it needs where no GEMM epilogue forms them (k_gat_cu) and the segment sums of its deferred form (k_gsd_seg); their launchers and C
// entry points.  A translation unit of its own since round 5: the twenty instantiations of the pass were a quarter of the main
// unit's device code and of its compile time.
#include "fn_internal.h"

namespace {
using fni::fail;
using fni::launch_status;
using fni::tune;
using fni::stamps;
using fni::bad_edge_term;
#include "gat_fwd.inc"
#include "gat_bwd_one.inc"
}  // namespace

namespace fni {
static bool one_pass_heads(int heads) { return heads == 2 || heads == 4 || heads == 8 || heads == 1; }
int prep_gat_bwd_one(const float* g_out, const float* h, const float* p_sorted, const float* cdot, const float* g_s_dst,
                            const fn_edge_term* et, const float* att, int att_w, int dst_off, int src_off, const fn_gat_plan* plan,
                            float neg_slope, float* g_h, float* dz_sorted, float* g_s_orig, float* part_a, int* n_part_a, float* part_e,
                            int* n_part_e, int heads, GatBwdOneArgs* A, int64_t share) {
    if (!g_out || !h || !cdot || !g_s_dst || !att || !plan || !g_h || !part_a || !n_part_a || !n_part_e || !et)
        return fail(FN_EINVAL, "fn_gat_bwd_one_f32: bad argument");
    if (et->mode != 0 && bad_edge_term(et, plan ? plan->m : 1)) return fail(FN_EINVAL, "fn_gat_bwd_one_f32: bad edge term");
    if (plan->m > 0 && (!p_sorted || !plan->dpos_s || !plan->dst_s)) return fail(FN_EINVAL, "fn_gat_bwd_one_f32: null edge buffer");
    if (et->mode == 2 && !part_e) return fail(FN_EINVAL, "fn_gat_bwd_one_f32: null part_e");
    if ((att_w | dst_off | src_off) & 3) return fail(FN_EINVAL, "fn_gat_bwd_one_f32: att blocks must be 16-byte aligned");
    if (!one_pass_heads(heads)) return fail(FN_EUNSUPPORTED, "heads must be 1, 2, 4 or 8 (128 = heads * head_dim)");
    *n_part_a = 0;  *n_part_e = 0;
    *A = GatBwdOneArgs{g_out, h, p_sorted, cdot, g_s_dst, att, att_w, dst_off, src_off, *et, *plan, neg_slope, g_h, part_a, part_e,
                       dz_sorted, g_s_orig, 1, 0, 0, et->x_src, nullptr, tune(FN_TUNE_ONE_TIER6) != 0 ? 1 : 0, nullptr, nullptr};
    if (plan->n == 0) return 0;
    if (plan->m == 0) {
        // a level with nodes but no edges (a batch of single-fragment molecules: the fragment-bond graph): the row loop's per-edge
        // loads are unconditional with clamped indices (position 0), so every per-edge array needs SOME readable word behind it
        // -- the [n, H] dots table stands in (the values are never used: no lane has an edge) -- and nothing per-edge is written
        A->p_sorted = cdot;
        A->et.x_sorted = cdot;
        A->x_src = nullptr;
        A->dz_sorted = nullptr;
        A->g_s_orig = nullptr;
    }
    if (plan->n > (1 << 23) || plan->m * heads > (1 << 28))
        return fail(FN_EUNSUPPORTED, "fn_gat_bwd_one_f32: level too large for 32-bit byte offsets (n <= 2^23 rows, m*heads <= 2^28)");
    // persistent half-waves pipelining R rows each; every block writes one row of partial sums (<= 1024 blocks).  The kernel runs
    // three workgroups per CU (its twelve gradient rows in flight cost the fourth), so 768 are resident at once: a launch of more
    // pays a second, mostly empty round.  share > 0: this level's part of a launch that carries several (by rows)
    const int64_t groups = (plan->n + kBwdRows - 1) / kBwdRows;
    int64_t resident = share > 0 ? share : (int64_t)tune(FN_TUNE_ONE_BLOCKS);
    if (resident > FN_MAX_PART || resident < 1) resident = 1024;
    A->rows_per_hw = (int)((groups + resident - 1) / resident);
    A->nblk = (int)((plan->n + (int64_t)kBwdRows * A->rows_per_hw - 1) / ((int64_t)kBwdRows * A->rows_per_hw));
    *n_part_a = A->nblk;
    *n_part_e = et->mode == 2 ? A->nblk : 0;
    {
        int64_t n_u64 = 0;
        unsigned long long* sb = stamps(&n_u64);
        if (sb && n_u64 >= (int64_t)A->nblk * (kBwdRows / 2) * 16) A->stamps = sb;
    }     // dev aid, see GatBwdOneArgs
    return 0;
}
// the engine's configuration of a level (gat_bwd_one.inc, EN): its uniform flags become compile-time constants.  An absent level
// (nblk == 0) of a multi-level launch does not matter
static bool one_kind_en(const GatBwdOneArgs& A) {
    if (A.nblk == 0) return true;
    const int kl = edge_class(&A.et);
    return tune(FN_TUNE_ENGINE_CONST) != 0 && A.p_edge_major != 0 && A.stamps == nullptr && A.pl.m >= 2 && A.dz_em == nullptr && A.dz_sorted == nullptr &&
           (kl != 1 || A.x_src != nullptr) && (kl != 0 || A.g_s_orig != nullptr);
}
int launch_gat_bwd_one(const GatBwdOneArgs& A, int heads, hipStream_t st) {
    if (A.nblk == 0) return 0;
    const int kl = edge_class(&A.et);
    if (heads == 4 && !A.dz_em && one_kind_en(A)) {      // the engine's launches (four heads)
        if (kl == 0) hipLaunchKernelGGL((k_gat_bwd_one<4, 0, kBwdRows, false, true>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
        else if (kl == 1) hipLaunchKernelGGL((k_gat_bwd_one<4, 1, kBwdRows, false, true>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
        else hipLaunchKernelGGL((k_gat_bwd_one<4, FN_MAX_EDGE_K, kBwdRows, false, true>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
        return launch_status("fn_gat_bwd_one_f32");
    }
    if (A.dz_em) {            // the deferred form (DF): four heads
        if (heads != 4) return fail(FN_EUNSUPPORTED, "one-pass backward, deferred form: four heads");
        if (kl == 0) hipLaunchKernelGGL((k_gat_bwd_one<4, 0, kBwdRows, true>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
        else if (kl == 1) hipLaunchKernelGGL((k_gat_bwd_one<4, 1, kBwdRows, true>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
        else hipLaunchKernelGGL((k_gat_bwd_one<4, FN_MAX_EDGE_K, kBwdRows, true>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
        return launch_status("fn_gat_bwd_one_f32 (deferred form)");
    }
    FN_DISPATCH_H(heads, {
        if (kl == 0) hipLaunchKernelGGL((k_gat_bwd_one<HH, 0, kBwdRows>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
        else if (kl == 1) hipLaunchKernelGGL((k_gat_bwd_one<HH, 1, kBwdRows>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
        else hipLaunchKernelGGL((k_gat_bwd_one<HH, FN_MAX_EDGE_K, kBwdRows>), dim3(A.nblk), dim3(kBwdRows * 32), 0, st, A);
    });
    return launch_status("fn_gat_bwd_one_f32");
}
// bond (edge class 1) + atom (class 0) + fragment-bond (class FN_MAX_EDGE_K) levels as one launch; any of them may be absent
// (nblk == 0).  Levels whose edge class does not fit their slot take launches of their own.
int launch_gat_bwd_one3(const GatBwdOneArgs& A, const GatBwdOneArgs& B, const GatBwdOneArgs& C, int heads, hipStream_t st) {
    const bool okA = A.nblk == 0 || edge_class(&A.et) == 1, okB = B.nblk == 0 || edge_class(&B.et) == 0,
               okC = C.nblk == 0 || edge_class(&C.et) == FN_MAX_EDGE_K;
    const int live = (A.nblk > 0) + (B.nblk > 0) + (C.nblk > 0);
    if (!okA || !okB || !okC || live < 2) {
        if (int rc = launch_gat_bwd_one(A, heads, st)) return rc;
        if (int rc = launch_gat_bwd_one(B, heads, st)) return rc;
        return launch_gat_bwd_one(C, heads, st);
    }
    const int interleave = tune(FN_TUNE_ONE_INTERLEAVE) != 0 ? 1 : 0;
    const bool df = (A.nblk && A.dz_em) || (B.nblk && B.dz_em) || (C.nblk && C.dz_em);
    if (df) {
        // every level of the launch deferred, or -- the mixed form's boundary launch -- the bond / fragment-bond levels deferred and the
        // atom level (the layer below's) not
        const bool ac = !(A.nblk && !A.dz_em) && !(C.nblk && !C.dz_em);
        if (heads != 4 || !ac) return fail(FN_EUNSUPPORTED, "one-pass backward, deferred form: four heads; the bond and fragment-bond levels of a launch share a form");
        if (B.nblk && !B.dz_em)
            hipLaunchKernelGGL((k_gat_bwd_one3<4, kBwdRows, true, false>), dim3(A.nblk + B.nblk + C.nblk), dim3(kBwdRows * 32), 0, st, A, B, C, interleave);
        else
            hipLaunchKernelGGL((k_gat_bwd_one3<4, kBwdRows, true>), dim3(A.nblk + B.nblk + C.nblk), dim3(kBwdRows * 32), 0, st, A, B, C, interleave);
        return launch_status("attention backward, one pass, deferred form (bond + atom + fragment-bond levels)");
    }
    if (heads == 4 && one_kind_en(A) && one_kind_en(B) && one_kind_en(C)) {      // the engine's launches (four heads)
        hipLaunchKernelGGL((k_gat_bwd_one3<4, kBwdRows, false, false, true>), dim3(A.nblk + B.nblk + C.nblk), dim3(kBwdRows * 32), 0, st, A, B, C, interleave);
        return launch_status("attention backward, one pass (bond + atom + fragment-bond levels)");
    }
    FN_DISPATCH_H(heads, hipLaunchKernelGGL((k_gat_bwd_one3<HH, kBwdRows>), dim3(A.nblk + B.nblk + C.nblk), dim3(kBwdRows * 32), 0, st, A, B, C, interleave));
    return launch_status("attention backward, one pass (bond + atom + fragment-bond levels)");
}
int launch_gat_cu(CuTasks& T, int heads, hipStream_t st) {
    int blocks = 0, live = 0;
    for (int i = 0; i < T.n; ++i) {
        if (T.t[i].n <= 0) continue;
        CuTask t = T.t[i];
        if (!t.g || !t.out || !t.c || (t.out2 && (!t.sigma || !t.u))) return fail(FN_EINVAL, "fn_gat_cu_f32: null argument");
        if (((uintptr_t)t.g | (uintptr_t)t.out | (uintptr_t)t.out2) & 15) return fail(FN_EINVAL, "fn_gat_cu_f32: rows must be 16-byte aligned");
        t.first = blocks;
        t.nblk = row_grid(t.n, kGridCap);
        blocks += t.nblk;
        T.t[live++] = t;
    }
    T.n = live;
    if (!live) return 0;
    FN_DISPATCH_H(heads, hipLaunchKernelGGL(k_gat_cu<HH>, dim3(blocks), dim3(kBlock), 0, st, T));
    return launch_status("fn_gat_cu_f32");
}

int launch_gsd_seg(const GsdSegTasks& T, int blocks, hipStream_t st) {
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(k_gsd_seg, dim3(blocks), dim3(kBlock), 0, st, T);
    return launch_status("k_gsd_seg");
}
}  // namespace fni

using fni::fail;
using fni::launch_status;
using fni::launch_gat_bwd_one;
using fni::launch_gat_cu;
using fni::prep_gat_bwd_one;

extern "C" {

int fn_gat_cu_f32(const float* g_out, const float* out, const float* out2, const float* sigma, float scale, float* c, float* u,
                  int64_t n, int heads, fn_stream_t stream) {
    if (n < 0) return fail(FN_EINVAL, "fn_gat_cu_f32: negative size");
    CuTasks T{};
    T.n = 1;
    T.t[0] = CuTask{g_out, out, out2, sigma, scale, c, u, n, 0, 0};
    return launch_gat_cu(T, heads, S(stream));
}

int fn_gat_bwd_one_f32(const float* g_out, const float* h, const float* p_sorted, const float* cdot, const float* g_s_dst,
                       const fn_edge_term* et, const float* att, int att_w, int dst_off, int src_off, const fn_gat_plan* plan,
                       float neg_slope, float* g_h, float* dz_sorted, float* g_s_orig, float* part_a, int* n_part_a, float* part_e,
                       int* n_part_e, int p_edge_major, float* dz_em, int heads, fn_stream_t stream) {
    GatBwdOneArgs A;
    if (dz_em && heads != 4) return fail(FN_EUNSUPPORTED, "fn_gat_bwd_one_f32: the deferred form (dz_em) is written for four heads");
    // (the deferred form never reads g_s_dst: the dots table stands in where the argument check wants a pointer)
    if (int rc = prep_gat_bwd_one(g_out, h, p_sorted, cdot, dz_em && !g_s_dst ? cdot : g_s_dst, et, att, att_w, dst_off, src_off, plan, neg_slope, g_h,
                                  dz_sorted, g_s_orig, part_a, n_part_a, part_e, n_part_e, heads, &A)) return rc;
    A.p_edge_major = p_edge_major ? 1 : 0;
    A.dz_em = dz_em;
    return launch_gat_bwd_one(A, heads, S(stream));
}

int fn_gat_gsd_f32(const float* dz_em, const fn_gat_plan* plan, const float* h, float* g_s_dst, float* part_a, int n_part_a,
                   fn_stream_t stream) {
    if (!plan || !g_s_dst || !part_a || !h || n_part_a < 0 || n_part_a > FN_MAX_PART) return fail(FN_EINVAL, "fn_gat_gsd_f32: bad argument");
    if (plan->n == 0 || n_part_a == 0) return 0;
    // (k_gsd_seg reads rowptr[row], rowptr[row + 1] of every row, edges or not)
    if (!plan->rowptr_d || (plan->m > 0 && !dz_em)) return fail(FN_EINVAL, "fn_gat_gsd_f32: null edge buffer");
    GsdSegTasks T{};
    T.n = 1;
    // (a level without edges: every extent is empty, nothing of dz is read -- any readable word will do)
    T.t[0] = GsdSegTask{plan->m > 0 ? dz_em : h, plan->rowptr_d, plan->pos_base_d, plan->n, g_s_dst, nullptr, h, part_a, 0, n_part_a};
    if (int rc = fni::launch_gsd_seg(T, n_part_a, S(stream))) return rc;
    return launch_status("fn_gat_gsd_f32");
}

}  // extern "C"
