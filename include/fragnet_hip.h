/*
 * libfragnet_hip.so -- C-ABI of the MI355X (gfx950) kernels behind FragNet's message-passing hot path.
 *
 * This is the drop-in boundary of SURVEY.md §8(b): everything the reference obtains from the
 * un-vendored torch-scatter / torch_geometric wheels on this path, plus the fused per-level
 * kernels that replace the reference's index_select/cat/mul/sum/scatter chains.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no torch types.
 *   - every pointer is a DEVICE pointer into caller-owned memory; the library never allocates,
 *     frees or retains pointers; workspaces are caller-provided.
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no
 *     internal synchronisation, no default-stream use; stateless and re-entrant -- with two documented
 *     exceptions that are process-wide settings, meant to be set once before the first compute call and not
 *     changed while another thread is inside the library: the tuning table (fn_set_tuning) and the profiling
 *     stamp buffer (fn_debug_set_stamps).  Kernels never read either; only launch decisions on the host do.
 *   - return 0 on success, a negative FN_E* code for an argument error, or a positive
 *     hipError_t taken right after the launch.  fn_last_error() describes the last failure
 *     on the calling thread.
 *   - float data is fp32, row-major, feature width D = 128 (= heads * head_dim) where stated;
 *     index data is int32 in plans (int64 only where the batch dict hands it over).
 *
 * Reference call sites each entry point replaces are cited as file:line under
 * /root/reference/fragnet/.
 */
#ifndef FRAGNET_HIP_H
#define FRAGNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FN_ABI_VERSION 12
#define FN_D 128              /* feature width of every node table on the path (emb_dim) */
#define FN_MAX_TASKS 16       /* CSR builds fused into one fn_plan_build call */
#define FN_MAX_EDGE_K 8       /* widest raw edge attribute folded in-kernel (6 for fragment bonds) */

#define FN_EINVAL (-1)        /* null pointer / negative size / misaligned */
#define FN_EUNSUPPORTED (-2)  /* heads or width not in {1,2,4,8} x 128 */
#define FN_ETOOMANY (-3)      /* more than FN_MAX_TASKS tasks */

typedef void* fn_stream_t;    /* hipStream_t */

int fn_abi_version(void);

/* Process-wide tuning knobs (defaults are the measured best for MI355X; the bench uses them for A/B runs).
 * FN_TUNE_FWD_BLOCKS: workgroups of the attention kernels that are resident at once in a TRAINING pass (forward with the
 *   dropout epilogue, source pass of the backward): default 1024 = 256 CUs x 4 (768 until round 5's re-sweep of the launch
 *   shapes, profiles/r05_launch_shape_sweep.txt); a level with more row groups than that gives
 *   every half-wave several consecutive rows to software-pipeline (4 at ESOL batch 512).  FN_TUNE_FWD_BLOCKS_EVAL (1792 =
 *   256 x 7) is the same for the plain forward (inference, or training without dropout). */
#define FN_TUNE_FWD_BLOCKS 0
#define FN_TUNE_GEMM_SLOTS 1   /* > 0: cap on the workgroups of a projection GEMM launch (1024 = 256 CUs x 4 resident); a launch with
                                * more 64x64 output tiles then walks several row tiles per workgroup, prefetching the next tile's
                                * rows.  Default 0 = one tile per workgroup (faster at every measured size) */
#define FN_TUNE_RETIRED_2 2     /* (retired in round 3, no effect: parameter-gradient work and the fragment-bond chain forked onto side streams --
                                * a forked hipGraph replays SLOWER than the serial one on ROCm 7.2, 1.84 vs 1.69 ms per step) */
#define FN_TUNE_WGRAD_BLOCKS 3 /* target workgroup count of the grouped weight-gradient launch of a backward pass (default 192 since round 5's re-sweep -- with layer 0's workgroups the launch then has 1.6 per CU; 256: + 1 % per step; 512 wrote twice the partials) */
#define FN_TUNE_RETIRED_4 4     /* (retired in round 3, no effect: the molecule-resident fused FORWARD of round 2 -- measured equal at 512
                                * molecules, slower elsewhere; source kept out of the build under tools/probe/retired/mol_fused.inc) */
#define FN_TUNE_RETIRED_5 5     /* (retired: phase skew of that kernel) */
#define FN_TUNE_RETIRED_6 6     /* (retired: register-resident-weights projection kernel k_proj128, 1-4 % slower per step) */
#define FN_TUNE_FUSE_ROWDOTS 7 /* 1 (default): inside fn_encoder_forward the bond-graph attention kernel also writes the atom graph's edge
                                * term <new_bond, a[:, d:d+128]> from the row it holds in registers; 0: a separate row-dots launch */
#define FN_TUNE_WGRAD_DIRECT 8 /* 1 (default): the grouped K = 128 weight-gradient partials run as k_wgrad128_multi (operands straight from
                                * global memory in the MFMA layout, csrc/wgrad128.inc); 0: the LDS-staged k_linear128_wgrad_multi (also
                                * what layer 0's narrow products use) */
#define FN_TUNE_RETIRED_9 9     /* (retired: wave-independent projection kernel k_proj_direct, 1 % slower per step) */
#define FN_TUNE_FWD_BLOCKS_EVAL 10
#define FN_TUNE_DST_BLOCKS 11  /* target workgroup count of the backward destination pass (default 1536: three rows per half-wave at ESOL batch
                                * 512; never more than three rows, see prep_gat_bwd_dst) */
#define FN_TUNE_SRC_BLOCKS 12  /* resident workgroups of the backward source pass (default 512; <= 1024: every block writes a row of partial sums) */
#define FN_TUNE_RD_BLOCKS 13   /* (default 512; 256 until round 5) workgroups of the edge-term backward that shares the source pass's launch (each writes a row of partial sums) */
#define FN_TUNE_GEMM_COLAUNCH 14 /* inside fn_encoder_*: the 128 -> 128 projections (forward) and input-gradient products (backward) that do
                                 * not depend on an attention pass ride along as extra workgroups of that pass's launch (the atom projection
                                 * beside the bond + fragment-bond levels, the next layer's bond / fragment-bond projections beside the atom
                                 * level; mirrored in the backward), layer 0's three raw-feature projections share a launch and all weight-gradient
                                 * partial products of a backward pass are one launch.  != 0 (default 2): on, the GEMM workgroups first in the
                                 * launch (after / interleaved with the attention workgroups measured slower and were removed in round 3:
                                 * profiles/r02e_colaunch_ab.txt); 0: separate launches */
#define FN_TUNE_RETIRED_15 15    /* (retired: persistent riding GEMM workgroups walking several tiles, slower) */
#define FN_TUNE_RETIRED_16 16    /* (retired: raised wave priority of the riding workgroups, no effect) */
#define FN_TUNE_RETIRED_17 17    /* (retired in round 5: the two-pass backward's atom chain beside the bond chain -- superseded by the one-pass
                                  * backward; the two passes remain as the general path, in plain dependency order) */
#define FN_TUNE_RETIRED_18 18    /* (retired in round 5: the molecule-resident single-pass backward of round 3 -- 39 us against 28 for a level;
                                  * source kept out of the build under tools/probe/retired/mol_bwd.hip) */
#define FN_TUNE_RETIRED_19 19    /* (retired: test hook of that kernel) */
#define FN_TUNE_MOL_TAIL 20           /* 1 (default): batches marked molecule-contiguous (fn_encoder.mol_contiguous) run the last layer's
                                       * fragment tail -- fragment sums, fragment graph, readout and their backward -- as one
                                       * molecule-resident launch each way (csrc/mol_tail.inc); 0: the separate launches;
                                       * 2 (test hook): fused, every molecule on the global-memory path of oversize molecules */
#define FN_TUNE_WGRAD0_ROWS 21        /* layer 0's weight-gradient products (K = 167 / 17 / 6): rows per block as a multiple of the
                                       * per-product rule (M / 256 rounded up to 32); tens digit = the product with K > 128 (atoms),
                                       * units digit = the narrow ones.  Default 23 */
#define FN_TUNE_BWD_ONE 22            /* 1: fn_encoder_backward runs every attention level's backward (bond / atom / fragment-bond graph) as
                                       * ONE source-owner pass (csrc/gat_bwd_one.inc): the forward then also writes out2 / sigma and the
                                       * producers of the gradient rows write the node-local dots c, g_s_dst; 0: the two-pass backward */
#define FN_TUNE_ONE_BLOCKS 23         /* target workgroups per LEVEL of a one-pass backward launch (default 768 = five rows per half-wave at the bond level of ESOL batch 512; every block writes a row of
                                       * partial sums, so at most FN_MAX_PART).  Round 5's re-sweep: 1024 + 1.0 %, 512-896 within noise, 384 and fewer slower */
#define FN_TUNE_PAD_SKIP 24           /* 1 (default): with the one-pass backward on a molecule-contiguous static-shape batch the kernels skip the
                                       * padding rows (zero rows out, no gathers, no GEMM tiles, no weight-gradient rows); 0: they are processed */
#define FN_TUNE_ONE_INTERLEAVE 25      /* 1: in the one-pass backward's three-level launch the bond level's workgroups alternate with the atom /
                                       * fragment-bond levels'; 0 (default): level after level (alternating measured 42-46 us against 37-40) */
#define FN_TUNE_ONE_TIER6 26           /* 1 (default): the attention kernels' gather tiers include 6 rows (forward 4 / 6 / 8, one-pass backward
                                       * 4 / 6 / 8 / 12 per round trip); 0: 4 / 8 (/ 12) */
#define FN_TUNE_RIDER_PIECES 27       /* 16-byte pieces per thread of the Adam slice that rides in the deferred-reduction launch (fn_encoder.adam_rider):
                                       * 1 (default) = as many 1024-thread workgroups as the slice has KiB x 16 */
#define FN_TUNE_RIDER_AT 28           /* which launch of the one-pass encoder backward carries fn_encoder.adam_rider: 0 (default) the last one (the
                                       * deferred reductions: + 7 us there for the 12.9 - 4.5 us the step's Adam launch saves), 1 the first one (the
                                       * molecule-resident fragment tail, when it runs: + 11.7 us) */
#define FN_TUNE_DEFER_GSD 29           /* 1: with the one-pass backward and four heads (gat2) the forward writes NO second output (out2 / sigma): the
                                       * pass leaves dz at the edges' destination-order slots and the terms that need g_s_dst = the sum of a row's
                                       * segment are added by whoever reads g_h next -- the input-gradient product (one more MFMA step, k = the four
                                       * heads), the weight-gradient kernels (the block's partial; side product U = G^T X for dL/da_dst), one small
                                       * launch for layer 0.  0 (default): out2 / sigma in the forward, <g, out2> in the producers' epilogues.
                                       * Measured at ESOL batch 512 (round 5, same call): the forward launches 18 us shorter, the backward 32 us
                                       * longer (weight-gradient launch + 21, layer 0's launch + 12): 0.797 against 0.783 ms per step.
                                       * 2 (round 6): the MIXED form -- layers >= 1 deferred, layer 0 keeps out2 / sigma (no k_gsd_seg, layer 0's
                                       * weight-gradient kernels plain); the two boundary launches carry both epilogues */
#define FN_TUNE_FWD_BLOCKS_EVAL_LARGE 30 /* the plain forward (inference) of a level with more than 4 x FN_TUNE_FWD_BLOCKS_EVAL row groups (2048+ molecules
                                       * per batch) runs this many workgroups instead (default 6144): with the fixed count every half-wave walked 12-46
                                       * rows and the launch waited for its slowest workgroups.  0: one count for every size (rounds 1-4) */
#define FN_TUNE_FWD_TAIL_ROWS 31       /* > 0: in the two-level forward launch (bond + fragment-bond graph) the SECOND level's half-waves take at most this
                                       * many rows: its workgroups are dispatched last, so long-lived ones are the launch's tail (only large batches
                                       * reach the cap: 2048 molecules per batch forward-only 1.13 -> 1.28 M molecules/s).  Evaluation passes only (no dropout
                                       * epilogue), as FN_TUNE_FWD_BLOCKS_EVAL_LARGE.  Default 1; 0: no cap */
#define FN_TUNE_DENSE_TILES 32         /* 1 (default): fn_dense_fwd_f32 runs inputs of >= 192 tiles of 64 x 128 (1536+ rows at N = 1024) with K a multiple of 32 on
                                       * workgroup-shared LDS macro-tiles (k_dense_fwd_tiles, round 6: 62 -> 43 us at M = 2048, K = N = 1024); 0: the per-wave
                                       * operand kernel for every shape (A/B, and the parity test of the two against each other) */
#define FN_TUNE_ENGINE_CONST 33        /* 1 (default): the attention launches of the engine (four heads, levels of >= 2 edges, no probs_orig / stamp buffer)
                                       * run kernel instances in which their uniform run-time flags are compile-time constants -- a training pass:
                                       * edge-major probabilities, relu(dropout(.)) epilogues with p > 0 (forward kind 2, the one-pass backward's EN
                                       * instances); an evaluation pass: head-major probabilities, ReLU epilogues without dropout (forward kind 3).  Same
                                       * arithmetic, bit-identical results, fewer scalar tests and branches in the issue-bound row loops (round 6: 0.769 ->
                                       * 0.755 ms per training step, the forward-only sweep - 4 ... - 7 %); 0: the general instances for every launch (A/B,
                                       * and the parity test of the two against each other) */
#define FN_TUNE_COUNT 34
int fn_set_tuning(int key, int value);
/* Profiling aid (process-wide, like the tuning knobs): while `buf` (device, n_u64 >= 16 * 4 * workgroups 64-bit words) is set, every wave
 * of the one-pass attention backward (fn_gat_bwd_one_f32) writes s_memtime stamps of its phases into it (tools/probe/bwd_one_probe.py
 * --stamps).  NULL switches it off (the default: the kernel then pays one uniform branch per phase). */
int fn_debug_set_stamps(void* buf, int64_t n_u64);
/* Profiling aid (process-wide; ABI 12): four hipEvent_t (created with timing enabled) or NULL entries -- events[0] / [1] are recorded
 * right before / behind the bond-graph level's forward launch of layer 0 in fn_encoder_forward (k_gat_fwd_pair: bond + fragment-bond
 * levels), events[2] / [3] around the same level's one-pass backward launch in fn_encoder_backward (k_gat_bwd_one3 of layer 0).
 * On a capturing stream they become EXTERNAL event-record nodes (hipEventRecordExternal), so hipEventElapsedTime between a pair
 * after a replay is that launch's duration INSIDE the captured step -- what bench.py's `roofline` quotes.  NULL (the default, also
 * `events` == NULL) records nothing.  The events stay the caller's. */
int fn_debug_set_profile_events(void* const* events);
const char* fn_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Graph plan: destination-sorted CSRs, built once per batch and reused by all layers, fwd+bwd.
 * Replaces the implicit index handling inside torch_scatter.scatter_add / scatter_softmax
 * (model/gat/gat2.py:153,162,210,216,234,257,265,303,309,820,821) and
 * torch_geometric.utils.add_self_loops (gat2.py:179: `n_loops` identity items i -> segment i
 * appended after the real ones).
 *
 * One call builds up to FN_MAX_TASKS CSRs over a concatenated segment space:
 *   rowptr_all[seg_base + s]            global position of segment s's first item
 *   perm_all  [item_base + p]           task-local item id at sorted position p, ascending inside
 *                                       a segment (=> deterministic, reference summation order)
 * For a GAT level, two tasks are paired (role DST keyed by destination, role SRC keyed by source):
 *   DST task: aux_a[pos] = source node of the edge at pos; aux_b[item] = position of edge `item` (inverse perm)
 *             aux_c[pos] = position of the same edge in the paired SRC task
 *   SRC task: aux_a[pos] = destination node; aux_b[pos] = that edge's position (task-local) in the
 *             paired DST task
 * ------------------------------------------------------------------------------------------ */
#define FN_ROLE_PLAIN 0
#define FN_ROLE_DST 1
#define FN_ROLE_SRC 2

typedef struct fn_csr_task {
    const int64_t* key;        /* [n_real] segment id per item (device)                        */
    const int64_t* other_key;  /* [n_real] the opposite endpoint (roles DST/SRC), else NULL     */
    int64_t n_real;            /* items with an explicit key                                    */
    int64_t n_loops;           /* identity items appended: item n_real+i belongs to segment i   */
    int64_t n_seg;             /* number of segments                                            */
    int64_t item_base;         /* filled by fn_plan_layout                                      */
    int64_t seg_base;          /* filled by fn_plan_layout                                      */
    int32_t role;              /* FN_ROLE_*                                                     */
    int32_t partner;           /* index of the paired task (roles DST/SRC), else -1             */
} fn_csr_task;

/* Host-side helper: fills item_base/seg_base, returns totals. */
int fn_plan_layout(fn_csr_task* tasks, int n_tasks, int64_t* total_items, int64_t* total_segs);

/* ws_i32 must hold FN_PLAN_WS(total_segs, total_items) int32.  ws_i32[total_segs + total_items] is a
 * status word: non-zero after the call completes if any key was outside [0, n_seg). */
#define FN_PLAN_WS(total_segs, total_items) ((total_segs) + 2 * (total_items) + 4 + 2 * ((total_segs) / 2048 + 1) + 2)
#define FN_PLAN_WS_ZEROED(total_segs, total_items) ((total_segs) + (total_items) + 4 + 2 * ((total_segs) / 2048 + 1) + 2)   /* the leading part that must be zero (FN_PLAN_PREZEROED) */
int fn_plan_build(const fn_csr_task* tasks, int n_tasks,
                  int32_t* rowptr_all /*[total_segs+1]*/, int32_t* perm_all /*[total_items]*/,
                  int32_t* aux_a /*[total_items]*/, int32_t* aux_b /*[total_items]*/, int32_t* aux_c /*[total_items]*/,
                  int32_t* ws_i32, int32_t flags /*FN_PLAN_**/, fn_stream_t stream);

/* The same plan for a MOLECULE-CONTIGUOUS batch in one launch (csrc/mol_plan.hip).  collate_fn concatenates molecules
 * (dataset/data.py:877-948): every index space (atoms, directed bonds, bond-graph edges, fragments, fragment connections,
 * fragment-bond-graph edges, molecules, ...) is a concatenation of per-molecule ranges and every key of molecule i is smaller
 * than every key of molecule i + 1, so the stable sort of each CSR is the concatenation of per-molecule sorts: one workgroup per
 * molecule sorts in LDS.  `offsets` (device, int32 [n_spaces][n_mols + 1]) = first index of molecule i in each space (the
 * collate's cumulative counts); node_space / item_space name, per task, the spaces its segments and its items live in.
 * Static-shape batches (fn_stage_padded): counts_dev = number of real molecules, offsets rows beyond it repeat the real
 * totals, cap = array lengths, pad_mod = reserved slots per space; the padding tail (item c of a field pointing into space s
 * holds cap[s] - 1 - (c - n_real) % pad_mod[s]) is written in closed form.  max_per_mol: upper bound of a molecule's extent in
 * each space (sizes the LDS tile: <= 64 KB in total, else FN_EUNSUPPORTED); a larger molecule sets status bit 2 (value 4), a key
 * outside its molecule's node range bit 1 (value 2).  pad_hint: upper bound of the padding items per space (grid sizing; 0: none).
 * Same outputs, bit for bit, as fn_plan_build; ws_i32 only carries the status word (same place). */
#define FN_MAX_SPACES 8
typedef struct fn_mol_layout {
    const int32_t* offsets;
    int32_t n_spaces, pad_;
    int64_t n_mols;
    int32_t node_space[FN_MAX_TASKS], item_space[FN_MAX_TASKS];
    const int32_t* counts_dev;             /* nullable */
    int64_t cap[FN_MAX_SPACES], pad_mod[FN_MAX_SPACES], max_per_mol[FN_MAX_SPACES], pad_hint[FN_MAX_SPACES];
} fn_mol_layout;
int fn_plan_build_mol(const fn_csr_task* tasks, int n_tasks, const fn_mol_layout* layout, int32_t* rowptr_all, int32_t* perm_all,
                      int32_t* aux_a, int32_t* aux_b, int32_t* aux_c, int32_t* ws_i32, int32_t flags, fn_stream_t stream);
/* flags: the caller has already zeroed rowptr_all[0 .. total_segs] and the whole of ws_i32 on this stream (a captured step
 * lets the staging launch in front of the replay do it, FN_STAGE_ZERO): fn_plan_build skips its zeroing launch */
#define FN_PLAN_PREZEROED 1

/* ------------------------------------------------------------------------------------------
 * One attention level (bond graph, atom graph, fragment-bond graph, fragment graph):
 *   z_e = s_dst[dst] + s_src[src] + s_edge[e];  l_e = LeakyReLU(z_e);  p = softmax over in-edges
 *   out[dst] = sum_e p_e * h[src]
 * Replaces, per level, index_select x3 + cat + mul + sum + LeakyReLU + scatter_softmax + mul +
 * scatter_add at gat2.py:146-164, 196-218, 250-267, 286-311.
 *
 * `att` is the reference's attention parameter [heads, att_w] = [dst | edge | src] blocks
 * (a_b / f_a_b: att_w = 3d; a / f: att_w = 2d + 128).
 * ------------------------------------------------------------------------------------------ */

/* s_dst[n,H] = <h[n,head,:], att[head, dst_off:+d]>, s_src likewise (gat2.py:150,208,254,300). */
int fn_node_scalars_f32(const float* h /*[n,128]*/, const float* att, int att_w, int dst_off, int src_off,
                        float* s_dst /*[n,H]*/, float* s_src /*[n,H]*/, int64_t n, int heads, fn_stream_t stream);

/* Edge term of the logit, always addressed by DESTINATION-SORTED edge position (coalesced, no
 * dependent gather).  All per-edge arrays of a level are HEAD-MAJOR ([H][m], [K][m]): the two consecutive
 * in-edges a lane owns are then one 8-byte load.
 * mode 0: s_sorted [H,m] given (fn_row_dots_sorted_f32; 0 at loop positions).
 * mode 2: s[pos,h] = <embW x_sorted[pos] + embb, att[h, mid_off:+d_e]> folded in-kernel (the reference's
 * edge_attr_bond_embed / edge_attr_fbond_embed Linear(K -> d_e), gat2.py:139,242); x_sorted is the raw
 * attribute permuted once per batch by fn_sort_edge_attr_f32 (it is the same in every layer).  A loop item's raw attribute is
 * zero (x_sorted is 0 at loop positions), so its term is <embb, att[h, mid_off:+d_e]> -- the reference adds self loops to the atom
 * graph only, which is mode 0 (a zero row there: term 0). */
typedef struct fn_edge_term {
    int32_t mode;
    int32_t K;                /* mode 2: raw attribute width (1 or 6) */
    int32_t d_e;              /* mode 2: embed width (= head_dim)     */
    int32_t mid_off;          /* mode 2: offset of the edge block in att */
    const float* s_sorted;    /* mode 0: [H, m] */
    const float* x_sorted;    /* mode 2: [K, m] */
    const float* embW;        /* mode 2: [d_e, K]    */
    const float* embb;        /* mode 2: [d_e]       */
    const float* x_src;       /* mode 2, nullable, read by fn_gat_bwd_one_f32 only: [K, m] the raw attribute in SOURCE order
                               * (fn_sort_edge_attr_src_f32); NULL: that pass gathers it from x_sorted */
} fn_edge_term;

typedef struct fn_gat_plan {          /* slices of the fn_plan_build outputs for one level */
    const int32_t* rowptr_d;  /* [n+1] global positions (DST task)  */
    const int32_t* eid_d;     /* [m]   original edge id at sorted pos */
    const int32_t* src_d;     /* [m]   source node at sorted pos      */
    const int32_t* rowptr_s;  /* [n+1] (SRC task)                     */
    const int32_t* dst_s;     /* [m]   destination node               */
    const int32_t* dpos_s;    /* [m]   position in the DST order      */
    const int32_t* inv_d;     /* [m]   original edge id -> position in the DST order */
    const int32_t* spos_d;    /* [m]   DST position -> position in the SRC order     */
    int32_t pos_base_d;       /* item_base of the DST task            */
    int32_t pos_base_s;       /* item_base of the SRC task            */
    int64_t n;                /* nodes                                */
    int64_t m;                /* edges incl. loop items               */
    int64_t m_real;           /* edges with an explicit attribute     */
} fn_gat_plan;

/* a plain CSR of fn_plan_build (segment sums, pooling, molecule membership) */
typedef struct fn_seg_plan {
    const int32_t* rowptr;    /* [n_seg+1] global positions */
    const int32_t* perm;      /* [n_items] */
    const int64_t* index;     /* [n_items] the original key (for the gather backward) */
    int64_t n_seg, n_items;
    int32_t pos_base, pad_;
} fn_seg_plan;

/* Optional fused epilogue of the forward kernel: y = relu?(dropout(out)) with the Philox stream of
 * fn_dropout_act_f32 (block index = element / 4), so the standalone backward kernel applies to it. */
typedef struct fn_act_epilogue {
    float* y;                 /* [n,128]; NULL = no fused activation */
    float p;                  /* dropout probability (0 = none)      */
    int32_t relu;
    uint64_t seed, offset;
    const uint64_t* offset_dev; /* nullable: a device-resident counter ADDED to offset when the kernel runs, so that a
                                 * captured hipGraph draws fresh masks on every replay (the owner advances it in-graph) */
} fn_act_epilogue;

/* p_sorted [H,m] (head-major): probabilities in destination-sorted order; the sign bit carries "z_e <= 0"
 * (the LeakyReLU branch) for the backward pass.  probs_orig (nullable) [m,H] in original edge
 * order, unsigned -- the reference's attn_probs. */
int fn_gat_fwd_f32(const float* h, const float* s_dst, const float* s_src, const float* att, int att_w,
                   const fn_edge_term* et, const fn_gat_plan* plan, float neg_slope,
                   float* out /*[n,128], nullable when act->y is given*/, float* p_sorted /*[H,m]*/,
                   float* probs_orig /*nullable*/,
                   float* out2 /*nullable: [n,128] = sum_e lambda_e p_e h[src_e], lambda_e = 1 (z_e > 0) or neg_slope*/,
                   float* sigma /*with out2: [n,H] = sum_e lambda_e p_e -- what fn_gat_bwd_one_f32's caller needs*/,
                   int p_edge_major /*1: p_sorted is written [m,H] (one cache line per edge, for fn_gat_bwd_one_f32) instead of [H,m]*/,
                   const fn_act_epilogue* act /*nullable*/, int heads, fn_stream_t stream);

/* Backward, destination pass.  Writes, per edge, (|p|, dz) into pz_src [H,m,2] at the edge's slot in SOURCE
 * order (so the source pass streams them), dz_sorted [H,m] in mode 0 (= dL/ds_sorted, the gradient of the edge
 * term), g_s_dst [n,H]; mode 2 writes per-block partial sums part_e [grid, H*(K+1)].
 * Returns the grid size used through *n_part_e. */
int fn_gat_bwd_dst_f32(const float* g_out, const float* h, const float* p_sorted, const fn_edge_term* et,
                       const fn_gat_plan* plan, float neg_slope,
                       float* dz_sorted /*mode 0, nullable: [H,m] destination-sorted (autograd path)*/,
                       float* g_s_orig /*mode 0, nullable: [m_real,H] original edge order (engine path)*/,
                       float* pz_src, float* g_s_dst,
                       float* part_e /*mode2: [FN_MAX_PART, H*(K+1)]*/, int* n_part_e,
                       int heads, fn_stream_t stream);

/* Backward, source pass: g_h [n,128] = sum_{e: src=n} p_e g_out[dst] + g_s_dst*a_dst + g_s_src*a_src,
 * and per-block partial sums of dL/da_dst, dL/da_src: part_a (column-major [256][FN_MAX_PART]). */
int fn_gat_bwd_src_f32(const float* g_out, const float* h, const float* pz_src,
                       const float* g_s_dst, const float* att, int att_w, int dst_off, int src_off,
                       const fn_gat_plan* plan, float* g_h, float* part_a, int* n_part_a,
                       int heads, fn_stream_t stream);

/* Backward of one attention level as ONE source-owner pass (csrc/gat_bwd_one.inc; the autograd of gat2.py:146-169, 196-219,
 * 250-268, 286-312).  Replaces fn_gat_bwd_dst_f32 + fn_gat_bwd_src_f32 when the caller supplies the two node-local dots
 *     cdot[t,h]    = <g_out[t,h,:], out[t,h,:]>                          (= sum_e p_e <g_out[t], h[src_e]>, since out = sum_e p_e h[src_e])
 *     g_s_dst[t,h] = <g_out[t,h,:], out2[t,h,:]> - cdot[t,h] sigma[t,h]  (= sum_{e -> t} dz_e)
 * (fn_gat_cu_f32, or the epilogue of the input-gradient GEMM inside fn_encoder_backward).  Outputs as the two passes': g_h
 * [n,128]; mode 0: dz_sorted [H,m] (destination order) and / or g_s_orig [m_real,H] (original edge order), both nullable;
 * mode 2: part_e [*n_part_e, H*(K+1)]; part_a column-major [256][FN_MAX_PART] with *n_part_a rows, for fn_gat_bwd_finalize_f32. */
int fn_gat_bwd_one_f32(const float* g_out, const float* h, const float* p_sorted, const float* cdot, const float* g_s_dst,
                       const fn_edge_term* et, const float* att, int att_w, int dst_off, int src_off, const fn_gat_plan* plan,
                       float neg_slope, float* g_h, float* dz_sorted, float* g_s_orig, float* part_a, int* n_part_a,
                       float* part_e, int* n_part_e, int p_edge_major /*layout of p_sorted, as fn_gat_fwd_f32 wrote it*/,
                       float* dz_em /*nullable; non-null (four heads): the DEFERRED form, see below*/, int heads, fn_stream_t stream);
/* The deferred form (ABI 11; what fn_encoder_backward runs for four heads when FN_TUNE_DEFER_GSD is set -- it is OFF by default): g_s_dst is not read and the forward
 * needs no out2 / sigma.  The pass writes dz of every edge at its destination-order slot, dz_em [m,4] edge-major, and leaves the two
 * terms that need g_s_dst[t,h] = the sum of row t's contiguous dz_em segment OUT of its results: g_h lacks g_s_dst[s] a_dst, and the
 * a_dst columns of part_a are zero.  fn_gat_gsd_f32 forms g_s_dst [n,4] from dz_em and overwrites those columns of part_a (rows
 * 0 .. n_part_a-1) with the partials of dL/da_dst = sum_t g_s_dst[t,h] h[t, head h's columns]; the caller adds g_s_dst[s] a_dst to
 * g_h (inside the engine the consumers of g_h do: the input-gradient product as one more MFMA step, the weight-gradient kernels
 * on their dY operand). */
int fn_gat_gsd_f32(const float* dz_em, const fn_gat_plan* plan, const float* h, float* g_s_dst, float* part_a, int n_part_a,
                   fn_stream_t stream);
/* c[t,h] = scale <g_out[t,h,:], out[t,h,:]>, u[t,h] = <g_out[t,h,:], out2[t,h,:]> - c[t,h] sigma[t,h] for n rows.  `out` may be the
 * level's relu(dropout(.)) output with scale = 1 - p when g_out reaches the rows through that gate only. */
int fn_gat_cu_f32(const float* g_out, const float* out, const float* out2, const float* sigma, float scale, float* c, float* u,
                  int64_t n, int heads, fn_stream_t stream);

/* Reduces the partials into g_att [H, att_w] (dst/src blocks, and the edge block in mode 2) and,
 * in mode 2, g_embW [d_e,K], g_embb [d_e].  g_att must be zero-initialised by the caller. */
int fn_gat_bwd_finalize_f32(const float* part_a, int n_part_a, const float* part_e, int n_part_e,
                            const fn_edge_term* et, const float* att, int att_w, int dst_off, int src_off,
                            float* g_att, float* g_embW, float* g_embb, int heads, fn_stream_t stream);

#define FN_MAX_PART 4096      /* upper bound on partial rows any kernel writes */

/* attention mass per SOURCE node: scatter_add(attn_probs, source) at gat2.py:165,219,268,312 */
int fn_attn_by_src_f32(const float* p_sorted, const fn_gat_plan* plan, float* attn /*[n,H]*/, int heads,
                       fn_stream_t stream);

/* s_sorted[j, pos] = <feat[eid(pos), 0:128], A[j*lda + off : +128]>, j < J <= 8, 0 at loop positions: the
 * full-width edge term of the atom and fragment graphs (edge block of `a` / `f`, gat2.py:203-208, 293-300),
 * written directly in destination-sorted order. */
int fn_row_dots_sorted_f32(const float* feat /*[m_real,128]*/, const float* A, int lda, int off, int J,
                           const fn_gat_plan* plan, float* s_sorted /*[J,m] head-major*/, fn_stream_t stream);
/* g_feat[e,:] = sum_j g_s_sorted[j, inv_d[e]] A[j]; part: column-major partial sums of g_A[j,:] */
int fn_row_dots_sorted_bwd_f32(const float* g_s_sorted, const float* feat, const float* A, int lda, int off, int J,
                               const fn_gat_plan* plan, float* g_feat, float* part, int* n_part, fn_stream_t stream);
/* x_sorted[:,pos] = x[eid(pos),:] (0 at loop positions): once per batch for the raw edge attributes */
int fn_sort_edge_attr_f32(const float* x /*[m_real,K]*/, int K, const fn_gat_plan* plan, float* x_sorted /*[K,m]*/,
                          fn_stream_t stream);
/* x_src[:,q] = x[edge at SOURCE-order position q,:] (0 for loop items): the copy the one-pass backward streams */
int fn_sort_edge_attr_src_f32(const float* x /*[m_real,K]*/, int K, const fn_gat_plan* plan, float* x_src /*[K,m]*/,
                              fn_stream_t stream);
/* out[(c / 128) * ld + off + c % 128] = sum_{r < n_rows} part[c * FN_MAX_PART + r], c < cols (partials are column-major) */
int fn_colsum_f32(const float* part, int n_rows, int cols, float* out, int ld, int off, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Node projections projection_b / projection_a / projection_fb (nn.Linear(K -> 128), gat2.py:142,189,247)
 * on the fp32 matrix cores (exact fp32 FMA chains).  K <= 168.
 *   fn_transpose_w_f32      Bt[K,128] = W[128,K]^T            (once per step per weight)
 *   fn_linear128_f32        Y[M,128] = X[M,K] Bt + bias        (forward; input gradient with Bt = W, bias = NULL)
 *   fn_linear128_wgrad_f32  dW[128,K] = dY^T X, db[128] = colsum(dY); ws holds fn_linear128_wgrad_ws(M,K) floats
 * ------------------------------------------------------------------------------------------ */
int fn_transpose_w_f32(const float* W, int K, float* Bt, fn_stream_t stream);
int fn_linear128_f32(const float* X, int K, const float* Bt, const float* bias /*nullable*/, float* Y, int64_t M,
                     const fn_act_epilogue* act_bwd /*nullable: Y *= dropout mask * (act_bwd->y > 0), the backward of
                     act(dropout(.)) fused into an input-gradient GEMM.  With relu set, act_bwd->y must be the SAVED OUTPUT
                     relu(dropout(x)) of that stream: it is positive exactly where the element was kept and passed the ReLU,
                     so the kernel scales by 1/(1-p) where y > 0 and does not replay the Philox stream*/, fn_stream_t stream);
int64_t fn_linear128_wgrad_ws(int64_t M, int K);
int fn_linear128_wgrad_f32(const float* dY, const float* X, int K, int64_t M, float* ws, float* dW, float* db,
                           fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * torch_scatter.scatter_add / scatter_softmax along dim 0 on a CSR built by fn_plan_build
 * (gat2.py:234 atom->fragment sum; gat2.py:820-821 and pretrain_heads.py:93-94 pooling).
 * ------------------------------------------------------------------------------------------ */
int fn_segment_sum_f32(const float* src /*[items, src_ld >= width]*/, int64_t src_ld, const int32_t* rowptr,
                       const int32_t* perm, int32_t pos_base, float* out /*[n_seg,width]*/, int64_t n_seg,
                       int64_t width, int64_t n_items /*picks the long-segment kernel*/, fn_stream_t stream);
/* backward of the above and of index_select: out[i,:] = table[index[i],:] */
int fn_gather_rows_f32(const float* table, const int64_t* index, float* out, int64_t rows, int64_t width,
                       fn_stream_t stream);
/* probs[item,:] = softmax of logits over the items of each segment, per column (original item order) */
int fn_segment_softmax_f32(const float* logits, const int32_t* rowptr, const int32_t* perm, int32_t pos_base,
                           float* probs, int64_t n_seg, int64_t width, fn_stream_t stream);
/* g_logits = p * (g_p - sum_seg p*g_p) */
int fn_segment_softmax_bwd_f32(const float* probs, const float* g_probs, const int32_t* rowptr, const int32_t* perm,
                               int32_t pos_base, float* g_logits, int64_t n_seg, int64_t width, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense layers of the prediction heads, FTHead1-5 (gat2.py:569-751): Linear -> relu(dropout(.)) on [molecules, width],
 * fp32 matrix cores, 32x64 output tiles, the element-wise work fused in (csrc/dense_head.inc).
 *   fn_dense_fwd_f32   Y[M,N] = X[M,K] W[N,K]^T + bias;  with act: Y = relu?(dropout(.)), the Philox stream of
 *                      fn_dropout_act_f32 over Y's elements (act->y is ignored: the activation is applied in place)
 *   fn_dense_bwd_f32   g_y = dL/d(X W^T + bias), i.e. already through the backward of this layer's own activation;
 *                      dW[N,K] = g_y^T X;  db[N] = column sums of g_y (nullable);  g_x[M,K] = g_y W (nullable: first
 *                      layer), and with gate_scale > 0 additionally g_x = X > 0 ? g_x * gate_scale : 0 -- the backward of
 *                      the layer BELOW's relu(dropout(.)), whose saved output X is (gate_scale = 1 / (1 - p)), so the next
 *                      call receives its g_y ready.  One launch.
 * K and N multiples of 4 and <= 65536, M <= FN_DENSE_MAX_ROWS (one workgroup reduces over all of M: taller inputs go through
 * fn_gate_colsum_f32 + library GEMMs).  With M = 0, fn_dense_bwd_f32 writes zeros to dW and db.
 * ------------------------------------------------------------------------------------------ */
#define FN_DENSE_MAX_ROWS 4096
int fn_dense_fwd_f32(const float* X, const float* W, const float* bias /*nullable*/, float* Y, int64_t M, int64_t K, int64_t N,
                     const fn_act_epilogue* act /*nullable*/, fn_stream_t stream);
int fn_dense_bwd_f32(const float* g_y, const float* X, const float* W, float* g_x /*nullable, [max(M,M_out),K]*/, float gate_scale /*0: none*/,
                     float* dW, float* db /*nullable*/, int64_t M, int64_t K, int64_t N,
                     int64_t M_out /*rows [M, M_out) of g_x are set to 0 (padding rows); <= M: none*/, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * act(dropout(x)) between layers (gat2.py:396-397, 414-418, 436-440): Philox-4x32 mask (seven rounds, csrc/fn_internal.h) keyed by
 * (seed, offset + element/4), y = relu(keep ? x/(1-p) : 0); relu = 0 gives plain dropout.
 * Backward recomputes the mask from the same (seed, offset).
 * ------------------------------------------------------------------------------------------ */
int fn_dropout_act_f32(const float* x, float* y, int64_t numel, float p, uint64_t seed, uint64_t offset,
                       const uint64_t* offset_dev /*nullable, see fn_act_epilogue*/, int relu, fn_stream_t stream);
int fn_dropout_act_bwd_f32(const float* g_y, const float* y, float* g_x, int64_t numel, float p, uint64_t seed,
                           uint64_t offset, const uint64_t* offset_dev, int relu, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Bond-graph topology from edge_index (dataset-side; reference fragnet/dataset/data.py:116-127 get_bond_pair_bond_graph,
 * :157-182 one-bond fragments, :403-410): edge_index_bonds_graph [2, Eb] = ordered pairs (i, j) of directed bonds of a
 * molecule sharing exactly one atom, i-major / j ascending, then per molecule the mutual pairs of its two-atom
 * components (lowest atom first).  Bond id = column of the batched, molecule-contiguous edge_index [2, E];
 * atom_mol [N] = molecule of every atom (the batch vector).  Two calls because Eb is only known on the device:
 * count (fills ws, writes *total = Eb), then fill into out [2, total].  The cos(theta) attribute needs coordinates
 * and is not produced.  ws: fn_bond_graph_ws(E, B) int32.
 * ------------------------------------------------------------------------------------------ */
/* mode FN_GRAPH_BONDS: the rule above on edge_index / batch.  mode FN_GRAPH_FBONDS: the fragment-bond graph of
 * data.py:131-154 on frag_index / frag_batch -- a molecule with exactly two connection nodes pairs the ones whose
 * (begin, end) differ, every other molecule uses the share-exactly-one rule, no extras.  (Its edge attribute is the sum
 * of the two node features, data.py:291-303, a plain gather-add.) */
#define FN_GRAPH_BONDS 0
#define FN_GRAPH_FBONDS 1
int64_t fn_bond_graph_ws(int64_t E, int64_t B);
int fn_bond_graph_count(const int64_t* edge_index /*[2,E]*/, const int64_t* atom_mol /*[N]*/, int64_t E, int64_t N, int64_t B,
                        int mode, int32_t* ws, int64_t* total /*device [1]*/, fn_stream_t stream);
int fn_bond_graph_fill(const int64_t* edge_index, const int64_t* atom_mol, int64_t E, int64_t N, int64_t B, int mode,
                       const int32_t* ws, int64_t* out /*[2,total]*/, int64_t total, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Prediction-head small ops (FTHead1-5, gat2.py:631-637, 719-725, 745-751: Linear -> dropout -> act stacks on
 * [molecules, width]; the dense products themselves stay library GEMMs).
 * fn_gate_colsum_f32: backward of relu(dropout(.)) fused with the bias gradient of the Linear below it:
 *   g_x = (y > 0) ? g_y * scale : 0 (scale = 1/(1-p); the saved output encodes the mask), colsum[c] = sum_rows g_x[:, c].
 * fn_small_linear(_bwd)_f32: the last Linear of a head (n_classes <= FN_SMALL_LINEAR_MAX outputs) as one launch each
 *   way: y = x w^T + b;  g_x = g w (optionally gated by x > 0), dW = g^T x, db = colsum(g).  All sums run in a fixed order.
 * ------------------------------------------------------------------------------------------ */
#define FN_SMALL_LINEAR_MAX 16
/* Inputs taller than 2048 rows (the pretrain towers run on every edge / atom) are reduced in row chunks: pass a float
 * workspace of fn_gate_colsum_ws() / fn_small_linear_bwd_ws() elements (0 = not needed, ws may be NULL). */
int64_t fn_gate_colsum_ws(int64_t rows, int64_t cols);
int fn_gate_colsum_f32(const float* g_y /*[rows,cols]*/, const float* y /*[rows,cols]*/, float* g_x /*[rows,cols]*/,
                       float* colsum /*[cols]*/, int64_t rows, int64_t cols, float scale, float* ws, fn_stream_t stream);
int fn_small_linear_f32(const float* x /*[M,K]*/, const float* w /*[C,K]*/, const float* b /*[C] nullable*/, float* y /*[max(M,M_out),C]*/,
                        int64_t M, int64_t K, int64_t C, int64_t M_out /*rows [M, M_out) of y are set to 0: the padding
                        molecules of a static-shape batch; <= M: none*/, fn_stream_t stream);
int64_t fn_small_linear_bwd_ws(int64_t M, int64_t K, int64_t C);
int fn_small_linear_bwd_f32(const float* g /*[M,C]*/, const float* x /*[M,K]*/, const float* w /*[C,K]*/, float* g_x /*[M,K]*/,
                            float* dW /*[C,K]*/, float* db /*[C]*/, int64_t M, int64_t K, int64_t C,
                            float gate_scale /*0: none; > 0: g_x = x > 0 ? g_x * gate_scale : 0, the backward of the
                            relu(dropout(.)) that produced x (see fn_dense_bwd_f32)*/, float* ws, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The last Linear of a head + the loss on its outputs + that Linear's input gradient as ONE launch, with the sums that cross rows
 * riding in the head's first fn_dense_bwd launch (round 4).  Replaces, with the same numbers, the sequence
 *   fn_small_linear_f32 -> fn_masked_mse_f32 / fn_masked_bce_f32 -> fn_small_linear_bwd_f32
 * of a TRAINING step whose upstream gradient is d loss / d loss = 1 (train/utils.py:341 MSELoss()(out.view(-1), y), :297-304
 * compute_bce_loss, on FTHead1-5's last Linear, gat2.py:631-637).
 *   fn_small_linear_loss_f32   y[max(M,M_out),C] = x w^T + b (rows >= M: 0);  g[M,C] = d loss / d y;  g_x[M,K] = g w, gated by x > 0 when
 *                              gate_scale > 0;  loss_part[fn_small_linear_loss_ws(M_out)] = per-workgroup partial sums of the loss
 *                              (already divided by the denominator: their sum IS the loss).  kind FN_LOSS_MSE: loss =
 *                              sum_i row_w[i] sum_c (y - target)^2 / (sum_i row_w[i] * C);  FN_LOSS_BCE: BCE-with-logits averaged over
 *                              the entries with target > -0.5 and row_w > 0.  row_w [max(M,M_out)], target [max(M,M_out),C].
 *   fn_dense_bwd_tail_f32      fn_dense_bwd_f32 whose launch also runs  dW[C,K] = g^T x,  db[C] = colsum(g)  and
 *                              loss[0] = sum(loss_part)  (fixed-order sums) for the fused launch above; tail == NULL: fn_dense_bwd_f32.
 * ------------------------------------------------------------------------------------------ */
#define FN_LOSS_MSE 0
#define FN_LOSS_BCE 1
#define FN_SMALL_LINEAR_LOSS_MAX_K 1024
typedef struct fn_small_dw {
    const float* g;          /* [M,C]  d loss / d y of the last Linear */
    const float* x;          /* [M,K]  its input */
    float* dW;               /* [C,K] */
    float* db;               /* [C] */
    const float* loss_part;  /* [n_part], nullable with loss == NULL */
    float* loss;             /* [1], nullable */
    int64_t n_part, M, K, C;
} fn_small_dw;
int64_t fn_small_linear_loss_ws(int64_t M_out);
int fn_small_linear_loss_f32(const float* x /*[M,K]*/, const float* w /*[C,K]*/, const float* b /*[C] nullable*/, const float* target,
                             const float* row_w, int kind, float* y, float* g, float* g_x, float gate_scale /*0: none*/, float* loss_part,
                             int64_t M, int64_t K, int64_t C, int64_t M_out, fn_stream_t stream);
int fn_dense_bwd_tail_f32(const float* g_y, const float* X, const float* W, float* g_x, float gate_scale, float* dW, float* db,
                          int64_t M, int64_t K, int64_t N, int64_t M_out, const fn_small_dw* tail /*nullable*/, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A batch out of a resident flat store in one launch (the data side of the path: the reference's collate_fn, dataset/data.py:877-948,
 * for molecules that are kept concatenated in HBM -- fragnet_amd.dataset.FlatMolStore).  Batch molecule b = store molecule idx[b].
 *   offsets [n_spaces, B + 1] int32: first batch row of molecule b in every index space (plan.CollatedBatch.offsets: the caller has the
 *                                     store's molecule lengths on the host and builds it there)
 *   starts  [n_spaces, B]     int64: first STORE row of molecule idx[b] in every index space
 * Every field is one output tensor in index space `space`:
 *   FN_COLLATE_ROWS   dst [rows, width_words x 4 bytes] = the molecules' store rows, in batch order (features, labels: any 4-byte type)
 *   FN_COLLATE_BATCH  dst [rows] int64 = the batch molecule of every row (`batch`, `frag_batch`)
 *   FN_COLLATE_IDS    dst [width_words, rows] int64 out of src [width_words, src_rows]: index tensors (edge_index, frag_index, the two
 *                     bond-graph indices: width 2; atom_id_frag_id: width 1) whose values point into `rebase_space`: batch value =
 *                     stored value - (src_global ? first store row of the molecule there : 0) + first batch row of the molecule there
 * ------------------------------------------------------------------------------------------ */
#define FN_MAX_COLLATE_FIELDS 24
#define FN_COLLATE_ROWS 0
#define FN_COLLATE_BATCH 1
#define FN_COLLATE_IDS 2
typedef struct fn_collate_field {
    const void* src;
    void* dst;
    int64_t rows;            /* rows of dst in `space` */
    int64_t src_rows;        /* FN_COLLATE_IDS: rows of src (its second dimension) */
    int32_t width_words;     /* ROWS: 4-byte words per row; IDS: first dimension (1 or 2) */
    int32_t space, kind, rebase_space;
    int32_t src_global;
    int32_t max_seg_rows;    /* upper bound on one molecule's rows in `space` (the store's maximum); 0: unknown.  Sizes the launch only: a
                              * molecule's segment is cut into chunks that half-waves copy side by side */
} fn_collate_field;
int fn_collate_store(const fn_collate_field* fields, int n_fields, const int64_t* starts, const int32_t* offsets, int n_spaces, int64_t B,
                     const void* tables_host /*nullable: pinned host memory holding [starts | offsets]; a kernel then copies it into
                     `starts` (one device allocation that continues into `offsets`) in front of the collate launch -- no copy-engine
                     transfer on the step's stream*/, fn_stream_t stream);

/* torch.optim.Adam step (no amsgrad) on one flat fp32 tensor: finetune_gat2.py:257, pretrain_gat2.py:165.
 * `step` is the 1-based step count (bias corrections are computed on the host in double). */
int fn_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                float weight_decay, int64_t step, fn_stream_t stream);
/* the same update with the 1-based step count and the learning rate read from device memory (bias corrections computed
 * in the kernel), so that the launch can sit inside a captured hipGraph and still see a new step / rate on every replay */
int fn_adam_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev /*[1]*/, float beta1,
                    float beta2, float eps, float weight_decay, const int64_t* step_dev /*[1]*/, fn_stream_t stream);

/* cat(x[src_e], x[dst_e], e_attr[e]) -> [E, 384] for the bond-length head (pretrain_heads.py:67-70) */
int fn_edge_concat_f32(const float* x /*[N,128]*/, const float* e_attr /*[E,128]*/, const int64_t* edge_index /*[2,E]*/,
                       float* out /*[E,384]*/, int64_t E, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Readout and loss.  pooled = cat(scatter_add(x_atoms, batch), scatter_add(x_frags, frag_batch)) (gat2.py:820-823)
 * as one launch into a [B,256] buffer, its backward as one launch; the molecule-weighted MSE (MSELoss of
 * train/utils.py:341 over the rows with weight 1) with its gradient in one single-block launch.
 * ------------------------------------------------------------------------------------------ */
struct fn_seg_plan;
int fn_pool_cat_f32(const float* x_atoms /*[N,128]*/, const float* x_frags /*[F,128]*/, const struct fn_seg_plan* mol_atoms,
                    const struct fn_seg_plan* mol_frags, float* out /*[B,256]*/, fn_stream_t stream);
int fn_pool_cat_bwd_f32(const float* g /*[B,256]*/, const int64_t* batch /*[N]*/, const int64_t* frag_batch /*[F]*/,
                        float* g_atoms /*[N,128]*/, float* g_frags /*[F,128]*/, int64_t N, int64_t F, fn_stream_t stream);
int fn_masked_mse_f32(const float* out /*[B,T]*/, const float* y /*[B,T]*/, const float* w /*[B]*/, int64_t B, int T,
                      float* loss /*[1]*/, float* g_out /*[B,T] = dloss/dout*/, fn_stream_t stream);
/* compute_bce_loss (train/utils.py:297-304): mean over the valid entries (y > -0.5, w > 0) of BCE-with-logits(out, max(y, 0)),
 * and its gradient; one single-block launch (B*T is ~12 k for Tox21 at batch 1024). */
int fn_masked_bce_f32(const float* out /*[B,T]*/, const float* y /*[B,T]*/, const float* w /*[B]*/, int64_t B, int T,
                      float* loss /*[1]*/, float* g_out /*[B,T]*/, fn_stream_t stream);
/* loss = sum_k coef_k * masked_mse_k with coef_k = tasks[k].coef * (scale_dev[tasks[k].scale_idx] if scale_idx >= 0 else 1):
 * the pretrain loss 2*MSE(dihedral) + MSE(angle) + MSE(energy) (pretrain_utils.py:9-31) with the per-rank weights of the
 * per-edge / per-atom means, and all three gradients, in two multi-block launches.  ws: fn_masked_mse_multi_ws(n) floats. */
typedef struct fn_mse_task {
    const float *out, *y, *w;   /* [B,T], [B,T], [B] */
    float* g_out;               /* [B,T] = d loss / d out */
    int64_t B;
    int32_t T, scale_idx;
    float coef, pad_;
} fn_mse_task;
int64_t fn_masked_mse_multi_ws(int n_tasks);
int fn_masked_mse_multi_f32(const fn_mse_task* tasks, int n_tasks /*<= 4*/, const float* scale_dev /*nullable*/, float* ws,
                            float* loss /*[1]*/, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Static-shape staging for hipGraph replay.  A training step captured in a hipGraph has fixed tensor shapes, so
 * every batch (the dict of dataset/data.py:931-948) is copied into fixed-capacity buffers and the tail of each
 * buffer is filled with PADDING that is itself a valid, disconnected piece of graph: zero feature rows, and index
 * values pointing at the last `pad_mod` slots of the target index space (pad value at position i =
 * pad_hi - (i - n_real) % pad_mod: the first padding item points at the last slot), so padding only ever talks to padding and
 * in-degrees stay small.  One launch, all fields.
 * ------------------------------------------------------------------------------------------ */
#define FN_MAX_STAGE_FIELDS 40
#define FN_STAGE_ROWS 0 /* float32 [cap,width]  <- [n_real,width], zero rows after                         */
#define FN_STAGE_IDS 1  /* int64   [cap]        <- [n_real], pad ids after                                 */
#define FN_STAGE_COLS 2 /* int64   [2,cap]      <- [2,n_real] (edge_index layout), pad ids in both rows    */
#define FN_STAGE_MASK 3 /* float32 [cap]        =  1 for i < n_real, 0 after (loss weights); src unused    */
#define FN_STAGE_COUNT 4 /* int32 [1]           =  n_real (device-side copy of a count for the fused encoder)  */
#define FN_STAGE_ZERO 6  /* int32 [cap]         =  0 (workspace of a captured fn_plan_build, see FN_PLAN_PREZEROED); src unused */
#define FN_STAGE_OFFSETS 7 /* int32 [width][cap+1] <- [width][n_real+1]: per-molecule offsets of `width` index spaces (fn_mol_layout);
                            *                       entries behind molecule n_real repeat the space's total */
#define FN_STAGE_BUMP 5  /* int64 [1]          +=  n_real: a device-side counter of a captured step (Philox blocks, optimiser steps)
                          *                       advanced by the staging launch that precedes every replay instead of by a launch of its own */
typedef struct fn_stage_field {
    const void* src;
    void* dst;
    int64_t n_real, cap;
    int32_t width, kind;
    int64_t pad_hi, pad_mod;
} fn_stage_field;
int fn_stage_padded(const fn_stage_field* fields, int n_fields, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PretrainTask's tall towers (pretrain_heads.py:33-58, 77-88: `PretrainTask(128, 1)`, L = 2): Linear(128 -> 64) -> ReLU ->
 * Linear(64 -> 32) -> ReLU -> Linear(32 -> 1) on every atom (bond angle) / every directed bond (dihedral).  One launch each way
 * for up to FN_MAX_TOWERS towers (+ one reduction launch in the backward): the 41 KB of weights live in LDS, a workgroup walks
 * 32-row tiles, the hidden rows never leave the CU in the forward except as the saved h1 / h2.
 * fn_tower_bwd_f32: g_x [M,128] = dL/dx (nullable), g_w* / g_b* = parameter gradients (overwritten), ws = fn_tower_bwd_ws() floats.
 * ------------------------------------------------------------------------------------------ */
#define FN_MAX_TOWERS 4
typedef struct fn_tower {
    const float* x;                              /* [M,128] */
    const float *w1, *b1, *w2, *b2, *w3, *b3;    /* [64,128], [64], [32,64], [32], [1,32], [1] */
    float *h1, *h2;                              /* [M,64], [M,32]: relu outputs, written by the forward, read by the backward */
    float* out;                                  /* [M,1] (forward) */
    int64_t M;
    const float* g_out;                          /* [M,1] (backward) */
    float* g_x;
    float *g_w1, *g_b1, *g_w2, *g_b2, *g_w3, *g_b3;
} fn_tower;
int fn_tower_fwd_f32(const fn_tower* towers, int n, fn_stream_t stream);
int64_t fn_tower_bwd_ws(const fn_tower* towers, int n);
int fn_tower_bwd_f32(const fn_tower* towers, int n, float* ws, fn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Encoder engine: the reference's FragNet.forward (gat2.py:381-442: L x FragNetLayerA + act(dropout(.))) and
 * its backward pass as one call each.  The host side only walks the layers and enqueues kernels on `stream`;
 * nothing is allocated, nothing synchronises.  `ws` keeps the activations the backward pass reads.
 * Gradient pointers in `grads[l]` that the pass does not produce are left untouched: `f` for every layer but the
 * last (the fragment-graph output of inner layers is dead in the reference).
 * ------------------------------------------------------------------------------------------ */
#define FN_MAX_LAYERS 8

typedef struct fn_layer_weights {          /* parameters of one FragNetLayerA, or their gradients */
    float *proj_b_w, *proj_b_b;            /* projection_b  [128,Kb], [128]  (gat2.py:88)  */
    float *proj_a_w, *proj_a_b;            /* projection_a  [128,Ka], [128]  (gat2.py:95)  */
    float *proj_fb_w, *proj_fb_b;          /* projection_fb [128,Kfb], [128] (gat2.py:89)  */
    float *emb_b_w, *emb_b_b;              /* edge_attr_bond_embed  [d,1], [d]  (gat2.py:91) */
    float *emb_fb_w, *emb_fb_b;            /* edge_attr_fbond_embed [d,Kf], [d] (gat2.py:92) */
    float *a_b, *a, *f, *f_a_b;            /* attention vectors (gat2.py:98-109) */
} fn_layer_weights;

/* a slice of FlatAdam's buffers with the step count and the learning rate in device memory (see fn_adam_dev_f32) */
typedef struct fn_adam_slice {
    float* p;  const float* g;  float* m;  float* v;
    int64_t n;                       /* elements; p, g, m, v 16-byte aligned */
    const float* lr_dev;  const int64_t* step_dev;
    float beta1, beta2, eps, weight_decay;
    int32_t launched, pad_;          /* OUT (fn_encoder.adam_rider only): set to 1 by fn_encoder_backward once the launch that carries
                                      * the slice has been enqueued; the caller clears it before the call and updates the slice itself
                                      * when it is still 0 afterwards (ABI 11) */
} fn_adam_slice;

typedef struct fn_encoder {
    int32_t n_layers, heads;
    int32_t k_atom0, k_bond0, k_fbond0;    /* layer-0 feature widths (167, 17, 6); 128 afterwards */
    int32_t k_fattr;                       /* width of edge_attr_fbonds (6) */
    int32_t training;
    int32_t variant;                       /* 0: gat2; 1: gat2_lite (gat2_lite.py: every layer stops after the atom -> fragment sum;
                                            * out_fbond is not written, out_frags = relu(dropout(fragment sums)));
                                            * 2: gat2_edge (gat2_edge.py: no fragment-bond graph; the fragment graph's edge term is
                                            * <Linear(k_fattr -> 128)(cnx_attr), f[:, d:d+128]> with the Linear in w[l].emb_fb_w/_b
                                            * ([128, k_fattr], [128]) and cnx_attr sorted by the FRAGMENT graph in fattr_sorted
                                            * [k_fattr, frag.m]; proj_fb_* and f_a_b are placeholders, out_fbond is not written) */
    float drop_p, pad2_;
    uint64_t seed, offset;                 /* Philox stream; fn_encoder_rng_blocks() offsets are consumed */
    const uint64_t* offset_dev;            /* nullable device counter added to offset at run time (hipGraph replays) */
    int64_t N, E, F, EF;
    fn_gat_plan bond, atom, fbond, frag;
    fn_seg_plan a2f;
    const float *x_atoms, *bond_nodes, *fbond_nodes;     /* layer-0 inputs */
    const float *cos_sorted, *fattr_sorted;              /* fn_sort_edge_attr_f32 outputs: [1, bond.m] and [k_fattr, fbond.m] */
    const float *cos_raw, *fattr_raw;                    /* nullable.  When set (edge order of the batch: [bond.m_real] and
                                                          * [fbond.m_real, k_fattr]) fn_encoder_forward permutes them into
                                                          * cos_sorted / fattr_sorted itself, inside its one prologue launch;
                                                          * the two *_sorted buffers must then be writable */
    fn_layer_weights w[FN_MAX_LAYERS];
    float* ws;
    int64_t ws_floats;                     /* >= fn_encoder_ws_floats() */
    /* Molecule CSRs (optional; n_mols = 0: absent).  collate_fn concatenates molecules, so the atoms / bonds / fragments /
     * connections / graph edges of molecule i are contiguous ranges (dataset/data.py:877-948).  They drive the molecule-resident
     * fragment tail (below) and the padding-row skip (FN_TUNE_PAD_SKIP). */
    fn_seg_plan mol_atoms, mol_frags;      /* atoms / fragments keyed by molecule (batch, frag_batch: gat2.py:820-821) */
    int64_t n_mols;
    const int32_t* counts_dev;             /* nullable device [1]: number of REAL molecules (the first ones) when the batch is
                                            * padded to static shapes (fn_stage_padded, FN_STAGE_COUNT); outputs of padding rows
                                            * are zero */
    int32_t* status;                       /* nullable device word, bits OR-ed in on malformed batches */
    /* Fused fragment tail (csrc/mol_tail.inc).  mol_contiguous: the caller's word that the batch has collate_fn's layout
     * (molecules concatenated: every index range of a molecule is contiguous and the molecule CSRs above are given).  Then the
     * last layer's atom -> fragment sum, fragment-graph attention and -- when `pooled` is set -- the readout
     * cat(scatter_add(out_atoms, batch), scatter_add(out_frags, frag_batch)) [n_mols, 256] (gat2.py:820-823) run as ONE
     * molecule-resident launch, and their backward (with dL/d(pooled) in `g_pooled`, nullable, added to g_atoms / g_frags) as
     * one more.  fn_encoder_fused_tail() says whether a descriptor takes that path; `pooled` / `g_pooled` must be NULL if not. */
    int32_t mol_contiguous;
    /* no_backward (was padding: 0 keeps the old behaviour): the caller's word that no fn_encoder_backward will follow this
     * fn_encoder_forward (torch.no_grad() / nothing requires a gradient).  An EVALUATION pass (training == 0) then saves nothing
     * for one: the attention probabilities of its levels are not stored, nor the raw bond rows once their only forward reader --
     * the atom graph's edge term -- was folded into the bond level's launch (FN_TUNE_FUSE_ROWDOTS).  Outputs are bit-identical;
     * fn_encoder_backward on such a descriptor returns FN_EINVAL.  Ignored by training passes. */
    int32_t no_backward;
    float* pooled;
    const float* g_pooled;
    /* fn_encoder_backward only, nullable: an Adam update (fn_adam_dev_f32's arithmetic) of a slice of the flat parameter buffer whose
     * gradients were complete before this backward pass began -- the prediction head's -- rides in the pass's last launch (the
     * deferred reductions), independent of everything that launch reduces.  The caller's own Adam launch then covers the rest
     * (torch.optim.Adam is element-wise: finetune_gat2.py:257).  ABI 10. */
    struct fn_adam_slice* adam_rider;
} fn_encoder;

int fn_encoder_fused_tail(const fn_encoder* e);      /* 1: fn_encoder_forward / _backward run the fused fragment tail */

int64_t fn_encoder_ws_floats(const fn_encoder* e);
int64_t fn_encoder_bwd_ws_floats(const fn_encoder* e);
uint64_t fn_encoder_rng_blocks(const fn_encoder* e);
/* out_bond and out_fbond may both be NULL: the caller reads neither (a finetune head pools atoms and fragments only, gat2.py:816-826)
 * and the last layer's activated bond / fragment-bond rows are not stored (the levels themselves still run: the atom and fragment
 * graphs read their raw rows).  The other outputs do not change a bit. */
int fn_encoder_forward(const fn_encoder* e, float* out_atoms /*[N,128]*/, float* out_frags /*[F,128]*/,
                       float* out_bond /*[E,128], nullable*/, float* out_fbond /*[EF,128], nullable*/, fn_stream_t stream);
/* g_* are dL/d(out_*) (nullable = zero); out_* are the forward outputs (needed for the ReLU mask; out_bond / out_fbond may be
 * NULL where the forward pass was given NULL -- their gradients must then be NULL too). */
int fn_encoder_backward(const fn_encoder* e, const float* out_atoms, const float* out_frags, const float* out_bond,
                        const float* out_fbond, const float* g_atoms, const float* g_frags, const float* g_bond,
                        const float* g_fbond, const fn_layer_weights* grads /*[n_layers]*/, float* scratch,
                        int64_t scratch_floats, fn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FRAGNET_HIP_H */
