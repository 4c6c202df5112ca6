/*
 * psg.h -- C ABI of libpsg.so: the MI355X (gfx950) implementation of PointSecGuard's
 * data-parallel hot path (PointNet++ SSG semantic-segmentation forward + colour-gradient backward
 * inside the NB/NU colour-perturbation attack loops).
 *
 * The reference has no FFI layer: its boundary for this path is the Python API that the harness
 * scripts import.  Each entry point below names the reference expression it
 * replaces (paths relative to the reference checkout); INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless marked "host";
 *   - every call is stream-ordered and non-blocking on `stream` (a hipStream_t, 0 = default
 *     stream); no call synchronises the device or spawns threads;
 *   - objects (ctx / model / workspace) allocate their device memory at create time only;
 *   - return value: PSG_OK (0) or a negative PSG_ERR_*; psg_last_error() gives the thread-local text;
 *   - feature tensors are POINT-MAJOR fp32: [batch][points][channels]; indices are int32.
 */
#ifndef PSG_H
#define PSG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSG_OK 0
#define PSG_ERR_ARG (-1)   /* bad argument / unsupported size */
#define PSG_ERR_HIP (-2)   /* a HIP runtime call failed */
#define PSG_ERR_STATE (-3) /* object used out of order (e.g. backward before forward) */

typedef struct psg_ctx psg_ctx;
typedef struct psg_pn2_model psg_pn2_model;
typedef struct psg_pn2_ws psg_pn2_ws;
typedef void *psg_stream; /* hipStream_t */

const char *psg_last_error(void);
const char *psg_version(void);
/* Environment switches (all optional; they select between tested code paths or turn on diagnosis output).  Writes
 * "NAME=value:kind;..." for every switch libpsg knows that is SET in this process's environment ("" when none) and
 * returns their number (< 0: error).  kind: 'p' = another tested path with the same results (to rounding where a sum
 * changes order), 'd' = diagnosis output only, 'r' = changes results - timing bisection switches that exist only in
 * libraries built with -DPSG_DIAG_BUILD (psg_diag_build() == 1; the default library has none, and bench.py refuses to
 * run on such a library).  The reference has no counterpart: it is a measurement-hygiene entry point. */
int psg_env_switches(char *buf, int cap);
int psg_diag_build(void);

/* One context per (process, device). */
int psg_ctx_create(int device, psg_ctx **out);
int psg_ctx_destroy(psg_ctx *ctx);

/* ------------------------------------------------------------------------------------------
 * Geometry unit ops, batched over P independent problems.  Problem p reads cloud (p % n_clouds).
 * ------------------------------------------------------------------------------------------ */

/* square_distance, PointNet/models/pointnet_util.py:19-40: out[b][i][j] = ((-2*dot) + |src_i|^2) + |dst_j|^2
 * with the reference's fp32 evaluation order (bit-exact).  src [B][N][3], dst [B][M][3], out [B][N][M]. */
int psg_square_distance(psg_ctx *ctx, const float *src, const float *dst, int B, int N, int M, float *out,
                        psg_stream stream);

/* farthest_point_sample, PointNet/models/pointnet_util.py:63-84.
 * xyz [n_clouds][N][3]; start [P] = the torch.randint draw of :75; out_idx [P][S].
 * Lowest index wins distance ties (torch.max CPU behaviour).  N <= 8192. */
int psg_fps(psg_ctx *ctx, const float *xyz, int n_clouds, int P, int N, int S, const int32_t *start,
            int32_t *out_idx, psg_stream stream);

/* index_points, pointnet_util.py:43-60: out[p][s][:] = points[p % n_clouds][idx[p][s]][:]. */
int psg_gather_points(psg_ctx *ctx, const float *points, int n_clouds, int P, int N, int C, const int32_t *idx,
                      int S, float *out, psg_stream stream);

/* query_ball_point, pointnet_util.py:87-107.  xyz [n_clouds][N][3]; new_xyz [P][S][3];
 * r2 = float32(radius**2); out_idx [P][S][K]: first K indices j (ascending) with
 * !(square_distance(new_xyz, xyz) > r2), padded with the first hit.  N <= 8192. */
int psg_ball_query(psg_ctx *ctx, const float *xyz, int n_clouds, const float *new_xyz, int P, int N, int S,
                   float r2, int K, int32_t *out_idx, psg_stream stream);

/* 3-NN + inverse-distance weights, pointnet_util.py:301-307.  xyz1 [n_clouds1][N][3] (fine),
 * xyz2 [P][S][3] (coarse); out_idx/out_w [P][N][3].  Stable ascending order.  3 <= S <= 8192. */
int psg_three_nn(psg_ctx *ctx, const float *xyz1, int n_clouds1, const float *xyz2, int P, int N, int S,
                 int32_t *out_idx, float *out_w, psg_stream stream);

/* ------------------------------------------------------------------------------------------
 * PointNet++ SSG sem-seg network (get_model, PointNet/models/pointnet2_sem_seg.py:6-40), eval mode.
 * ------------------------------------------------------------------------------------------ */

#define PSG_PN2_NUM_LAYERS 23 /* sa1..sa4 (3 each), fp4, fp3, fp2 (2 each), fp1 (3), conv1, conv2 */
#define PSG_PN2_NUM_CLASSES 13
#define PSG_PN2_IN_CHANNELS 9
/* the two PointNet++ sem-seg architectures of the reference */
#define PSG_PN2_ARCH_SSG 0        /* PointNet/models/pointnet2_sem_seg.py: single-scale grouping, 23 conv layers */
#define PSG_PN2_ARCH_MSG 1        /* PointNet/models/pointnet2_sem_seg_msg.py: two radii per SA level, 35 conv layers */
#define PSG_PN2_MSG_NUM_LAYERS 35

/* Build the device-resident, MFMA-packed weights.  `weights[i]` / `biases[i]` are HOST pointers to the
 * row-major [cout][cin] weight and [cout] bias of layer i with the eval-mode BatchNorm already
 * folded in, in the order sa1.0-2, sa2.0-2, sa3.0-2, sa4.0-2, fp4.0-1, fp3.0-1, fp2.0-1, fp1.0-2,
 * conv1(+bn1), conv2.  Layer shapes are fixed by the architecture (pointnet2_sem_seg.py:9-19).
 * Blocking (uploads synchronously); not for use inside a timed loop. */
int psg_pn2_model_create(psg_ctx *ctx, const float *const *weights, const float *const *biases,
                         psg_pn2_model **out);
/* Same for either architecture.  PSG_PN2_ARCH_MSG (replaces pointnet2_sem_seg_msg.py:7-21 + pointnet_util.py:210-227)
 * takes its n_layers = 35 layers in state_dict order: sa1 scale 0 layers 0-2, sa1 scale 1 layers 0-2, sa2 ..., sa4
 * (conv_blocks.i.j + bn_blocks.i.j), fp4.0-1, fp3.0-1, fp2.0-1, fp1.0-2, conv1(+bn1), conv2. */
int psg_pn2_model_create_arch(psg_ctx *ctx, int arch, const float *const *weights, const float *const *biases,
                              int n_layers, psg_pn2_model **out);
int psg_pn2_model_destroy(psg_pn2_model *model);

/* Workspace: geometry plan for up to `max_forwards` forwards of a batch of `batch` rooms of
 * `n_point` points, plus activations / ReLU masks / gradient buffers for ONE forward in flight. */
int psg_pn2_ws_create(psg_ctx *ctx, int batch, int n_point, int max_forwards, psg_pn2_ws **out);
/* Workspace for a model of architecture `arch`; forward / backward / nb_attack refuse a mismatched pair. */
int psg_pn2_ws_create_arch(psg_ctx *ctx, int arch, int batch, int n_point, int max_forwards, psg_pn2_ws **out);
int psg_pn2_ws_destroy(psg_pn2_ws *ws);
size_t psg_pn2_ws_bytes(const psg_pn2_ws *ws);

/* Per-launch timing with HIP events recorded on the launch stream (bench.py's roofline figure).
 * While enabled every kernel launch of plan_build / forward / backward / nb_attack on this workspace is
 * bracketed by two events.  prof_read (blocking: waits for the events) sums the elapsed ms and counts per
 * kernel tag: 0-3 sa1-4 fwd, 4-7 fp1-4 fwd (4 = fp1+head), 8-11 fp1-4 bwd, 12-15 sa1-4 bwd, 16 fps,
 * 17 ball query, 18 three_nn, 19 gather, 20 ce grad, 21 pgd step, 22 gradient memset.  n_tags >= 23. */
int psg_pn2_prof_enable(psg_pn2_ws *ws, int on);
int psg_pn2_prof_read(psg_pn2_ws *ws, int n_tags, double *total_ms, int *counts);

/* Diagnostics (blocking): copies the in-kernel clock stamps written when the environment variable
 * PSG_DIAG has bit 256 set -- per workgroup of the fp1+head forward kernel {s_memtime, s_memrealtime} at
 * entry and exit -- used to report the shader clock the chip holds under this load. */
int psg_pn2_debug_read(psg_pn2_ws *ws, unsigned long long *host_out, int n_words);

/* Geometry for `n_forward` forwards at once (sample_and_group's FPS + ball query of the four SA
 * levels, and the 3-NN tables of the four FP levels; pointnet_util.py:110-143, :301-307).
 * x0 [batch][n_point][9] point-major rooms (only channels 0:3 are read);
 * starts [n_forward][4][batch] = the torch.randint draws in reference call order.
 * Geometry depends on xyz and the draws only, never on colour, so an attack builds it once for
 * all of its iterations (n_forward*batch independent problems fill the chip). */
int psg_pn2_plan_build(psg_pn2_ws *ws, const float *x0, const int32_t *starts, int n_forward, psg_stream stream);

/* Read-back of plan slices for parity tests (device pointers into the workspace).
 * what: 0 fps idx [S_l], 1 group idx [S_l][K] (MSG: scale 0, K = 16), 2 nn idx [N_l][3], 3 nn weights [N_l][3],
 * 4 xyz of level+1, 5 group idx of MSG scale 1 [S_l][32] (null for SSG) */
const void *psg_pn2_plan_ptr(const psg_pn2_ws *ws, int what, int level, int forward, int room);

/* get_model.forward (pointnet2_sem_seg.py:22-40) with the geometry of plan slot `forward`.
 * x0 [batch][n_point][9]; logp_out [batch][n_point][13] = log_softmax; l4_out (nullable)
 * [batch][16][512] (MSG: [batch][16][1024]) = l4_points point-major.  Leaves the ReLU / arg-max masks in the
 * workspace.  For PSG_PN2_ARCH_MSG this is pointnet2_sem_seg_msg.py:23-42. */
int psg_pn2_forward(psg_pn2_model *model, psg_pn2_ws *ws, int forward, const float *x0, float *logp_out,
                    float *l4_out, psg_stream stream);

/* Input-gradient backward of the forward that last ran in `ws` (what autograd derives for a leaf on
 * the input features): dlogp [batch][n_point][13] = d loss / d log-probs;
 * dx0_out [batch][n_point][9] = d loss / d input features (geometry treated as constant: channels
 * 3:9 are exact, channels 0:3 exclude the paths through relative coordinates and 3-NN weights,
 * which no attack uses).  No weight gradients (the attack path never needs them). */
int psg_pn2_backward(psg_pn2_model *model, psg_pn2_ws *ws, int forward, const float *dlogp, float *dx0_out,
                     psg_stream stream);

/* The per-iteration launches of psg_pn2_nb_attack one at a time (round 6), so that a caller - the teacher-forced parity
 * tests - can drive exactly the kernels the fused attack loop runs between nontarget.py:29 and :39 / target.py:31 and :43:
 *   psg_pn2_forward_lean         psg_pn2_forward without the module outputs only psg_pn2_activation reads (no l4_out);
 *   psg_pn2_backward_colour      the input gradient on the colour channels only: dx0_out[.][3..5] are written, the other six
 *                                channels of a row are left untouched (level 0: compact rows, the first layer's three colour
 *                                columns on the vector pipe, one gather thread per point);
 *   psg_pn2_backward_colour_pgd  the same backward with the update of nontarget.py:37-39 / target.py:41-43 applied by the
 *                                gradient's last gather: x [batch][n_point][9] is updated in place on channels 3..5 exactly as
 *                                psg_pgd_step would (ori [batch][n_point][3], mask nullable [n_point], dir = +1 ascent / -1
 *                                descent, last = the reference's un-projected final step); no gradient is written. */
int psg_pn2_forward_lean(psg_pn2_model *model, psg_pn2_ws *ws, int forward, const float *x0, float *logp_out,
                         psg_stream stream);
int psg_pn2_backward_colour(psg_pn2_model *model, psg_pn2_ws *ws, int forward, const float *dlogp, float *dx0_out,
                            psg_stream stream);
int psg_pn2_backward_colour_pgd(psg_pn2_model *model, psg_pn2_ws *ws, int forward, const float *dlogp, float *x,
                                const float *ori, const uint8_t *mask, float alpha, float eps, float dir, int last,
                                psg_stream stream);

/* Per-layer activations of the last forward, for parity tests: which 0..3 = l1..l4 points (sa1..sa4
 * outputs), 4..6 = fp4, fp3, fp2 outputs.  Returns a device pointer [batch][points][channels]. */
const float *psg_pn2_activation_ptr(const psg_pn2_ws *ws, int which);
int psg_pn2_activation_channels(const psg_pn2_ws *ws, int which);   /* channels of that tensor (architecture dependent) */

/* ------------------------------------------------------------------------------------------
 * Attack-loop arithmetic (PointNet/attacks/torchattacks/attacks/nontarget.py, target.py).
 * ------------------------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------------------------
 * Per-operator entry points: what the STAND-ALONE forwards of the reference's public modules run on
 * (inside get_model / DenseDeepGCN the same arithmetic is one fused kernel per module).  Point-major rows, device
 * pointers, stream-ordered, no allocation; BatchNorm (eval) is folded into w / bias by the caller.  Input gradients only.
 * ------------------------------------------------------------------------------------------ */
/* sample_and_group's row assembly, pointnet_util.py:126-140: rows[(b,s,k)] = [xyz[g] - new_xyz[s], feat[g]] with
 * g = gidx[b][s][k] (feat_first = 1: [feat, rel_xyz], the MSG order :251-254).  rows_out [B*S*K][3+D]. */
int psg_group_rows(const float *xyz, const float *feat, const float *new_xyz, const int32_t *gidx, int B, int N, int S,
                   int K, int D, int feat_first, float *rows_out, psg_stream stream);
/* its transpose w.r.t. the features (index_points backward, :136): dfeat [B][N][D], zeroed here. */
int psg_group_rows_bwd(const float *drows, const int32_t *gidx, int B, int N, int S, int K, int D, int feat_first,
                       float *dfeat, psg_stream stream);
/* one shared 1x1 convolution over rows (Conv -> BN(eval) -> ReLU, pointnet_util.py:200-203, 317-319):
 * out = [relu](in . w^T + bias) [* scale + shift], w [M][K]; scale / shift (or NULL): BatchNorm AFTER the ReLU, ResGCN's
 * BasicConv order (torch_nn.py:55-75); mask_out: ReLU bits [rows][ceil(M/32)] words or NULL. */
int psg_pw_mlp_fwd(const float *in, int ld_in, int rows, int K, const float *w, const float *bias, int relu, int M,
                   float *out, int ld_out, uint32_t *mask_out, const float *scale, const float *shift, psg_stream stream);
/* its input gradient: din = (dout . w) * [ReLU bits of the layer below, or NULL]; wT [K][M] = w transposed. */
int psg_pw_mlp_bwd(const float *dout, int ld_dout, int rows, int M, const float *wT, const uint32_t *mask_below, int K,
                   float *din, int ld_din, psg_stream stream);
/* backward through a layer's own ReLU (and the BatchNorm scale after it, or NULL): g[row][c] = bit ? g * scale[c] : 0. */
int psg_apply_relu_bits(float *g, int ld, const uint32_t *bits, const float *scale, int rows, int M, psg_stream stream);
/* PointNetSetAbstraction.forward after grouping, pointnet_util.py:200-205: n_layers shared layers over the grouped rows
 * [n_groups*K][cin], max over the K samples.  scratch_a/b: [rows][max width]; masks[l]: ReLU bits of layer l (or NULL array);
 * out [n_groups][widths[n-1]], arg: winning sample (first on ties, like torch.max). */
int psg_sa_mlp_max_fwd(const float *rows_in, int n_groups, int K, int cin, int n_layers, const int *widths,
                       const float *const *w, const float *const *bias, float *scratch_a, float *scratch_b,
                       uint32_t *const *masks, float *out, uint8_t *arg, psg_stream stream);
int psg_sa_mlp_max_bwd(const float *dout, const uint8_t *arg, int n_groups, int K, int cin, int n_layers,
                       const int *widths, const float *const *wT, const uint32_t *const *masks, float *scratch_a,
                       float *scratch_b, float *drows_in, psg_stream stream);
/* PointNetFeaturePropagation's interpolation + concat, pointnet_util.py:301-314: out [B][N][D1+D2] =
 * [feat1 (NULL when D1 = 0), sum_j w_j feat2[idx_j]]; idx / w from psg_three_nn. */
int psg_three_interp_fwd(const float *feat2, const int32_t *idx, const float *w, const float *feat1, int B, int N, int S,
                         int D1, int D2, float *out, psg_stream stream);
/* transpose: dfeat2 [B][S][D2] (zeroed here) += w_j * dout[b][n][col0 + c], dout rows of ld floats. */
int psg_three_interp_bwd(const float *dout, int ld, int col0, const int32_t *idx, const float *w, int B, int N, int S,
                         int D2, float *dfeat2, psg_stream stream);
/* pairwise_distance, ResGCN/gcn_lib/dense/torch_edge.py:32-42: x [B][N][C] -> out [B][N][N], the reference's fp32 order;
 * sq: scratch [B*N]. */
int psg_gcn_pairwise_distance(const float *x, int B, int N, int C, float *sq, float *out, psg_stream stream);
/* torch.max_pool2d(fusion, [N, 1]) of DenseDeepGCN.forward (ResGCN/sem_seg_dense/architecture.py:64) on point-major rows
 * x [B][N][C] (C a multiple of 64): out_max [B][C] = maximum over the N points, out_arg [B][C] = the row that holds it
 * (lowest row on equal values); scratch: [B*C] 8-byte words (cleared here).  SURVEY 8(b)'s `psg_global_max`. */
int psg_global_max(const float *x, int B, int N, int C, unsigned long long *scratch, float *out_max, int32_t *out_arg,
                   psg_stream stream);
/* EdgeConv2d.forward, torch_vertex.py:31-35 (BasicConv = Conv -> ReLU -> BatchNorm, torch_nn.py:55-75), 16 neighbours,
 * 64 output channels: y_i = max_k(scale * relu(W.[x_i, x_j - x_i] + b) + shift).  x [R][ld_x], nbr [R][16] room-local,
 * R = rooms * N; wcat [128][C] = [W1 - W2 ; W2], bcat [128] = [b, 0]; pq scratch [R][128]; arg [R][64]. */
int psg_edgeconv_fwd(const float *x, int ld_x, int R, int N, int C, const int32_t *nbr, const float *wcat,
                     const float *bcat, const float *scale, const float *shift, float *pq, float *out, int ld_out,
                     uint8_t *arg, psg_stream stream);
/* MRConv2d's gather, torch_vertex.py:16-19: cat [R][2C] = [x_i, max_k (x_j - x_i)], arg [R][C] = winning neighbour; the
 * BasicConv after it is psg_pw_mlp_fwd with scale / shift.  _bwd: dx [R][ld_dx] = dcat_x - dcat_m, + dcat_m at the winner. */
int psg_mrconv_gather_fwd(const float *x, int ld_x, int R, int N, int C, const int32_t *nbr, float *cat, uint8_t *arg,
                          psg_stream stream);
int psg_mrconv_gather_bwd(const float *dcat, int R, int N, int C, const int32_t *nbr, const uint8_t *arg, float *dx, int ld_dx,
                          psg_stream stream);
/* its input gradient: dx [R][ld_dx] (C columns); wcat_t [C][128]; dpq: scratch [R][128], holds [dP | dQ] when the
 * call returns (every element is written by the call: nothing needs zeroing; the result is bit-reproducible). */
int psg_edgeconv_bwd(const float *dy, int ld_dy, int R, int N, int C, const int32_t *nbr, const uint8_t *arg,
                     const float *scale, const float *wcat_t, float *dpq, float *dx, int ld_dx, psg_stream stream);

/* [B][C][N] channel-major (reference layout) <-> [B][N][C] point-major. */
int psg_to_point_major(const float *src_cn, int B, int C, int N, float *dst_nc, psg_stream stream);
int psg_to_channel_major(const float *src_nc, int B, int C, int N, float *dst_cn, psg_stream stream);

/* d/dlogp of scale * sum_i CE(logp_i, y_i) with CE applied ON TOP of the log-probs (a second
 * log_softmax), nontarget.py:26,34 / target.py:27,39.  labels [rows] int32, or a single class
 * when labels == NULL (`target`).  rows_active: only the first rows_active rows get a gradient
 * (target.py:36 uses batch row 0 only); the rest are zeroed.  cost_out (nullable): 1 float, += cost. */
int psg_ce_logp_grad(const float *logp, const int32_t *labels, int target, int rows, int rows_active, int n_cls,
                     float scale, float *dlogp_out, float *cost_out, psg_stream stream);

/* One NB / tar_NB update on the colour channels 3:6 of point-major rooms x [B][N][9], in place:
 *   stepped = x + dir*alpha*sign(g);  eta = clamp(stepped - ori, -eps, eps);
 *   x = last ? stepped : clamp(ori + eta, 0, 1)          (nontarget.py:37-39, target.py:41-43)
 * `last` reproduces the reference returning the UN-projected final step.
 * grad [B][N][9] (channels 3:6 read), ori [B][N][3], mask (nullable) [N] uint8 shared by all rooms. */
int psg_pgd_step(float *x, const float *grad, const float *ori, const uint8_t *mask, int B, int N, float alpha,
                 float eps, float dir, int last, psg_stream stream);

/* Fused NB_attack / tar_NB_attack on a batch (nontarget.py:18-42, target.py:18-45):
 * plan_build for `iters` forwards, then iters x (forward, CE grad, backward, pgd_step), all
 * stream-ordered on `stream` with no host synchronisation.
 * images/adv_out [B][9][N] channel-major (reference layout); labels [B][N] int32 (ignored when
 * targeted); starts [iters][4][B] int32 (device); mask (nullable) [N] uint8.
 * targeted = 0: ascent on CE_sum(all rooms)/N.  targeted = 1: descent on CE_mean(room 0, target). */
int psg_pn2_nb_attack(psg_pn2_model *model, psg_pn2_ws *ws, const float *images, const int32_t *labels,
                      const int32_t *starts, const uint8_t *mask, float eps, float alpha, int iters, int targeted,
                      int target, float *adv_out, psg_stream stream);

/* ---- NU (Adam in tanh space) attack arithmetic: nontarget.py:52-135, target.py:62-175 -------------
 * w / m / v are [B][N][3] fp32 (the attack variable and its Adam moments); colours live in channels 3:6
 * of the point-major rooms x0 [B][N][9]; `mask` (nullable, [N] uint8) restricts every op to masked points. */

/* w = 0.5*log((1+x)/(1-x)), x = 2c-1 (inverse_tanh_space, nontarget.py:110-116). */
int psg_nu_inverse_tanh(const float *x0, int B, int N, float *w_out, psg_stream stream);

/* colour = 1/2*(tanh(w)+1) written into x0 (tanh_space, nontarget.py:107-108). */
int psg_nu_tanh_color(const float *w, const uint8_t *mask, int B, int N, float *x0, psg_stream stream);

/* f-loss (nontarget.py:119-128 / target.py:148-168): per point clamp(tsign*(p_y - max_{k!=y} p_k), min=-kappa)
 * on p = softmax(log-probs); y = labels[row] or `target` when labels == NULL.  Writes d(sum f)/d(logp),
 * adds the sum to *f_sum (nullable) and the arg-max class to pred_out (nullable). */
int psg_nu_f_loss_grad(const float *logp, const int32_t *labels, int target, int rows, int n_cls, float kappa,
                       float tsign, float *dlogp_out, float *f_sum, int32_t *pred_out, psg_stream stream);

/* f-loss of the ResGCN NU attacks on raw logits [rows][n_cls] (ResGCN/.../attacks/colper.py:108-113,
 * tcolper.py:145-163); the reference's one-hot masking makes a 0 take part in each max.  mode 0 = NU_attack.f over
 * all rows; mode 1 = tar_NU non_f, mode 2 = tar_NU tar_f (class `target`), both over batch row 0 under `mask` only.
 * Writes scale * d(sum f)/d(logits), adds sum f to *f_sum (nullable), arg-max class to pred_out (nullable). */
int psg_gcn_f_loss_grad(const float *logits, const int32_t *labels, int target, const uint8_t *mask, int mode, int rows,
                        int n_point, int n_cls, float kappa, float tsign, float scale, float *dlogits_out, float *f_sum,
                        int32_t *pred_out, psg_stream stream);

/* Smooth loss (nontarget.py:131-135): for each of the N adversarial colours (rows of `adv_color`, stride
 * in floats) the nb smallest Euclidean distances to the N reference colours; adds their total to *dist_sum
 * and writes d(total)/d(adv colour) to grad_out [N][3].  Distances are evaluated directly (the reference's
 * cdist's matmul-expansion distance is reproduced).  Passing ref_color == adv_color selects the ResGCN
 * variants' smooth(adv, adv) (colper.py:115-120): the gradient then flows through both arguments.  nb <= 16, N <= 8192. */
int psg_smooth_knn(const float *adv_color, int adv_stride, const float *ref_color, int ref_stride, int N, int nb,
                   float *dist_sum, float *grad_out, psg_stream stream);

/* One optimiser step: g = dx0[colour] + 2*c_l2*(colour - ori) (+ c_smooth*smooth_grad on room 0), chained
 * through tanh_space, then torch.optim.Adam's single-tensor update (betas, eps, bias correction for `step`
 * >= 1) on w/m/v in place.  Adds sum((colour-ori)^2) to *l2_sum (nullable). */
int psg_nu_adam_step(float *w, float *m, float *v, const uint8_t *mask, const float *dx0, const float *x0,
                     const float *ori, const float *smooth_grad, float c_smooth, float c_l2, float lr, float beta1,
                     float beta2, float eps, int step, int B, int N, float *l2_sum, psg_stream stream);

/* B INDEPENDENT one-room attacks advanced in lockstep: the reference's NU / tar_NU attack called once per room
 * (target.py:62-133 and nontarget.py:52-105 at B = 1; BASELINE configs[2] applied per room), served by one launch per
 * operation instead of B.  Same arithmetic per room as the entry points above; what differs is the bookkeeping: the mask
 * is [B][N] (every room its own mask row), the smoothness term is evaluated for every room (grad_out [B][N][3]), the
 * loss sums are per room ([B] floats) and rooms whose loop has ended (room_active[b] == 0, nullable = all active) are
 * left untouched by the optimiser step.  N must be a multiple of 64 for the f-loss sums. */
int psg_nu_tanh_color_rooms(const float *w, const uint8_t *mask_rooms, int B, int N, float *x0, psg_stream stream);
int psg_nu_f_loss_grad_rooms(const float *logp, const int32_t *labels, int target, int B, int N, int n_cls, float kappa,
                             float tsign, float *dlogp_out, float *f_sum_rooms, int32_t *pred_out, psg_stream stream);
/* nn_state (nullable): int32 [B][N][nb], the neighbour indices of every colour; written by every call, and with
 * have_prev != 0 read first: the largest current distance to last call's neighbours is a rigorous upper bound of the
 * nb-th smallest distance, so the scan starts from it (same result; the optimiser moves colours a little per step). */
int psg_smooth_knn_rooms(const float *adv_color, int adv_stride, size_t adv_room_stride, const float *ref_color,
                         int ref_stride, size_t ref_room_stride, int B, int N, int nb, float *dist_sum_rooms,
                         float *grad_out, int32_t *nn_state, int have_prev, psg_stream stream);
int psg_nu_adam_step_rooms(float *w, float *m, float *v, const uint8_t *mask_rooms, const float *dx0, const float *x0,
                           const float *ori, const float *smooth_grad_rooms, float c_smooth, float c_l2, float lr,
                           float beta1, float beta2, float eps, int step, int B, int N, const uint8_t *room_active,
                           float *l2_sum_rooms, psg_stream stream);

/* Per-step statistics and exit latch of the NU attacks: what the reference evaluates on the host after every optimiser
 * step - nontarget.py:87,95-96 (`correct / 4096 < 1 / 13`), target.py:105-121 (`target_acc` > 0.9, or < 1 / 13 for the
 * untargeted goal) - for G attacks of `rows` batch rows each (rows = B: one call on a batch; rows = 1: rooms in lockstep).
 * mode 0 = NU_attack, 1 = tar_NU_attack without a target (hits = correct points on the mask), 2 = with a target (hits =
 * points on the mask predicted as `target`).  scal [3][G]: the step's f / Smooth / L2 sums, moved into hist_step [5][G]
 * = {n_correct, n_hits, f, Smooth, L2} and zeroed.  The first step whose test fires copies the group's x0 rows
 * [rows][N][9] into out_cn [rows][9][N], stores the step in exit_step [G] (-1 before) and clears active [G]. */
int psg_nu_step_latch(const int32_t *pred, const int32_t *labels, int target, const uint8_t *mask_groups,
                      const int32_t *n_mask, int G, int rows, int N, int mode, float *scal, float *hist_step,
                      const float *x0, float *out_cn, uint8_t *active, int32_t *exit_step, int step, psg_stream stream);

/* A WINDOW of consecutive optimiser steps of NU_attack / tar_NU_attack on a PointNet++ network, enqueued by ONE call:
 * per step exactly the sequence the host loop of the reference runs (nontarget.py:77-96, target.py:93-121) -
 * psg_nu_tanh_color(_rooms), psg_pn2_forward on plan slot slot0 + i, psg_nu_f_loss_grad(_rooms), psg_pn2_backward,
 * psg_smooth_knn_rooms, psg_nu_adam_step(_rooms), psg_nu_step_latch - with the same arguments those entry points take.
 * (G, rows) = (1, B): one attack on a batch of B rows; (R, 1): R one-room attacks in lockstep.  The reference's host work
 * (restart test, learning-rate halving, exit) sits between windows; inside one nothing returns to the host. */
typedef struct psg_nu_window_args {
    psg_pn2_model *model;
    psg_pn2_ws *ws;
    int slot0, step0, n_steps;   /* steps step0 .. step0 + n_steps - 1 use plan slots slot0 .. */
    int G, rows, N, mode;        /* mode: see psg_nu_step_latch */
    int use_target, target;      /* the f-loss takes `target` instead of the labels */
    int neighbour, warm_first;   /* Smooth term: neighbour count; 1 = nn_state holds the previous step's lists */
    int adam_t0;                 /* optimiser steps taken since the optimiser was created */
    float kappa, tsign, c_smooth, c_l2, lr, beta1, beta2, eps;
    float *w, *m, *v;            /* [G*rows][N][3] */
    const uint8_t *mask;         /* [G][N] (rows = 1) or [N] (G = 1), nullable for NU_attack */
    const int32_t *n_mask;       /* [G], modes 1 and 2 */
    float *x0;                   /* [G*rows][N][9] */
    const float *ori;            /* [G*rows][N][3] */
    const int32_t *labels;       /* [G*rows][N] */
    float *logp, *dlogp, *dx0;   /* [G*rows][N][13], [..][13], [..][9] scratch */
    float *sgrad;                /* [G][N][3] */
    int32_t *pred;               /* [G*rows][N] */
    float *scal;                 /* [3][G], zero before the first window */
    int32_t *nn_state;           /* [G][N][neighbour] */
    float *hist;                 /* history rows [5][G] of the window's steps, row of step0 first */
    float *out;                  /* [G*rows][9][N] */
    uint8_t *active;             /* [G] */
    int32_t *exit_step;          /* [G] */
} psg_nu_window_args;
/* `graph` (nullable): a handle that lets windows of one SHAPE (same buffers, plan slots and step count; only step0, adam_t0
 * and lr differ) be replayed as a hipGraph - the first such window runs eagerly, the second is captured, later ones are
 * replayed with their step constants read from a device row.  Same results as the eager sequence. */
#define PSG_NU_GRAPH_MAX_STEPS 16
typedef struct psg_nu_graph psg_nu_graph;
int psg_nu_graph_create(psg_nu_graph **out);
int psg_nu_graph_destroy(psg_nu_graph *graph);
int psg_pn2_nu_window(const psg_nu_window_args *args, psg_nu_graph *graph, psg_stream stream);
/* Bookkeeping of a handle: out4 = {captures tried, captures failed, windows replayed as a graph, windows run eagerly}.  A
 * failed capture (the legacy default stream refuses capture; another API call can invalidate one) costs speed, not
 * correctness - the window runs eagerly - so it is counted instead of raised, and not retried for the same shape. */
int psg_nu_graph_stats(const psg_nu_graph *graph, long long *out4);
/* The same four counters summed over every replayed loop of the process (psg_pn2_nu_window, psg_gcn_nb_attack,
 * psg_rla_bim_attack; for the last two the unit is an attack iteration). */
int psg_capture_stats(long long *out4);

/* Segmentation statistics of NB_nontarget_test_semseg.py:199-205: for every class l accumulates
 * seen[l] += #(gt==l), inter[l] += #(pred==l & gt==l), uni[l] += #(pred==l | gt==l) where
 * pred = argmax(logp) (first index on ties).  counters: int64 [3][n_cls] = seen, inter, uni.
 * pred_out (nullable) int32 [rows]. */
int psg_seg_stats(const float *logp, const int32_t *labels, int rows, int n_cls, long long *counters,
                  int32_t *pred_out, psg_stream stream);

/* ------------------------------------------------------------------------------------------
 * Whole-scene evaluation harness.
 * ------------------------------------------------------------------------------------------ */

/* add_vote (PointNet/NB_nontarget_test_semseg.py:55-62): for every block point r with weight[r] != 0 (weight
 * nullable = all), pool[point_idx[r]][pred[r]] += 1, pred = the given labels (int32 [rows]) or, when `pred` is
 * null, argmax(logp[r]) with the first index on ties (`seg_pred.max(2)[1]`).  pool: int32 [n_points][n_cls],
 * accumulated in place.  *bad_flag (int32, device) is OR-ed with 1 if an index is out of range. */
int psg_vote_add(const float *logp, const int32_t *pred, const int32_t *point_idx, const float *weight, int rows,
                 int n_cls, int n_points, int32_t *pool, int32_t *bad_flag, psg_stream stream);

/* Per-scene statistics of NB_nontarget_test_semseg.py:219-229: pred = np.argmax(pool, 1) (first index on ties),
 * counters int64 [3][n_cls] += seen, correct, union against `labels` (int32 [n_points]).  pred_out nullable. */
int psg_vote_stats(const int32_t *pool, const int32_t *labels, int n_points, int n_cls, long long *counters,
                   int32_t *pred_out, psg_stream stream);

/* torch.dist(a, b) (p = 2) of NB_nontarget_test_semseg.py:184: out[0] = sqrt(sum (a-b)^2), accumulated in double
 * in a fixed order.  scratch256: 256 doubles of device scratch. */
int psg_l2_dist(const float *a, const float *b, size_t n, double *scratch256, float *out, psg_stream stream);

/* ------------------------------------------------------------------------------------------
 * ResGCN-28 dense DeepGCN sem-seg network (ResGCN/sem_seg_dense/architecture.py:6-68), eval mode:
 * head EdgeConv(9->64) on the xyz kNN graph, n_blocks-1 residual dynamic EdgeConv blocks (feature-space
 * kNN, k = 16, dilation 1..n_blocks-1), fusion 1x1 conv + global max, prediction MLP.  BasicConv order is
 * Conv -> ReLU -> BatchNorm (ResGCN/gcn_lib/dense/torch_nn.py:55-66).
 * ------------------------------------------------------------------------------------------ */
typedef struct psg_gcn_model psg_gcn_model;
typedef struct psg_gcn_ws psg_gcn_ws;

/* tensors: HOST fp32 pointers in the order documented in csrc/psg_resgcn.hip (per EdgeConv: conv weight
 * [64][2C], bias, BN weight/bias/running_mean/running_var; fusion_block; prediction.0/.1 with BN; prediction.3).
 * n_tensors must equal 6*n_blocks + 20.  Blocking. */
int psg_gcn_model_create(psg_ctx *ctx, const float *const *tensors, int n_tensors, int n_blocks, psg_gcn_model **out);
/* The reference's configuration switches (ResGCN/sem_seg_dense/architecture.py:26-39 `opt.block`, gcn_lib/dense/
 * torch_vertex.py:44-49 `opt.conv`): block = ResDynBlock2d / PlainDynBlock2d / DenseDynBlock2d, conv = EdgeConv2d /
 * MRConv2d.  Same tensor order as psg_gcn_model_create with the reference's shapes for that configuration (dense:
 * conv weight of block e is [64][2*64e], fusion_block [1024][64 n(n+1)/2], prediction.0 [512][1024 + 64 n(n+1)/2]). */
#define PSG_GCN_BLOCK_RES 0
#define PSG_GCN_BLOCK_PLAIN 1
#define PSG_GCN_BLOCK_DENSE 2
#define PSG_GCN_CONV_EDGE 0
#define PSG_GCN_CONV_MR 1
int psg_gcn_model_create_cfg(psg_ctx *ctx, const float *const *tensors, int n_tensors, int n_blocks, int block, int conv,
                             psg_gcn_model **out);
int psg_gcn_model_destroy(psg_gcn_model *model);
int psg_gcn_ws_create(psg_ctx *ctx, int batch, int n_point, int n_blocks, psg_gcn_ws **out);
int psg_gcn_ws_create_cfg(psg_ctx *ctx, int batch, int n_point, int n_blocks, int block, int conv, psg_gcn_ws **out);
int psg_gcn_ws_destroy(psg_gcn_ws *ws);
size_t psg_gcn_ws_bytes(const psg_gcn_ws *ws);
/* Measurement only (bench.py `roofline`): per-launch HIP-event timing on the launch stream, like psg_pn2_prof_*.  Tags:
 * 0 fused feature-space kNN kernel (torch_edge.py:32-59), 1 other kNN launches, 2 per-vertex EdgeConv GEMM, 3 edge max,
 * 4 fusion + prediction, 5 backward; flops = algorithmic FLOPs of the tagged launches.  While enabled the attack loop does
 * not use its hipGraph. */
int psg_gcn_prof_enable(psg_gcn_ws *ws, int on);
int psg_gcn_prof_read(psg_gcn_ws *ws, int n_tags, double *total_ms, int *counts, double *flops);

/* DenseDilatedKnnGraph (torch_edge.py:45-79) on point-major features x [batch][n_point][C]: neighbours at
 * ranks 0, d, 2d, ... of the ascending pairwise_distance row, k = 16; out_idx [batch][n_point][16].
 * Distances follow the reference's fp32 order (ascending-k fmaf dot, (|xi|^2 + -2 xi.xj) + |xj|^2);
 * equal distances resolve to the lowest index (torch.topk leaves that order unspecified). */
int psg_gcn_knn(psg_gcn_ws *ws, const float *x, int C, int dilation, int32_t *out_idx, psg_stream stream);
/* Measurement only: counters of the feature-space kNN kernel since the last reset, for workspaces created while the
 * environment holds PSG_GCN_KNN_STATS=1 (all zero otherwise).  The kernel decides most of a distance row on a bf16 MFMA
 * approximation with a proven error bound and evaluates the reference's fp32 distance (torch_edge.py:41-43) only for the
 * candidates that can hold a wanted rank; tiles whose bound cannot be kept take the all-fp32 path inside the same launch.
 * host_out8: [0] 32-query tiles, [1] tiles that took the exact path, [2] rows ranked on the fast path, [3] exact distances
 * evaluated there, [4] row cuts, [5] entries held at the end of the stream, [6..7] 0.  Synchronises the device. */
int psg_gcn_knn_stats(psg_gcn_ws *ws, unsigned long long *host_out8, int reset);

/* DenseDeepGCN.forward (architecture.py:58-68): x0 [batch][n_point][9] point-major -> logits [batch][n_point][13]. */
int psg_gcn_forward(psg_gcn_model *model, psg_gcn_ws *ws, const float *x0, float *logits_out, psg_stream stream);

/* Input-gradient backward of the resident forward (kNN graphs are constants, as under torch.no_grad in the
 * reference): dlogits [batch][n_point][13] -> dx0_out [batch][n_point][9]. */
int psg_gcn_backward(psg_gcn_model *model, psg_gcn_ws *ws, const float *dlogits, float *dx0_out, psg_stream stream);

/* colper.NB_attack (ResGCN/sem_seg_dense/attacks/torchattacks/attacks/colper.py:17-39): CrossEntropyLoss (mean)
 * on logits, sign ascent on colour, L-inf projection; images/adv_out [batch][9][n_point]. */
int psg_gcn_nb_attack(psg_gcn_model *model, psg_gcn_ws *ws, const float *images, const int32_t *labels, float eps,
                      float alpha, int iters, float *adv_out, psg_stream stream);

/* Teacher forcing for parity tests: nbr (device, [n_blocks][batch][n_point][16]) replaces the kNN graphs of
 * every following forward; NULL restores the dynamic graphs.  (Feature-space kNN has near-ties that no two
 * fp32 pipelines break alike, so value parity is checked on identical graphs and graph parity by overlap.) */
int psg_gcn_set_graphs(psg_gcn_ws *ws, const int32_t *nbr, psg_stream stream);

/* parity-test read-back: neighbour table of EdgeConv `block` [batch][n_point][16]; block outputs [batch][n_point][64*n_blocks] */
const int32_t *psg_gcn_edge_ptr(const psg_gcn_ws *ws, int block);
const float *psg_gcn_feats_ptr(const psg_gcn_ws *ws);

/* ------------------------------------------------------------------------------------------
 * RandLA-Net input pipeline (first piece): exact k-nearest neighbours of 3-D points.
 * Replaces DataProcessing.knn_search (RandLA-Net/helper_tool.py:158-167) = nearest_neighbors.knn_batch(support,
 * query, k, omp=True) (utils/nearest_neighbors/knn_.cxx:103-134, nanoflann kd-tree + OpenMP on the host).
 * support [batch][n_support][3], query [batch][n_query][3] device fp32; out_idx [batch][n_query][k] int32: the k
 * support points nearest to each query in ascending (squared distance, index) order, 1 <= k <= 16.  Distances are
 * evaluated like nanoflann's L2 metric, so neighbour sets equal the reference's (order may differ among exact ties,
 * where nanoflann's order depends on its tree traversal).
 * ------------------------------------------------------------------------------------------ */
int psg_knn_points(psg_ctx *ctx, const float *support, const float *query, int batch, int n_support, int n_query, int k,
                   int32_t *out_idx, psg_stream stream);

/* ------------------------------------------------------------------------------------------
 * RandLA-Net: inference graph, colour gradient and BIM attack for ONE cloud per
 * workspace (ConfigS3DIS.val_batch_size = 1), 5 levels, k = 16, 13 classes.  Replaces Network.inference
 * (RandLA-Net/RandLANet.py:150-190 with :323-410) and the BIM loop of ares/ares/attack/bim.py:66-116,190-236.
 * PARITY UNPINNED: the reference is a TensorFlow-1 graph that cannot run in this environment; the checker is a
 * source-reading restatement (oracle/randla_net.py).
 * tensors: 6 HOST pointers per layer in forward order (pointsecguard_amd.synthetic.randla_layer_specs: fc0; per encoder
 * level mlp1, LFAmlp1, LFAatt_pooling_1fc, LFAatt_pooling_1mlp, LFAmlp2, LFAatt_pooling_2fc, LFAatt_pooling_2mlp, mlp2,
 * shortcut; decoder_0; Decoder_layer_0..4; fc1, fc2, fc): weight [cout][cin], bias or NULL, BatchNorm gamma, beta,
 * moving mean, moving variance or NULL (eps 1e-6, folded on the host).
 * ------------------------------------------------------------------------------------------ */
#define PSG_RLA_NUM_LAYERS 55
/* Possibility-based crop sampler of the RandLA-Net input pipeline (RandLA-Net/main_S3DIS.py:116-187, spatially_regular_gen):
 * one object per cloud, points [n][3] f32 and the float64 possibility of every point resident on the device.
 * argmin = np.argmin / np.min of the possibility (:134-137); query = the cloud's KDTree.query(pick, k)[1][0] (:151-156): the k
 * nearest points ascending by (float64 squared distance, index); update = possibility[idx] += (1 - d / max d)^2 with the
 * reference's float32 d (:163-166).  What the reference draws from numpy's generator stays on the host
 * (pointsecguard_amd/randla/sampler.py). */
typedef struct psg_rla_sampler psg_rla_sampler;
int psg_rla_sampler_create(psg_ctx *ctx, const float *points_host, const double *possibility_host, int n_points,
                           psg_rla_sampler **out);
int psg_rla_sampler_destroy(psg_rla_sampler *s);
int psg_rla_sampler_argmin(psg_rla_sampler *s, int *index_out, double *value_out, psg_stream stream);
int psg_rla_sampler_query(psg_rla_sampler *s, const double *pick_host3, int k, int32_t *out_idx, psg_stream stream);
int psg_rla_sampler_update(psg_rla_sampler *s, const int32_t *idx, int k, const double *pick_host3, float *scratch, psg_stream stream);
int psg_rla_sampler_possibility(psg_rla_sampler *s, double *host_out);

typedef struct psg_rla_model psg_rla_model;
typedef struct psg_rla_ws psg_rla_ws;
int psg_rla_model_create(psg_ctx *ctx, const float *const *tensors, int n_tensors, psg_rla_model **out);
int psg_rla_model_destroy(psg_rla_model *model);
int psg_rla_ws_create(psg_ctx *ctx, int n_points, psg_rla_ws **out);
/* A workspace for `batch` clouds of n_points each that are attacked together (the reference's val_batch_size is 1,
 * helper_tool.py:52, so this is a launch-coalescing device: the clouds stay independent, every neighbour / pooling /
 * interpolation index stays inside its cloud, and one launch of each kernel serves all of them).  Every [n_points][..]
 * argument of the calls below becomes [batch][n_points][..].  l_inf metric only (the l_2 step normalises per cloud). */
int psg_rla_ws_create_batch(psg_ctx *ctx, int n_points, int batch, psg_rla_ws **out);
int psg_rla_ws_destroy(psg_rla_ws *ws);
size_t psg_rla_ws_bytes(const psg_rla_ws *ws);
/* Measurement only: HIP-event timing of every GEMM launch (tag 0: the 1x1 convolutions and attention scores of
 * RandLANet.py:323-410) with its algorithmic FLOPs; while enabled the BIM loop does not use its hipGraph. */
int psg_rla_prof_enable(psg_rla_ws *ws, int on);
int psg_rla_prof_read(psg_rla_ws *ws, int n_tags, double *total_ms, int *counts, double *flops);
/* The same launches split by kernel (n_tags >= 4): 0 = the 64 x 64-tile row GEMM of the point-sized layers, 1 = 128 x 128
 * tiles, 2 = 256 x 64 tiles, 3 = the row-per-thread kernel of the 8-32 channel layers; bytes = algorithmic bytes of the
 * tagged launches (input rows, output rows, addend rows, weights, mask words): these kernels are bandwidth-bound. */
int psg_rla_prof_read_kernels(psg_rla_ws *ws, int n_tags, double *total_ms, int *counts, double *flops, double *bytes);
/* xyz [n_points][3] device: builds the index pyramid of main_S3DIS.py:198-207 (psg_knn_points) and the relative
 * position encodings.  Sub-sampling is the reference's: the first n / ratio points of each level. */
int psg_rla_set_cloud(psg_rla_ws *ws, const float *xyz, psg_stream stream);
/* what: 0 neighbour idx [n_l][16], 1 up-sampling idx [n_l]; device pointers into the workspace (tests) */
const int32_t *psg_rla_index_ptr(const psg_rla_ws *ws, int what, int level);
/* features [n_points][6] = (xyz, rgb) -> logits [n_points][13] */
int psg_rla_forward(psg_rla_model *model, psg_rla_ws *ws, const float *features, float *logits_out, psg_stream stream);
/* d loss / d logits -> d loss / d features [n_points][6] (through the feature path only: the relative-position branch
 * is constant for a colour attack); consumes the resident forward */
int psg_rla_backward(psg_rla_model *model, psg_rla_ws *ws, const float *dlogits, float *dfeatures_out, psg_stream stream);
/* the attack's loss (bim.py:110-116) and its gradient w.r.t. the logits; loss_out (nullable) device scalar */
int psg_rla_colper_grad(const float *logits, const int32_t *labels, int n, float *dlogits, float *loss_out, psg_stream stream);
/* The same with the targeted attacks' mask and sign (TBIM bim.py:393-397 / :350-351, tar_NUattack.py:105-110): ys = the
 * labels the hinge is taken against (the target class on the origin points), mask [n] bytes or NULL = which points
 * contribute, sign = -1 for goal 't' (grad = -grad). */
int psg_rla_colper_grad_masked(const float *logits, const int32_t *ys, const uint8_t *mask, float sign, int n, float *dlogits,
                               float *loss_out, psg_stream stream);
/* One BIM update (bim.py:84-98) of the colour half of feat [n][6] from dfeat [n][6]; ori [n][3] the clean colours; l_2:
 * norms = 2 floats and delta = [n][3] floats of scratch.  psg_rla_bim_attack's step as its own entry, for the attacks whose
 * loop reads an accuracy back every iteration (TBIM.batch_attack, bim.py:484-505). */
int psg_rla_bim_step(float *feat, const float *dfeat, const float *ori, int n, float eps, float alpha, int l2_metric, float *norms,
                     float *delta, psg_stream stream);
/* NUattack / tar_NUattack (ares/ares/attack/NUattack.py:12-74, tar_NUattack.py:12-84): adversarial colours of the tanh-space
 * variable d_ws into feat [n][6] columns 3..5 (mask: only those points move) and dist2[0] = |adv - x|_2^2; then one TF1-Adam
 * step (t = 1, 2, ..) on d_ws of loss = |adv - x|_2 + c * score from dfeat = d score / d features. */
int psg_rla_nu_color(const float *xs, const float *dws, const uint8_t *mask, int n, float *feat, float *dist2, psg_stream stream);
int psg_rla_nu_adam_step(const float *xs, float *dws, float *m, float *v, const uint8_t *mask, const float *feat,
                         const float *dfeat, const float *dist2, int n, float c, float lr, int t, psg_stream stream);
/* `iters` BIM updates (goal 'ut') of the colour half of `features`; l2_metric 0: l_inf, 1: l_2 (bim.py:84-98);
 * builds the cloud's geometry itself.  adv_features_out [n_points][6]. */
int psg_rla_bim_attack(psg_rla_model *model, psg_rla_ws *ws, const float *features, const int32_t *labels, float eps,
                       float alpha, int iters, int l2_metric, float *adv_features_out, psg_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* PSG_H */
