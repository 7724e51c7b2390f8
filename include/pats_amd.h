/*
 * pats_amd.h - C ABI of libpats_amd.so: the MI355X (gfx950) implementation of the PATS
 * patch-area optimal-transport hot path.
 *
 * Every entry point takes plain DEVICE pointers (fp32 / int64 / uint8, contiguous, row-major,
 * layouts exactly as the reference's tensors) plus sizes and a HIP stream; no torch types.  All
 * launches are asynchronous on `stream` (NULL = the default stream); nothing synchronises the
 * device unless stated.  Return value: PATS_OK or a PATS_ERR_* code, message in pats_last_error().
 * Inputs are never modified unless the name says `_inplace`.  Thread-safe (stateless).
 *
 * Citations are paths relative to the reference repository (zju3dv/pats).
 */
#ifndef PATS_AMD_H
#define PATS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pats_stream_t; /* hipStream_t */

enum {
    PATS_OK = 0,
    PATS_ERR_INVALID = 1,     /* bad shape / null pointer / workspace too small  (reference: TORCH_CHECK -> RuntimeError) */
    PATS_ERR_UNSUPPORTED = 2, /* shape outside what the kernels cover */
    PATS_ERR_LAUNCH = 3,      /* hip launch / runtime failure */
    PATS_ERR_NO_DEVICE = 4
};

/* Sinkhorn arithmetic.  LOG = max-subtracted log-sum-exp sweeps exactly as
 * models/modules.py:137-143.  KERNEL = the same fixed-point iteration carried in the linear
 * domain on K = exp(Z + u1 + v1) after one LOG warm-up sweep (row/col normalisation by
 * matrix-vector products, duals folded back into log space at the end), with an in-kernel guard
 * that re-runs a problem in LOG when its scaling vectors leave the safe fp32 range.  AUTO = KERNEL
 * where implemented, LOG otherwise. */
enum { PATS_SINKHORN_AUTO = 0, PATS_SINKHORN_LOG = 1, PATS_SINKHORN_KERNEL = 2 };

const char* pats_version(void);
/* The ABI this header describes.  It is bumped whenever an exported function changes its argument list (round 2 added
 * `row_nomatch` to pats_iterative_expand_f32 under the same symbol: a caller built against the older header would pass
 * its stream where the new pointer goes).  A C consumer checks `pats_abi_version() == PATS_ABI_VERSION` once after
 * loading the library; pats_amd/_lib.py does.  New arguments now come with new entry points instead. */
#define PATS_ABI_VERSION 6
int pats_abi_version(void);
const char* pats_last_error(void);
/* number of HIP devices visible (0 on a CPU-only box; never fails) */
int pats_device_count(void);
/* process-wide default for PATS_SINKHORN_AUTO (returns the previous value) */
int pats_set_sinkhorn_mode(int mode);
/* Fine level of pats_cost_ot_f32 / pats_cost_ot_flags_f32 (variant 2, 145 x 145): 0 (default) = the MFMA cost kernel, then the
 * register-block Sinkhorn kernel; 1 = ONE kernel that builds the score tile, turns it into the register blocks through LDS
 * and solves - the scores never reach HBM (second_layer.py:100-105 in one launch).  Same bits either way
 * (tests/test_gpu_parity.py); measured 5.69 against 5.62 ms per 20 224 problems.  Returns the
 * previous setting; PATS_FINE_FUSED=1 in the environment sets the initial one. */
int pats_set_fine_fused(int on);

/* Number of problems (on the current device, since the last reset) whose linear-domain solve left the
 * guard band and was re-solved with log-sum-exp sweeps.  Results are the same either way; a high rate
 * only costs time.  Synchronises the whole device (hipDeviceSynchronize) before the read and after the
 * reset, so launches on any stream are counted and none races with the reset.  No reference counterpart. */
int pats_sinkhorn_fallbacks(int64_t* count, int reset);

/* ---- a1-a3: cost build -------------------------------------------------------------------
 * out[b,i,j] = 0.1f * ( (sum_d d0[b,d,i] * d1[b,d,j]) / sqrtf(D) )
 * replaces  scores = einsum('bdn,bdm->bnm', mdesc0, mdesc1) / D**.5 ; 0.1 * scores
 *           models/first_layer.py:110-111,114  second_layer.py:100-101,104  third_layer.py:156-158
 * d0 [batch,D,n], d1 [batch,D,m] (channel-major), out [batch,n,m].  fp32 in, fp32 out; the contraction splits every
 * operand into an fp16 hi + lo pair (exact products on the fp16 matrix pipe, fp32 accumulation: on operands of magnitude
 * 0.004 .. 1023 at least as close to float64 as an fp32 fma chain; an operand below 0.004 has a subnormal lo half, which the
 * matrix pipe flushes, and is carried with an absolute error <= 2^-20, i.e. <= 1e-6 |y| per product) and redoes a tile with the fp32 MFMA when an operand exceeds +-1023. */
int pats_cost_f32(const float* d0, const float* d1, int64_t batch, int D, int n, int m, float* out,
                  pats_stream_t stream);

/* ---- a6: log_sinkhorn_iterations(Z, log_mu, log_nu, iters)  models/modules.py:137-143 ------
 * Z [batch,M,N], log_mu [batch,M], log_nu [batch,N] -> out [batch,M,N] = Z + u + v.
 * workspace: pats_sinkhorn_workspace_bytes(batch, M, N) bytes of device memory (may be NULL if
 * that returns 0). */
size_t pats_sinkhorn_workspace_bytes(int64_t batch, int M, int N);
int pats_sinkhorn_f32(const float* Z, int64_t batch, int M, int N, const float* log_mu,
                      const float* log_nu, int iters, float* out, void* workspace,
                      size_t workspace_bytes, pats_stream_t stream);

/* ---- a4: log_optimal_transport(scores, alpha, ns, iters)  models/modules.py:145-162 --------
 * scores [batch,m,n]; alpha: DEVICE pointer to one float (the reference's 0-d `bin_score.abs()`);
 * ns [batch,n] target areas (the reference's [b,1,n]) -> Z [batch,m+1,n+1] log-plan with dustbin
 * row/col, already `- norm`.  workspace: pats_ot_workspace_bytes(batch, m+1, n+1). */
size_t pats_ot_workspace_bytes(int64_t batch, int M, int N);
int pats_log_optimal_transport_f32(const float* scores, int64_t batch, int m, int n,
                                   const float* alpha, const float* ns, int iters, float* Z,
                                   void* workspace, size_t workspace_bytes, pats_stream_t stream);

/* ---- a5: log_optimal_transport2(scores, one, ns, iters)  models/modules.py:165-182 ---------
 * scores [batch,m,n] whose last row/col are already the dustbin; one: DEVICE pointer to one float
 * or NULL (= 1.0f); ns [batch,n-1] -> Z [batch,m,n].  bias_k > 0 additionally applies the
 * caller's  Z[:,:,-1] += log(k); Z[:,-1,:] += log(k)  (second_layer.py:107-112); 0 = none.
 * workspace: pats_ot2_workspace_bytes(batch, m, n) - 0 for 65 x 65, one guard flag per problem for
 * 145 x 145 (the resident kernels), pats_ot_workspace_bytes otherwise. */
size_t pats_ot2_workspace_bytes(int64_t batch, int m, int n);
/* the same plus est_position's if_nomatching2 = (Z.max(1).indices == m-1) for the first n-1 columns
 * (second_layer.py:243,248; first index wins ties): col_nomatch [batch,n-1] uint8, written by the 145 x 145
 * kernel's own epilogue (other shapes: one extra pass over Z).  Row flags come from pats_iterative_expand_f32. */
int pats_log_optimal_transport2_flags_f32(const float* scores, int64_t batch, int m, int n,
                                          const float* one, const float* ns, int iters, float bias_k,
                                          float* Z, uint8_t* col_nomatch, void* workspace,
                                          size_t workspace_bytes, pats_stream_t stream);
int pats_log_optimal_transport2_f32(const float* scores, int64_t batch, int m, int n,
                                    const float* one, const float* ns, int iters, float bias_k,
                                    float* Z, void* workspace, size_t workspace_bytes,
                                    pats_stream_t stream);

/* ---- a1-a3 + a4/a5 fused: descriptors -> log-plan, no score matrix round trip ---------------
 * variant 1 = log_optimal_transport (Z [batch,n+1,m+1], ns [batch,m]);
 * variant 2 = log_optimal_transport2 (Z [batch,n,m], ns [batch,m-1]).
 * scalar = alpha (variant 1) or one (variant 2), device pointer (NULL = 0 / 1). */
int pats_cost_ot_f32(const float* d0, const float* d1, int64_t batch, int D, int n, int m,
                     int variant, const float* scalar, const float* ns, int iters, float bias_k,
                     float* Z, void* workspace, size_t workspace_bytes, pats_stream_t stream);
size_t pats_cost_ot_workspace_bytes(int64_t batch, int D, int n, int m, int variant);
/* variant 2 with the column flags of pats_log_optimal_transport2_flags_f32 (same workspace). */
int pats_cost_ot_flags_f32(const float* d0, const float* d1, int64_t batch, int D, int n, int m,
                           int variant, const float* scalar, const float* ns, int iters, float bias_k,
                           float* Z, uint8_t* col_nomatch, void* workspace, size_t workspace_bytes,
                           pats_stream_t stream);

/* Measurement hook: `event` (a hipEvent_t, or NULL) is recorded once, by the next two-kernel pats_cost_ot*_f32 call of the calling
 * thread, between its cost-build launch and its Sinkhorn launch (bench.py times the two kernels inside its steps with it).  ABI 5. */
int pats_set_cost_ot_mid_event(void* event);

/* The fine level launched over a CAPACITY of batch_cap problems with the number in use on the device (throughput mode:
 * batch_dev = the row table's total, &chunk_base[Cmax] of pats_chunk_rows_device): workgroups of problems >= *batch_dev return
 * at once - no cost build, no solve, no log-domain redo; their rows of Z / col_nomatch are left untouched.  variant 2,
 * n = m = 145 only; everything else as pats_cost_ot_flags_f32 (workspace sized for batch_cap).  ABI 5. */
int pats_cost_ot_flags_counted_f32(const float* d0, const float* d1, int64_t batch_cap, const int64_t* batch_dev,
                                   int D, int n, int m, int variant, const float* scalar, const float* ns, int iters,
                                   float bias_k, float* Z, uint8_t* col_nomatch, void* workspace,
                                   size_t workspace_bytes, pats_stream_t stream);

/* ---- a7: post-OT reductions ----------------------------------------------------------------
 * colmass: out[b,j] = sqrtf(sum_{i<M-1} expf(Z[b,i,j]) + 1e-8f), j < N-1   first_layer.py:117-118
 * bias   : Z[:,:,-1] += logf(k); Z[:,-1,:] += logf(k) in place              second_layer.py:107-112
 * exp    : out = expf(Z)                                                    third_layer.py:159 */
int pats_colmass_sqrt_f32(const float* Z, int64_t batch, int M, int N, float* out,
                          pats_stream_t stream);
/* colmass plus est_position's if_nomatching2 = (scores.max(1).indices == M-1) in the same pass over the
 * columns (first_layer.py:163,167): col_nomatch [batch,N-1] uint8.  Either output may be NULL. */
int pats_colmass_flags_f32(const float* Z, int64_t batch, int M, int N, float* out, uint8_t* col_nomatch,
                           pats_stream_t stream);
int pats_dustbin_bias_inplace_f32(float* Z, int64_t batch, int M, int N, float k,
                                  pats_stream_t stream);
int pats_exp_f32(const float* Z, int64_t count, float* out, pats_stream_t stream);

/* ---- a8: scores.max(2).indices / scores.max(1).indices, first index wins ties ---------------
 * first_layer.py:162  second_layer.py:243.  row_arg [batch,M], col_arg [batch,N]; either NULL. */
int pats_argmax_f32(const float* Z, int64_t batch, int M, int N, int64_t* row_arg,
                    int64_t* col_arg, pats_stream_t stream);

/* ---- a9-a11: Iterative_expand_matrix + Compute_scaling  utils/utils.py:1179-1297,1321-1340 --
 * P [batch,M,N] = exp(Z) incl. dustbin row/col (or Z itself when input_is_log != 0: the kernel
 * exponentiates on load, saving the exp(Z) round trip of first_layer.py:174 / second_layer.py:255);
 * scalex, scaley [batch,N-1]; lim3 = limitation[3]; (h, w) = the TRUE grid the caller built
 * positions/ranges for (Compute_positions_and_ranges, utils.py:1527-1537).
 * outputs: whole_cost, core_cost, x_scale, y_scale [batch,M-1]; average_point [batch,M-1,2];
 * bound [batch,M-1,4] int64 (up,down,left,right).
 *
 * row_nomatch (optional, [batch,M-1] uint8): est_position's if_nomatching1 = (scores.max(2).indices == N-1)
 * (first_layer.py:162-164, second_layer.py:243-245) taken from the values as handed in - the row is in LDS
 * here anyway, so the separate argmax pass over the plan disappears. */
int pats_iterative_expand_f32(const float* P, int input_is_log, int64_t batch, int M, int N,
                              const float* scalex, const float* scaley, int lim3, int h, int w,
                              float lower_bound, int iter_num, float* whole_cost, float* core_cost,
                              float* average_point, float* x_scale, float* y_scale, int64_t* bound,
                              uint8_t* row_nomatch, pats_stream_t stream);

/* The same over a capacity of batch_cap problems, *batch_dev of them in use (device-side count); the outputs of the others are
 * left untouched.  ABI 5. */
int pats_iterative_expand_counted_f32(const float* P, int input_is_log, int64_t batch_cap, const int64_t* batch_dev,
                                      int M, int N, const float* scalex, const float* scaley, int lim3, int h, int w,
                                      float lower_bound, int iter_num, float* whole_cost, float* core_cost,
                                      float* average_point, float* x_scale, float* y_scale, int64_t* bound,
                                      uint8_t* row_nomatch, pats_stream_t stream);

/* ---- a12: split_patches(sum_cycle, height, width, max_once_used)  utils/utils.py:152-181 ----
 * HOST function on a host copy of the int32 cumsum (the reference syncs per comparison; here the
 * caller pays one D->H copy).  second/third: [height+1][2] int64.  Returns cycle_num (>= 1), or
 * a negative PATS_ERR code. */
int pats_split_patches(const int32_t* sum_cycle_host, int height, int width, int max_once_used,
                       int64_t* second_layer_set, int64_t* third_layer_set);
/* The same planner on the DEVICE for a batch of pairs (throughput mode: no host read at all):
 * sum_cycle [pairs, height*width] int32 device cumsums -> second/third [pairs, height+1, 2] int64
 * (unused rows zeroed) and cycle_num [pairs] int32, all device memory.  One thread per pair. */
int pats_split_patches_device(const int32_t* sum_cycle, int64_t pairs, int height, int width,
                              int max_once_used, int64_t* second_layer_set, int64_t* third_layer_set,
                              int32_t* cycle_num, pats_stream_t stream);

/* ---- a13: Compute_imgs bounds  utils/utils.py:1350-1382 ------------------------------------
 * x_scale, y_scale [Np]; average_point [Np,2]; if_nomatching [Np] uint8; grid (height,width).
 * -> bound5 [Np,5] int64, only the first *K rows valid (y0,y1,x0,x1,img*10000+patch) in patch
 *    order; K_out: DEVICE int64 count; x_scale_new, y_scale_new, average_new [Np,2]. */
int pats_compute_imgs_bounds_f32(const float* x_scale, const float* y_scale,
                                 const float* average_point, const uint8_t* if_nomatching, int Np,
                                 int height, int width, int img, int64_t* bound5, int64_t* K_out,
                                 float* x_scale_new, float* y_scale_new, float* average_new,
                                 pats_stream_t stream);

/* ---- a13: left crops = origin_extract on the 32-px padded left image  utils.py:1300-1318,1383
 * left [n_img,H,W,3] HWC fp32; bound5/K as produced above (row k: image bound5[k,4] / 10000, patch
 * bound5[k,4] % 10000, the reference's `sequence`, utils.py:1374-1377) -> out [K,96,96,3].  K is read on
 * the host side by the caller (max rows = K_cap). */
int pats_left_crops_f32(const float* left, int n_img, int H, int W, const int64_t* bound5, int64_t K,
                        int height, int width, float* out, pats_stream_t stream);

/* a13 for a BATCH of images without host-side counts (throughput mode).  x_scale, y_scale [n_img,Np],
 * average_point [n_img,Np,2], if_nomatching [n_img,Np] -> bound5 [n_img*Np,5] compacted in (image, patch)
 * order (sequence = img * 10000 + patch, utils.py:1374-1377), K_img [n_img] matches per image and
 * K_total [1] = valid rows of bound5 (DEVICE int64), x_scale_new, y_scale_new, average_new [n_img,Np,2].
 * The `_counted` gathers are launched over K_cap rows and skip rows >= *K_dev. */
int pats_compute_imgs_bounds_batch_f32(const float* x_scale, const float* y_scale, const float* average_point,
                                       const uint8_t* if_nomatching, int n_img, int Np, int height, int width,
                                       int64_t* bound5, int64_t* K_img, int64_t* K_total, float* x_scale_new,
                                       float* y_scale_new, float* average_new, pats_stream_t stream);
int pats_left_crops_counted_f32(const float* left, int n_img, int H, int W, const int64_t* bound5,
                                int64_t K_cap, const int64_t* K_dev, int height, int width, float* out,
                                pats_stream_t stream);

/* ---- a14: tensor_resize(input, bound)  setup/library.cpp:47-66 (module def :92-93) ----------
 * input [n_img,C,Hp,Wp] fp32; bound [K,5] int64 (y0,y1,x0,x1,seq), image = seq / 10000;
 * crop rows [y0,y1) x cols [x0,x1] -> bilinear align_corners=True -> out [K,C,96,96].
 * One launch, no host sync (the reference does 5 .item() syncs per crop).  status: optional DEVICE
 * int32 that is set non-zero if any crop is empty / out of range (torch raises there); the kernel
 * itself clamps reads so it is always memory-safe.  K == 0 is a no-op. */
int pats_tensor_resize_f32(const float* input, int n_img, int C, int Hp, int Wp,
                           const int64_t* bound, int64_t K, float* out, int32_t* status,
                           pats_stream_t stream);
/* same, fused with the zero padding of utils.py:1352 and the HWC->CHW permute: reads the
 * UNPADDED right image [n_img,H,W,3] (margin = 128) and writes [K,96,96,3] (the layout the caller
 * permutes to at utils.py:1385). */
int pats_tensor_resize_hwc_f32(const float* right, int n_img, int H, int W, int margin,
                               const int64_t* bound, int64_t K, float* out, int32_t* status,
                               pats_stream_t stream);
int pats_tensor_resize_hwc_counted_f32(const float* right, int n_img, int H, int W, int margin,
                                       const int64_t* bound, int64_t K_cap, const int64_t* K_dev, float* out,
                                       int32_t* status, pats_stream_t stream);

/* ---- a17 + a18: ThirdLayer.Compute_result + match label  models/third_layer.py:161-170,184-217
 * scores [P,65,65] = exp(Z) (or Z when input_is_log); scale_x, scale_y [P,64]; p_s, p_t [P,2]
 * int64 -> mkpts0_f, mkpts1_f [P,16,2]; whole_loss [P,16]; label [P*16,2]; if_matching1 [P,16]
 * uint8.  W = 8, T = 5 as in the reference. */
int pats_compute_result_f32(const float* scores, int input_is_log, int64_t P, const float* scale_x,
                            const float* scale_y, const int64_t* p_s, const int64_t* p_t,
                            int outdoor, float* mkpts0_f, float* mkpts1_f, float* whole_loss,
                            float* label, uint8_t* if_matching1, pats_stream_t stream);
/* the same with the 4 bytes of device workspace whole_loss needs (its cross-problem count, :215) handed in by the
 * caller: no allocation inside the call (pats_compute_result_f32 takes a stream-ordered one), capturable in a HIP graph */
int pats_compute_result_ws_f32(const float* scores, int input_is_log, int64_t P, const float* scale_x,
                               const float* scale_y, const int64_t* p_s, const int64_t* p_t,
                               int outdoor, float* mkpts0_f, float* mkpts1_f, float* whole_loss,
                               float* label, uint8_t* if_matching1, void* workspace, size_t workspace_bytes,
                               pats_stream_t stream);

/* ---- a15: fine-level descriptor sampling  models/second_layer.py:71-86 -----------------------
 * feat0 [2B,64,48,48], feat1 [2B,64,24,24], feat2 [2B,128,12,12] (ResNet2.forward2 of the stacked
 * left|right crops), title [B,8] (= compress_1(desc_l)), rubbish [B,264] (= compress_2(desc_l))
 * -> desc [2,B,264,145]: 8 title channels, AvgPool2d(2,1,1)+grid samples at strides 4/2/1
 * (64+64+128 channels), dustbin feature column last.  desc[0] / desc[1] are mdesc inputs of the GNN. */
int pats_fine_descriptors_f32(const float* feat0, const float* feat1, const float* feat2,
                              const float* title, const float* rubbish, int64_t B, float* desc,
                              pats_stream_t stream);

/* a15 over a capacity of B_cap rows, *B_dev of them in use (device-side count; the desc blocks of the others are left
 * untouched); channels_last != 0: the maps are torch.channels_last as for pats_fine_descriptors_nhwc_f32.  ABI 5. */
int pats_fine_descriptors_counted_f32(const float* feat0, const float* feat1, const float* feat2,
                                      const float* title, const float* rubbish, int64_t B_cap,
                                      const int64_t* B_dev, int channels_last, float* desc, pats_stream_t stream);

/* ---- a16: third-level 8x8 window gather  models/third_layer.py:121-146 -----------------------
 * feat_f0, feat_f1 [B,128,52,52]; mkpts0_c, mkpts1_c [P,2] float (x, y) coarse points in crop
 * pixels; b_ids [P] int64; kenc [128,64] (= self.kenc(kpts)[0]); rubbish [B,128,144]
 * -> out0, out1 [P,128,65] (window cell t = wy*8 + wx, dustbin feature at column 64), and the
 * rounded points the caller keeps using (p_s_out, p_t_out [P,2] int64; may be NULL).
 * Out-of-map indices (torch.gather would raise) are clamped. */
int pats_third_descriptors_f32(const float* feat_f0, const float* feat_f1, const float* mkpts0_c,
                               const float* mkpts1_c, const int64_t* b_ids, const float* kenc,
                               const float* rubbish, int64_t P, int64_t B, float* out0, float* out1,
                               int64_t* p_s_out, int64_t* p_t_out, pats_stream_t stream);

/* ---- the whole third-level step in ONE launch: a3 + a5 + a7(exp) + a17 + a18 ------------------
 * feat0, feat1 [P,D,65] (D a multiple of 32, at most 512; 128 in the reference) -> cost build (third_layer.py:156-157),
 * log_optimal_transport2(0.1*scores, 1, scale, iters) (:158), exp (:159), Compute_result (:160,
 * :184-217) and the label (:161-170).  One wave per problem; the 65x65 plan never leaves the CU
 * unless Z_out != NULL ([P,65,65] log-plan).  scale [P,64] = target areas (the OT's `ns`);
 * scale_x, scale_y [P,64] = sqrt(scale + 1e-8) as the caller computes them (:153-154).
 * whole_loss is not produced (unused at inference; use pats_compute_result_f32 for it). */
int pats_third_level_f32(const float* feat0, const float* feat1, int64_t P, int D, const float* scale,
                         const float* scale_x, const float* scale_y, const int64_t* p_s,
                         const int64_t* p_t, int iters, int outdoor, float* mkpts0_f, float* mkpts1_f,
                         float* label, uint8_t* if_matching1, float* Z_out, pats_stream_t stream);

/* a16 / the third-level step when the number of problems lives on the DEVICE (throughput mode: the merge decides P and
 * nothing reads it back): the launch covers the capacity P_cap, workgroups past *P_dev leave at once and their output rows
 * are not written.  pats_third_level_counted_f32 takes scale_x = scale_y = NULL to form sqrt(scale + 1e-8)
 * (third_layer.py:153-154) in the kernel instead of reading the caller's copies. */
int pats_third_descriptors_counted_f32(const float* feat_f0, const float* feat_f1, const float* mkpts0_c,
                                       const float* mkpts1_c, const int64_t* b_ids, const float* kenc,
                                       const float* rubbish, int64_t P_cap, const int64_t* P_dev, int64_t B,
                                       float* out0, float* out1, int64_t* p_s_out, int64_t* p_t_out,
                                       pats_stream_t stream);
int pats_third_level_counted_f32(const float* feat0, const float* feat1, int64_t P_cap, const int64_t* P_dev, int D,
                                 const float* scale, const float* scale_x, const float* scale_y, const int64_t* p_s,
                                 const int64_t* p_t, int iters, int outdoor, float* mkpts0_f, float* mkpts1_f,
                                 float* label, uint8_t* if_matching1, pats_stream_t stream);

/* a15 / a16 on CHANNELS-LAST maps (torch.channels_last: logical [B,C,H,W], memory [B,H,W,C]) - the layout a backbone run
 * under MIOpen emits natively, and the one in which the per-pixel reads of second_layer.py:73-79 and of
 * `feat.permute(0, 2, 3, 1).reshape(-1, C)` + torch.gather (third_layer.py:139-140) are contiguous runs of 256 / 512
 * bytes.  Arguments, outputs and every output bit as in pats_fine_descriptors_f32 / pats_third_descriptors_*_f32; only
 * the memory order of feat0/1/2 ([2B,48,48,64], [2B,24,24,64], [2B,12,12,128]) and feat_f0/f1 ([B,52,52,128]) differs
 * (title, rubbish, kenc as before).  Maps and desc 16-byte aligned.  P_dev may be NULL (then P_cap points exist). */
int pats_fine_descriptors_nhwc_f32(const float* feat0, const float* feat1, const float* feat2,
                                   const float* title, const float* rubbish, int64_t B, float* desc,
                                   pats_stream_t stream);
int pats_third_descriptors_nhwc_f32(const float* feat_f0, const float* feat_f1, const float* mkpts0_c,
                                    const float* mkpts1_c, const int64_t* b_ids, const float* kenc,
                                    const float* rubbish, int64_t P_cap, const int64_t* P_dev, int64_t B,
                                    float* out0, float* out1, int64_t* p_s_out, int64_t* p_t_out,
                                    pats_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The steps either side of the OT path.  bool tensors are 1 byte, 0 / 1.
 * ---------------------------------------------------------------------------------------- */

/* ---- f3, throughput mode: the chunk loop of PATS.forward for a BATCH of pairs without a host read ----------------
 * Reference: models/first_layer.py:130-146 (cumsum of the matched flags, split_patches, one boolean mask + crop gather
 * + SecondLayer call per chunk), models/pats.py:33-39 (the loop; `if_nomatching1[-third_layer_set[num][1]:, :] = True`).
 * From if_nomatching1 [pairs, N] (N = height * width) this builds, on the device:
 *   sum_cycle [pairs,N] int32, cycle_num [pairs], second / third [pairs, height+1, 2]   (= pats_split_patches_device)
 *   masks [Cmax, pairs, N]      the chunk masks of first_layer.py:137-138 (1 = cell not in the chunk), chunk-major
 *   the fine level's ROW TABLE, rows ordered (chunk, pair, cell): chunk c of every pair is the contiguous block
 *     [chunk_base[c], chunk_base[c+1]) - chunk_base [Cmax+1], chunk_base[Cmax] = total rows;
 *     row_cell [rows_cap] = pair * N + cell (-1 past the total); row_crop [rows_cap] = the row's crop in the (image,
 *     patch)-ordered crop table of pats_compute_imgs_bounds_batch_f32 (crop_base [pairs+1] = first crop of every pair);
 *     row_forced [rows_cap] = 1 for the trailing rows pats.py:38-39 masks (and for padding); row_slot [Cmax, pairs*N] =
 *     the row of (chunk, cell) or -1.
 * Capacities are host-side: Cmax >= the largest chunk count (<= floor((N-1)/max_once_used) + 1 and <= height + 1),
 * rows_cap >= total rows (<= pairs * (N + (Cmax-1) * width)).  status (device int32): bit 0 = a pair had more than Cmax
 * chunks, bit 1 = more rows than rows_cap (rows were dropped) - read it when convenient, e.g. with the results. */
size_t pats_chunk_rows_workspace_bytes(int64_t pairs, int Cmax);
int pats_chunk_rows_device(const uint8_t* if_nomatching1, int64_t pairs, int height, int width, int max_once_used,
                           int Cmax, int64_t rows_cap, int32_t* sum_cycle, int32_t* cycle_num, int64_t* second,
                           int64_t* third, uint8_t* masks, int64_t* chunk_base, int64_t* crop_base, int32_t* row_cell,
                           uint8_t* row_forced, int32_t* row_crop, int32_t* row_slot, int32_t* status, void* workspace,
                           size_t workspace_bytes, pats_stream_t stream);

/* Scheduling aid, no reference counterpart: a HIP stream restricted to the compute units set in cu_mask (bit i of word
 * i / 32 = CU i; `words` 32-bit words - 8 for the 256 CUs of an MI355X).  The throughput path runs its HBM-bound stages
 * (crops, descriptor gathers) and its VALU-bound stages (the three solvers) of consecutive batches on two such streams with
 * disjoint masks (bench.py --overlap); wrap the handle with torch.cuda.ExternalStream to use it from PyTorch. */
int pats_stream_create_cu_mask(const uint32_t* cu_mask, int words, pats_stream_t* stream);
int pats_stream_destroy(pats_stream_t stream);

/* Profiling aid, no reference counterpart: launches an empty kernel named pats::profile_marker_kernel on `stream`, so that
 * a kernel trace can be cut to the region between two markers (bench.py brackets its timed steps with it). */
int pats_profile_marker(int tag, pats_stream_t stream);

/* SecondLayer.merge_patches_new / _old (models/second_layer.py:137-238) for every chunk of every pair of the row table
 * above, in the reference's order: chunk blocks one after the other (the chunks of a pair couple through scores_back,
 * pats.py:32,37), each block over all pairs at once.  trust_score / if_nomatching1_L2 [rows_cap,144] are updated in place
 * like the reference; scores_back [pairs, N, 16, 9] fp64 (zeros before the first chunk, pats.py:32: zero_scores_back != 0
 * clears it here); out [rows_cap,144] =
 * the returned if_nomatching with pats.py:38-39 applied (row_forced) and padding rows all "no match". */
size_t pats_merge_batch_workspace_bytes(int64_t pairs, int H, int W);
int pats_merge_patches_batch(int merge_new, int Cmax, int64_t pairs, int H, int W, int64_t rows_cap,
                             const int64_t* chunk_base, const int32_t* row_cell, const int32_t* row_slot,
                             const uint8_t* row_forced, float* trust_score, uint8_t* if_nomatching1_L2,
                             double* scores_back, int zero_scores_back, uint8_t* out, void* workspace,
                             size_t workspace_bytes, pats_stream_t stream);

/* The same merges for chunks [c_lo, c_hi) of the row table only, on tensors that hold table rows row_origin .. row_origin +
 * rows_local (trust_score, if_nomatching1_L2, out [rows_local,144]): PATS.forward's chunk loop (models/pats.py:33-39) walked
 * chunk by chunk - one launch per chunk, scores_back handed from call to call (zero_scores_back != 0 on the first,
 * pats.py:32), pats.py:38-39 applied through row_forced.  Rows of `out` outside the walked blocks read "no match". */
int pats_merge_patches_chunks(int merge_new, int Cmax, int c_lo, int c_hi, int64_t pairs, int H, int W, int64_t row_origin,
                              int64_t rows_local, const int64_t* chunk_base, const int32_t* row_cell, const int32_t* row_slot,
                              const uint8_t* row_forced, float* trust_score, uint8_t* if_nomatching1_L2, double* scores_back,
                              int zero_scores_back, uint8_t* out, void* workspace, size_t workspace_bytes, pats_stream_t stream);

/* PATS.forward's chunk loop (models/pats.py:33-78) walked chunk by chunk: everything between the network callbacks of ONE chunk in
 * two calls (csrc/chunk_walk.cpp - each runs the entry points above in the reference's order on `stream`; what they save is host
 * time).  B = the chunk's rows; the row table is pats_chunk_rows_device's, c the chunk, row_origin = chunk_base[c] (host copy).
 *   fine tail: second_layer.py:100-122 + pats.py:37-39,53-58.  In: the chunk's descriptors d0, d1 [B,264,145], one (device 1.0f),
 *   ns = scale_x * scale_y [B,144], the two scale heads.  Out: Z [B,145,145] (+ ln bias_k), col_nomatch / row_nomatch [B,144], the
 *   expansion's six outputs, merged [B,144] (the returned if_nomatching, tail rows forced), mkpts0 / mkpts1 [144 B,2], b_ids [144 B],
 *   *P_dev.  trust and row_nomatch are updated in place by the merge as in the reference.  wait_before_merge / record_after_merge:
 *   optional hipEvent_t handles - the stream waits for the first right before the merge and records the second right behind it (the
 *   merges of a pair are ordered, pats.py:37; with consecutive chunks on different streams only they need to wait for each other).
 *   third tail: third_layer.py:153-170 + pats.py:59-78.  In: the third level's descriptors over the capacity P_cap = 144 B with the
 *   count *P_dev, scale [P_cap,64], the rounded points, merged / points2 of the fine tail, the chunk's mask [h w] (level-0 flags),
 *   Compute_imgs' pts_new / scales [1,h w,2], `ones` = B bytes of 1.  Out: the third level's four outputs, if_nomatching16 [B,2304],
 *   pts16 [B,2304,2], matches_l / matches_r [2304 B,2], match_row, *M_dev. */
size_t pats_chunk_fine_tail_workspace_bytes(int64_t B, int64_t pairs, int H, int W);
int pats_chunk_fine_tail_f32(const float* d0, const float* d1, int64_t B, const float* one, const float* ns, int iters, float bias_k,
                             const float* scale_x, const float* scale_y, int merge_new, int Cmax, int c, int64_t pairs, int H, int W,
                             int64_t row_origin, const int64_t* chunk_base, const int32_t* row_cell, const int32_t* row_slot,
                             const uint8_t* row_forced, double* scores_back, int first_chunk, float* Z, uint8_t* col_nomatch,
                             float* trust, float* core, float* points, float* x_scale, float* y_scale, int64_t* bound,
                             uint8_t* row_nomatch, uint8_t* merged, float* mkpts0, float* mkpts1, int64_t* b_ids, int64_t* P_dev,
                             void* wait_before_merge, void* record_after_merge, void* workspace, size_t workspace_bytes,
                             pats_stream_t stream);
size_t pats_chunk_third_tail_workspace_bytes(int64_t B, int h, int w);
int pats_chunk_third_tail_f32(const float* feat0, const float* feat1, int64_t P_cap, const int64_t* P_dev, const float* scale,
                              const int64_t* p_s, const int64_t* p_t, int iters, int outdoor, const uint8_t* merged,
                              const float* points2, int64_t B, const uint8_t* chunk_mask, int h, int w, const float* pts_new,
                              const float* scales, const uint8_t* ones, float* mkpts0_f, float* mkpts1_f, float* label,
                              uint8_t* if_matching1, uint8_t* if_nomatching16, float* pts16, float* matches_l, float* matches_r,
                              int32_t* match_row, int64_t* M_dev, void* workspace, size_t workspace_bytes, pats_stream_t stream);

/* SecondLayer.merge_patches_new (merge_new != 0, reference models/second_layer.py:193-240) and
 * merge_patches_old (merge_new == 0, :137-191): resolves every 8-px cell among the up to nine 96x96
 * windows covering it.  Like the reference it works in place on trust_score [B,144] (border weighting
 * :194-198, and -10000 on matching cells for "new" :201) and on if_nomatching1_L2 [B,144] (:199-200),
 * and writes this chunk's scores into scores_back [batch_num, H/32*W/32, 16, 9] (fp64, :211; the
 * "new" variant reads the other patches' entries left there by earlier chunks, pats.py:32,37).
 * `out` [B,144] is the returned if_nomatching (rows = unmasked entries of if_nomatching1_L1, in
 * order).  The reference raises when the number of unmasked coarse patches differs from B; here
 * surplus patches are ignored and missing ones leave their rows "no match" - host wrappers validate. */
size_t pats_merge_workspace_bytes(int64_t B, int H, int W, int batch_num);
int pats_merge_patches(int merge_new, int64_t B, float* trust_score, int H, int W, int batch_num,
                       const uint8_t* if_nomatching1_L1, uint8_t* if_nomatching1_L2, double* scores_back,
                       uint8_t* out, void* workspace, size_t workspace_bytes, pats_stream_t stream);

/* Workspace of the order-preserving compactions below over n flags. */
size_t pats_compact_workspace_bytes(int64_t n);

/* Third-level inputs (reference models/pats.py:53-58): for every L2 cell with if_nomatching == 0, in
 * (row, cell) order, mkpts0 = ((cell % 12) * 4 + 2, (cell / 12) * 4 + 2) * 2, mkpts1 =
 * round(pts * 4)[(1, 0)] * 2 (round half to even) and the patch row b_ids.  Writes at most `capacity`
 * rows; *count (device int64) receives the number of surviving cells P. */
int pats_third_inputs_f32(const uint8_t* if_nomatching, const float* pts, int64_t B, float* mkpts0,
                          float* mkpts1, int64_t* b_ids, int64_t capacity, int64_t* count, void* workspace,
                          size_t workspace_bytes, pats_stream_t stream);

/* Scatter of the third-level results onto the 48x48 sub-cell grid (reference models/pats.py:59-67):
 * pts16 [B,2304,2] takes mkpts1_f [P,16,2] where the L2 cell survived (else the L2 point), and
 * if_nomatching16 [B,2304] = L2 flag OR label < -9.9; layout [B,12,4,12,4].  `label` is read with a
 * stride (2 for the reference's label[:, 0] of a [P*16,2] tensor). */
int pats_refine_scatter_f32(const uint8_t* if_nomatching, const float* pts, const float* mkpts1_f,
                            const float* label, int label_stride, int64_t B, int64_t P,
                            uint8_t* if_nomatching16, float* pts16, void* workspace,
                            size_t workspace_bytes, pats_stream_t stream);

/* get_result with layer_num = 2 (reference utils/utils.py:189-213, called at models/pats.py:73):
 * level 0 has batch_size rows of patch_size0[1]*patch_size0[2] cells of patch_size0[0] px, level 1 has
 * `rows1` rows (one per surviving level-0 cell, in order) of patch_size1[1]*patch_size1[2] sub-cells.
 * scale1_cell_stride = 2: scale1 is [rows1, n1, 2] as the reference materialises it; 0: scale1 is
 * [rows1, 2], one scale per row (what pats.py:70 repeats).  Output order = reference (row, sub-cell);
 * at most `capacity` rows are written, *count (device int64) receives M.  fp32, operation order of the
 * reference.  Workspace: pats_get_result_workspace_bytes(batch_size*n0, rows1, n1). */
size_t pats_get_result_workspace_bytes(int64_t cells0, int64_t rows1, int64_t cells1);
int pats_get_result_f32(int batch_size, const uint8_t* if_nomatching0, const uint8_t* if_nomatching1,
                        int64_t rows1, const float* average_point0, const float* average_point1,
                        const float* scale0, const float* scale1, int64_t scale1_cell_stride,
                        const int* patch_size0, const int* patch_size1, const uint8_t* left_choice0,
                        const uint8_t* left_choice1, float* matches_l, float* matches_r, int64_t capacity,
                        int64_t* count, void* workspace, size_t workspace_bytes, pats_stream_t stream);

/* get_result for every (chunk, pair) of a batch in ONE call (models/pats.py:68-73, utils/utils.py:189-213): level-0
 * batch = the Cmax * pairs chunk masks, level-1 rows = the row table of pats_chunk_rows_device.  pts_new, scales
 * [pairs, N, 2] are the per-pair tensors of Compute_imgs (not expanded over the chunks, not flipped - pats.py:71's
 * `.flip(dims=[2]) / 32.0` happens on load), pts16 [rows_cap, n1, 2] un-flipped (`/ 2.0` on load), the level-1 scale of a
 * row is the level-0 scale of its cell (pats.py:70).  match_row [capacity] (optional) = the row of every match: row_cell
 * [match_row] / N is its pair.  Same arithmetic and output order as pats_get_result_f32 on the expanded tensors. */
int pats_get_result_chunks_f32(int Cmax, int64_t pairs, const uint8_t* masks, const uint8_t* if_nomatching16,
                               int64_t rows_cap, const float* pts_new, const float* pts16, const float* scales,
                               const int* patch_size0, const int* patch_size1, const uint8_t* left_choice0,
                               const uint8_t* left_choice1, float* matches_l, float* matches_r, int32_t* match_row,
                               int64_t capacity, int64_t* count, void* workspace, size_t workspace_bytes,
                               pats_stream_t stream);

/* The matches of a batch grouped BY PAIR on the device (ABI 5; throughput mode's hand-over, no reference counterpart: the
 * reference runs one pair at a time).  Input = what pats_get_result_chunks_f32 wrote (matches in (chunk, pair, patch, sub-cell)
 * order, match_row, the count *M_dev) + the row table's row_cell / chunk_base; output = the same matches with every pair's list
 * contiguous and in the reference's order (its chunks one after the other), pair_off [pairs + 1] int64: pair p owns rows
 * [pair_off[p], pair_off[p + 1]) of out_l / out_r.  No host read. */
size_t pats_matches_by_pair_workspace_bytes(int Cmax, int64_t pairs);
int pats_matches_by_pair_f32(const float* matches_l, const float* matches_r, const int32_t* match_row, const int64_t* M_dev,
                             const int32_t* row_cell, const int64_t* chunk_base, int Cmax, int64_t pairs, int N,
                             float* out_l, float* out_r, int64_t* pair_off, void* workspace, size_t workspace_bytes,
                             pats_stream_t stream);
/* The same with the step's counters appended: pair_off has pairs + 4 entries - the pairs + 1 offsets, then M, P (*P_dev: the
 * third-level problem count of the step, may be null) and the row table's status - a batch's hand-over is ONE device-to-host copy. */
int pats_matches_by_pair_summary_f32(const float* matches_l, const float* matches_r, const int32_t* match_row, const int64_t* M_dev,
                                     const int32_t* row_cell, const int64_t* chunk_base, int Cmax, int64_t pairs, int N,
                                     float* out_l, float* out_r, int64_t* pair_off, const int64_t* P_dev, const int32_t* status,
                                     void* workspace, size_t workspace_bytes, pats_stream_t stream);

/* attention(query, key, value) of the GNN layers (reference models/modules.py:84-88; the core of
 * MultiHeadedAttention.forward :100-105): scores = q^T k / dim**.5 per (batch, head), softmax over the
 * keys, out = prob v.  query [batch,dim,heads,n], key / value [batch,dim,heads,m] (the view
 * modules.py:101-102 makes of the projections) -> out [batch,dim,heads,n]; prob [batch,heads,n,m] is
 * written only if non-null (the reference returns it, its caller discards it).  fp32 throughout; the
 * score matrix stays in LDS (32 query rows per workgroup up to m = 1024, 16 / 8 / 4 rows beyond; m <= 8416, PATS_ERR_UNSUPPORTED
 * past that). */
int pats_attention_f32(const float* query, const float* key, const float* value, int64_t batch, int dim,
                       int heads, int n, int m, float* out, float* prob, pats_stream_t stream);

/* ---- AttentionalPropagation of the GNN layers (reference models/modules.py:91-117; section 8f rank 4) ----------
 *   message = merge(attention(proj_q(x), proj_k(source), proj_v(source)))            MultiHeadedAttention.forward :100-105
 *   delta   = mlp(cat([x, message], dim=1)),  mlp = Conv1d(2C,2C) BatchNorm1d ReLU Conv1d(2C,C)   :107-117 (MLP :57-69)
 *   out     = residual + delta  if residual != NULL  (AttentionalGNN.forward: desc = desc + delta, :131-133)
 * x [batch,C,n], source [batch,C,m] (channel-major, as the reference's Conv1d sees them), out [batch,C,n].
 * Weights are DEVICE pointers; the Conv1d matrices are handed over TRANSPOSED ([C_in][C_out], row = input channel),
 * i.e. conv.weight[:, :, 0].t().contiguous().  BatchNorm: bn_train == 0 -> bn_a / bn_b are the folded running
 * statistics (scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale); bn_train != 0 -> bn_a /
 * bn_b are gamma / beta and the batch statistics over (batch, n) are computed here with bn_eps (PATS.eval leaves the
 * third layer in train mode, models/pats.py:112-120).  The 1x1 convolutions use the contraction of pats_cost_f32 (the
 * cat is never materialised), the
 * attention core is pats_attention_f32.  C % heads == 0, C % 8 == 0, m <= 8416. */
typedef struct pats_propagation_weights {
    const float *wq_t, *bq;   /* attn.proj[0]: [C][C] transposed, [C] */
    const float *wk_t, *bk;   /* attn.proj[1] */
    const float *wv_t, *bv;   /* attn.proj[2] */
    const float *wm_t, *bm;   /* attn.merge */
    const float *w1_t, *b1;   /* mlp[0]: [2C][2C] transposed, [2C] */
    const float *bn_a, *bn_b; /* mlp[1]: [2C] each, see above */
    const float *w2_t, *b2;   /* mlp[3]: [2C][C] transposed, [C] */
} pats_propagation_weights;
size_t pats_attentional_propagation_workspace_bytes(int64_t batch, int C, int n, int m);
int pats_attentional_propagation_f32(const float* x, const float* source, int64_t batch, int C, int heads,
                                     int n, int m, const pats_propagation_weights* weights, int bn_train,
                                     float bn_eps, const float* residual, float* out, void* workspace,
                                     size_t workspace_bytes, pats_stream_t stream);

/* The same layer with the weights additionally PACKED for the matrix pipe (ABI 5): pats_propagation_pack_f32 splits every
 * Conv1d matrix into fp16 hi + lo halves in MFMA fragment order once per layer into a caller-owned, 16-byte aligned device
 * buffer of pats_propagation_packed_bytes(C, heads) bytes (C % 8 == 0, C % heads == 0; 0 = no packed form for this shape).
 * With it
 *  - at the third level's shape (C = 128, 4 heads, n = m = 65) the layer runs as ONE kernel that keeps a problem's activations in
 *    LDS from the descriptors to the output (csrc/gnn_fused.hip; q / k / v rows and the merge's columns permuted to head-major;
 *    bn_train != 0: one kernel up to the hidden tensor, then the batch-statistics passes and the last convolution);
 *  - at every other shape the six convolutions stream the packed weights from L2 straight into the matrix pipe
 *    (csrc/conv_pk.hip: 64 columns x up to 288 input channels staged whole in LDS per workgroup) and, between 97 and 160 tokens
 *    with 33 .. 80 channels per head (the fine level: [264, 145], 4 heads), the attention core keeps its scores in registers
 *    from the first product to the second (csrc/attention145.hip).
 * The packing also FOLDS the merge Conv1d into mlp[0] (modules.py:104,116: message = Wm att + bm has one consumer, hidden = W1
 * (x | message) + b1 = W1x x + (W1m Wm) att + (W1m bm + b1)): W1m Wm and the bias are formed once, in double, and kept - as fp32
 * [2C][2C] + [2C] - at the end of the packed buffer; the packed layer runs five products instead of six and differs from the
 * two-product form by fp32 rounding only (PATS_GNN_FOLD=0, read at pack time and at run time, keeps the merge as its own product).
 * The weights must not change between the packing and the calls that use it.
 * PATS_GNN_FUSED=0 / PATS_CONV_PK=0 / PATS_ATTN145=0 and launches in which an activation left the fp16 range of the split
 * operands take the composition above (same workspace; device-side flags, no host read). */
/* The layers' overflow protocol, no reference counterpart.  Mode 0 (default, "inline"): every packed call queues its fp32
 * composition behind the fast kernels, gated on a device-side flag - right without a host read, ~16 launches per layer that
 * normally do nothing.  Mode 1 ("deferred"): none is queued; a fast kernel that meets an activation beyond the fp16 range raises
 * ONE sticky flag per device, and the outputs of that call are then NOT valid.  The caller reads the flag where it synchronises
 * anyway - pats_gnn_overflows(&raised, reset): drains the device, raised = 0 / 1 - and repeats the work in mode 0 if it is up.
 * pats_set_gnn_redo_mode returns the previous mode (and leaves it unchanged for any other argument).  Process-wide. */
int pats_set_gnn_redo_mode(int mode);
int pats_gnn_overflows(int64_t* raised, int reset);
size_t pats_propagation_packed_bytes(int C, int heads);
int pats_propagation_pack_f32(const pats_propagation_weights* w, int C, int heads, void* packed, size_t packed_bytes,
                              pats_stream_t stream);
int pats_attentional_propagation_packed_f32(const float* x, const float* source, int64_t batch, int C, int heads,
                                            int n, int m, const pats_propagation_weights* w, const void* packed,
                                            int bn_train, float bn_eps, const float* residual, float* out,
                                            void* workspace, size_t workspace_bytes, pats_stream_t stream);

/* (ABI 6) At the FINE level's shape - C = 264, 4 heads, n = m = 145 (models/second_layer.py:44,89) - and bn_train == 0 the packed
 * layer is ONE kernel (csrc/gnn_fine.hip): a persistent workgroup per CU owns a problem, the descriptors arrive as pre-split
 * fragment images by LDS DMA, every product's output lives in the accumulators, q / k / v / attention / hidden stay in a
 * per-workgroup scratch block.  The BatchNorm scale / shift applied there are the ones in `w` AT PACK TIME (eval mode).
 * pats_attentional_propagation_packed_f32 converts a layer's [batch, C, n] tensors on the way in and out;
 * pats_attentional_gnn_packed_f32 is AttentionalGNN.forward (models/modules.py:127-134: `layers` propagations on both descriptor
 * sets, cross[l] != 0 = 'cross', the residual of :133 included) with the descriptors kept in the kernel's own form between the
 * layers.  weights[l] / packed[l]: the layer's weights and its pats_propagation_pack_f32 buffer.  Returns PATS_ERR_UNSUPPORTED
 * at any other shape (nothing launched) or when the kernel's LDS attribute is refused at run time (then a workspace fill and the
 * two input conversions have already been queued: harmless, the outputs are untouched) - run the layers one by one then.  A launch that meets a non-finite value raises a
 * device-side flag; the per-layer compositions queued behind, gated on it, redo the stack (no host read).  PATS_GNN_FINE=0
 * switches the kernel off (both entry points take the round-4 kernels).
 * live (may be NULL): a device-side row count - throughput mode's row total; rows >= clamp(*live - live_off, 0, batch) of both
 * descriptor sets are not processed (their output rows hold whatever the conversion of the inputs left there).
 * pats_attentional_propagation_packed_counted_f32: one packed layer over a capacity with such a count (honoured by the one-kernel
 * layers at the third and the fine level's shapes in eval mode; ignored otherwise). */
size_t pats_attentional_gnn_packed_workspace_bytes(int64_t batch, int C, int heads, int n);
int pats_attentional_gnn_packed_f32(const float* desc0, const float* desc1, int64_t batch, const int64_t* live, int64_t live_off,
                                    int C, int heads, int n, int layers,
                                    const pats_propagation_weights* const* weights, const void* const* packed,
                                    const int* cross, float bn_eps, float* out0, float* out1, void* workspace,
                                    size_t workspace_bytes, pats_stream_t stream);
int pats_attentional_propagation_packed_counted_f32(const float* x, const float* source, int64_t batch, const int64_t* live,
                                                    int64_t live_off, int C, int heads, int n, int m,
                                                    const pats_propagation_weights* w, const void* packed, int bn_train,
                                                    float bn_eps, const float* residual, float* out, void* workspace,
                                                    size_t workspace_bytes, pats_stream_t stream);

/* ---- the descriptor heads either side of the GNN: Conv1d(kernel_size=1) and BatchNorm1d + ReLU -------------------
 * Replaces nn.Conv1d(k=1) wherever the path uses it alone - `final_proj` right before the cost build
 * (models/first_layer.py:34-36,105; models/second_layer.py:40-42,91) - and, chained, the MLP of
 * models/modules.py:57-69 that KeypointEncoder is (:70-82; first_layer.py:30-31,81; third_layer.py:96-97,139-140):
 *   y[b][o][t] = bias[o] + sum_c w[o][c] * f(x[b][c][t]),   f(v) = v                              (in_scale == NULL)
 *                                                           f(v) = max(0, v * in_scale[c] + in_shift[c])  otherwise
 *   (+ residual[b][o][t] if residual != NULL)
 * i.e. the BatchNorm1d + ReLU that follows a Conv1d inside MLP is applied while the NEXT Conv1d stages its input:
 * eval mode  -> in_scale = gamma / sqrt(running_var + eps), in_shift = beta - running_mean * in_scale (host side);
 * train mode -> pats_bn_fold_f32 below computes them from the batch statistics of the previous layer's output.
 * x [batch,K,n], y [batch,M,n] channel-major like the reference's Conv1d; w_t = conv.weight[:, :, 0].t().contiguous()
 * ([K][M], row = input channel); bias [M] or NULL.  K % 8 == 0 (pad the two keypoint coordinates with zero channels).
 * Same contraction as pats_cost_f32.  Workspace: pats_conv1x1_workspace_bytes(). */
size_t pats_conv1x1_workspace_bytes(void);
int pats_conv1x1_f32(const float* w_t, const float* bias, const float* x, int64_t batch, int K, int M, int n,
                     const float* in_scale, const float* in_shift, const float* residual, float* y, void* workspace,
                     size_t workspace_bytes, pats_stream_t stream);
/* BatchNorm1d in TRAIN mode (F.batch_norm on batch statistics: per channel over (batch, n), biased variance), folded:
 * scale[c] = gamma[c] / sqrt(var_c + eps), shift[c] = beta[c] - mean_c * scale[c].  Deterministic (fixed summation
 * order, double accumulation).  PATS.eval() leaves the third layer in train mode (models/pats.py:112-120), so its
 * KeypointEncoder normalises with these. */
size_t pats_bn_fold_workspace_bytes(int C);
int pats_bn_fold_f32(const float* h, int64_t batch, int C, int n, const float* gamma, const float* beta, float eps,
                     float* scale, float* shift, void* workspace, size_t workspace_bytes, pats_stream_t stream);

/* ---- the scale head: target descriptors -> the OT problem's column marginals `ns` -------------------------------------
 * Replaces  scale = exp(sigmoid(proj(desc1[:, :, :h*w] as [b,C,h,w])) * ln256 - ln256 / 2)  with proj = nn.Conv2d(C, 1,
 * kernel_size=3, padding=1):  models/first_layer.py:39-40,106-107 (scalex_proj, 15x20 grid);  models/second_layer.py:33-36,
 * 92-98 (scalex_proj and scaley_proj on the 12x12 grid, scale = scale_x * scale_y: heads = 2);  models/third_layer.py:88-89,
 * 151-152 (scale_proj, 8x8).  x [batch,C,ld] is the tensor the cost build takes (ld = h*w, or h*w + 1 with the dustbin
 * feature column, which the heads skip); weight [heads][C][3][3] = the Conv2d weights concatenated over heads; bias
 * [heads]; out [batch][h*w] is `ns` as log_optimal_transport / log_optimal_transport2 take it; per_head (optional, NULL to
 * skip) [batch][heads][h*w] receives the heads on their own - scale_x and scale_y, which SecondLayer.est_position takes
 * separately (second_layer.py:116-117).  h*w <= 512. */
int pats_scale_head_f32(const float* x, int64_t batch, int C, int ld, int h, int w, const float* weight,
                        const float* bias, int heads, float* out, float* per_head, pats_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PATS_AMD_H */
