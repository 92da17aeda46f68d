/*
 * dimo_hip.h -- C ABI of libdimo_hip.so (gfx950 / MI355X).
 *
 * These entry points are what the reference's native-extension bindings for the
 * deform-then-render path would bind.  Each replaces one CUDA torch-extension
 * call site of Friedrich-M/DIMO (paths relative to the reference repo):
 *
 *   dimo_raster_*      diff_gauss.GaussianRasterizer (6 outputs)     renderer/latent_gs_renderer.py:13-16,1132-1147,1255-1266
 *                      diff_gaussian_rasterization.GaussianRasterizer renderer/latent_gs_renderer.py:9-12,1149-1163,1268-1277
 *   dimo_knn           knn_cuda.KNN(k, transpose_mode=True)          main_train_dimo.py:24,502-509
 *   dimo_dist2         simple_knn._C.distCUDA2                       renderer/latent_gs_renderer.py:17,426
 *   dimo_ssim_*        fused_ssim.fused_ssim / src.loss.ssim         main_test_dimo.py:29,979 ; src/loss.py:132-175
 *   dimo_deform_*      the LBS block of Renderer.render              renderer/latent_gs_renderer.py:1187-1219
 *   dimo_timenet_*     TimeNet.forward / its autograd backward       renderer/latent_gs_renderer.py:184-245 ; src/pos_enc.py:6-54
 *   dimo_image_loss    the loss assembly of GUI.train_step            main_train_dimo.py:325-380 ; src/loss.py:178-243
 *   dimo_flat_adam_step torch.optim.Adam over the 12 groups           renderer/latent_gs_renderer.py:460-476
 *   dimo_executor_*    the render loop of GUI.train_step (forward, mirrored backward)  main_train_dimo.py:276-318,415
 *   dimo_farthest_point_sample  pytorch3d.ops.sample_farthest_points   main_train_dimo.py:511-515
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the library never
 *     allocates or frees caller-visible memory and keeps no global state (re-entrant);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises
 *     unless stated;
 *   - matrices are row-major 4x4 in the row-vector convention MiniCam builds
 *     (renderer/latent_gs_renderer.py:960-969): p_clip = [x y z 1] @ projmatrix;
 *   - return value: 0 = ok, negative = DIMO_E_* (never aborts).
 */
#ifndef DIMO_HIP_H
#define DIMO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIMO_OK 0
#define DIMO_E_ARG (-1)      /* bad argument (null pointer, negative size, unsupported degree ...) */
#define DIMO_E_LAUNCH (-2)   /* hip launch / runtime error */
#define DIMO_E_WORKSPACE (-3)/* workspace too small */

#define DIMO_TILE 16         /* tile edge in pixels (BLOCK_X = BLOCK_Y of the published rasterizer) */
#define DIMO_NFEAT 7         /* blended per-Gaussian features: r g b depth nx ny nz */

/* library / build identification: returns "dimo_hip gfx950 <version>" */
const char *dimo_version(void);
/* text of the calling thread's most recent DIMO_E_LAUNCH (HIP error string + source location) */
const char *dimo_last_error(void);

/* Optional kernel timing (the library's only process-global state; off by default).
 * While enabled, every instrumented kernel group is bracketed by a hipEvent pair recorded on the
 * stream it is launched on.  dimo_timing_read synchronises the device and returns the summed
 * duration and launch count of one group: "preprocess_fwd" "scan" "emit" "sort" "ranges" "blend_fwd"
 * "blend_bwd" "preprocess_bwd" "knn" "dist2" "ssim_fwd" "ssim_bwd" "deform_fwd" "deform_bwd" "image_loss" "adam"
 * "timenet_fwd" "timenet_bwd" "place" (binning, dimo_amd/csrc/binning.hip: "scan" = offsets + level-1 entries,
 * "sort" = per-bucket LDS sort, "ranges" = level-2 count, "place" = level-2 fill; "emit" is no longer used).
 * dimo_timing_enable(1) clears earlier records; returns the previous state.  dimo_timing_select restricts the
 * instrumentation to a comma-separated list of groups (NULL or "": all) -- two event records per launch are not
 * free, a throughput run that only needs one kernel's duration selects that kernel. */
int dimo_timing_enable(int on);
int dimo_timing_select(const char *names);
int dimo_timing_read(const char *name, double *total_ms, int64_t *launches);

/* ------------------------------------------------------------------ rasterizer workspaces
 * geom : per-Gaussian state written by preprocess (splat records, tile rects, tiles_touched,
 *        offsets, flags, block sums).  Needed by backward.
 * bin  : tile-instance state (per-tile lists of Gaussian ids, tile ranges, level-1 lists, blend checkpoints).
 *        Sized for a CAPACITY R_cap >= R (number of (Gaussian, tile) instances).  Needed by backward.
 * img  : per-pixel state (final transmittance, n_contrib).  Needed by backward.
 * Every workspace and scratch pointer of this header must sit on a 256-byte boundary (hipMalloc and torch's allocator
 * give that): the kernels use 16- and 64-byte vector accesses at the layouts' offsets.
 */
size_t dimo_raster_geom_bytes(int N);
/* (bin: N = the Gaussians of the model the workspace serves -- the binning's (supertile, depth bin) bucket count, and
 * with it the size of the unsorted level-1 array's bucket regions, follows N; every call that takes the workspace takes
 * the same N) */
size_t dimo_raster_bin_bytes(int N, int64_t R_cap, int H, int W);
size_t dimo_raster_img_bytes(int H, int W);

/* Byte offsets of the inspectable sub-buffers (used by the parity tests; all 256-B aligned).
 * geom: [0] splat float[N][16] = (x y A B C opacity r g b depth nx ny nz pad pad pad)
 *       [1] rect  uint16[N][4] = (xmin ymin xmax ymax) in tiles     [2] tiles_touched uint32[N]
 *       [3] offsets uint32[N] (inclusive scan)                      [4] flags uint8[N] (bit c = SH channel c clamped)
 *       [5] total  uint32[4]  = (R, overflow flag, level-1 entries, 0)
 * bin : [0] vals_sorted uint32[R_cap] (Gaussian ids)   [1] ranges uint32[T][2]   [2] instances per tile uint32[T]
 *       The published 64-bit sort key of an instance is (tile << 32 | fp32 depth bits).  Neither half is stored per
 *       instance: an instance's tile is the list it sits in (tile t owns slots [ranges[t][0], ranges[t][1])) and its
 *       depth bits are its Gaussian's (dimo_raster_depth_keys gathers them for inspection).
 *       (the instances are placed straight into their sorted slots: there is no unsorted emission to look at)
 * img : [0] final_T float[H*W]          [1] n_contrib uint32[H*W]
 */
int dimo_raster_geom_layout(int N, size_t out_offsets[6]);
int dimo_raster_bin_layout(int64_t R_cap, int H, int W, size_t out_offsets[3]);
int dimo_raster_img_layout(int H, int W, size_t out_offsets[2]);
/* Inspection (parity tests): the fp32 depth bits of every instance in list order -- the low half of the published
 * sort keys -- gathered from the instances' Gaussians into out uint32[R_cap] (the first min(R, R_cap) words). */
int dimo_raster_depth_keys(int N, int H, int W, int64_t R_cap, const void *geom, const void *bin, uint32_t *out,
                           void *stream);
/* Test entry (tests/test_gpu_binning_fuzz.py): the tile binning ALONE on a geometry workspace the caller filled as
 * the projection kernel would (tile rectangles, tiles touched, depth bits, the per-block words; bucket words zero) --
 * dimo_debug_bin_geom_layout: out = byte offsets of (rect, tiles, offsets, total, block_sums, key32, bucket words),
 * the workspace's bytes, the number of 256-Gaussian blocks, the supertile edge's log2 for an H x W image. */
int dimo_debug_bin_geom_layout(int N, int H, int W, size_t out[10]);
int dimo_debug_bin_instances(int N, int H, int W, int64_t R_cap, void *geom, void *bin, void *stream);
/* Diagnostic: per-workgroup phase trace of the binning kernels (tools/bin_trace.py; needs a library built with
 * DIMO_BIN_TRACE=1, else a non-NULL buffer is refused).  buffer = device memory for `capacity` records of 32 x u64,
 * NULL = off; returns the number of records written since the last call. */
int64_t dimo_debug_bin_trace(void *buffer, int64_t capacity);

/*
 * Stage 1 (per Gaussian): cull, project, 3D->2D covariance, conic, radius, tile rect, SH->RGB,
 * depth, view-space normal, tiles_touched + its inclusive scan.
 *   means3D[N,3] shs[N,M,3]|NULL colors_precomp[N,3]|NULL opacities[N] scales[N,3]|NULL rotations[N,4]|NULL
 *   cov3D_precomp[N,6]|NULL (exactly one of shs/colors_precomp; scales+rotations or cov3D_precomp)
 *   viewmatrix[16] projmatrix[16] campos[3]: device pointers.
 *   radii[N] int32 out.
 *   R_host: if non-NULL the call synchronises `stream` and stores R (the exact instance count) there,
 *           so the caller can size the bin workspace exactly; if NULL nothing synchronises and the
 *           caller must provide R_cap from a bound (overflow is flagged in geom total[1], never written OOB).
 */
int dimo_raster_preprocess_forward(int N, int sh_degree, int M, int H, int W, const float *means3D, const float *shs,
                                   const float *colors_precomp, const float *opacities, const float *scales,
                                   const float *rotations, const float *cov3D_precomp, float scale_modifier,
                                   const float *viewmatrix, const float *projmatrix, const float *campos,
                                   float tanfovx, float tanfovy, int32_t *radii, void *geom, size_t geom_bytes,
                                   int64_t *R_host, void *stream);

/*
 * Stage 2+3: per-tile instance lists in the order of a stable radix sort on (tile | fp32 depth bits) -- depth sort
 * of the Gaussians, then placement of the instances by counting (dimo_amd/csrc/binning.hip) --, tile ranges, and
 * front-to-back alpha compositing.  At most 32768 tiles.
 *   bg[3] device pointer.  out_color[3,H,W] out_depth[1,H,W] out_normal[3,H,W]|NULL out_alpha[1,H,W].
 *   out_normal == NULL selects the 4-output (diff_gaussian_rasterization) flavour.
 */
int dimo_raster_render_forward(int N, int H, int W, int64_t R_cap, const float *bg, const void *geom, void *bin,
                               size_t bin_bytes, void *img, size_t img_bytes, float *out_color, float *out_depth,
                               float *out_normal, float *out_alpha, void *stream);

/*
 * Backward of stages 3 and 1.  dL_d* image gradients may be NULL (treated as zero).
 * Outputs (all written, not accumulated): dL_dmeans3D[N,3] dL_dmeans2D[N,3] (NDC units, z = 0)
 * dL_dshs[N,M,3]|NULL dL_dcolors[N,3]|NULL dL_dopacity[N] dL_dscales[N,3]|NULL dL_drotations[N,4]|NULL
 * dL_dcov3D[N,6]|NULL.
 * scratch: dimo_raster_backward_scratch_bytes(N, R_cap) bytes.
 */
size_t dimo_raster_backward_scratch_bytes(int N, int64_t R_cap);
int dimo_raster_backward(int N, int sh_degree, int M, int H, int W, int64_t R_cap, const float *means3D,
                         const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                         const float *rotations, const float *cov3D_precomp, float scale_modifier,
                         const float *viewmatrix, const float *projmatrix, const float *campos, const float *bg,
                         float tanfovx, float tanfovy, const int32_t *radii, const void *geom, const void *bin,
                         const void *img, const float *dL_dcolor, const float *dL_ddepth, const float *dL_dnormal,
                         const float *dL_dalpha, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dshs,
                         float *dL_dcolors, float *dL_dopacity, float *dL_dscales, float *dL_drotations,
                         float *dL_dcov3D, void *scratch, size_t scratch_bytes, void *stream);

/* ------------------------------------------------------------------ nearest neighbours
 * dimo_knn: brute-force k nearest reference points (k <= 16) per query; NON-squared distances,
 * ascending; ties -> lowest reference index.  ref[M,3] query[N,3] dist[N,k] idx[N,k] (int64). */
int dimo_knn(int M, int N, int k, const float *ref, const float *query, float *dist, int64_t *idx, void *stream);
/* The same with SEEDS (k = 4 only; ignored otherwise): seed_idx[N,4] (int64, may be NULL, may alias idx) holds four
 * candidate neighbours per query, typically the previous training step's result.  They only prune the search (the
 * largest seed distance bounds the fourth-nearest distance from above); the output is identical to dimo_knn's for
 * ANY seed values -- a repeated or out-of-range index just switches the pruning off for that query. */
int dimo_knn_seeded(int M, int N, int k, const float *ref, const float *query, float *dist, int64_t *idx,
                    const int64_t *seed_idx, void *stream);

/* dimo_dist2: mean squared distance of each point to its 3 nearest other points. points[N,3] out[N]. */
int dimo_dist2(int N, const float *points, float *out, void *stream);
/* The same result (bit for bit) on a uniform grid, O(N) for reasonably spread points (the brute-force call is O(N^2):
 * 5 ms at 1e5 points, 0.5 s at 1e6).  workspace: dimo_dist2_workspace_bytes(N) bytes of device memory. */
size_t dimo_dist2_workspace_bytes(int N);
int dimo_dist2_grid(int N, const float *points, float *out, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ fused SSIM (11x11, sigma 1.5, zero padding)
 * img1,img2 [B,C,H,W]; ssim_sum: one float accumulator (zeroed by the call) receiving the SUM of the
 * SSIM map (mean = sum / (B*C*H*W)); partials [3,B,C,H,W] saved for backward (dm/dmu1, dm/dsigma1_sq,
 * dm/dsigma12 at every pixel).  Backward: dL_dimg1 = dL_dmean/(B*C*H*W) * conv^T(partials).
 * clamp_img1 != 0: img1 is read as clamp(img1, 0, 1) (the rasterizer's raw colour output; the gradient
 * returned is then w.r.t. the CLAMPED image and the caller applies the clamp mask). */
int dimo_ssim_forward(int B, int C, int H, int W, int clamp_img1, const float *img1, const float *img2,
                      float *ssim_sum, float *partials, void *stream);
int dimo_ssim_backward(int B, int C, int H, int W, int clamp_img1, const float *img1, const float *img2,
                       const float *partials, const float *dL_dmean /* 1 float, device */, float *dL_dimg1,
                       void *stream);
/* Value and gradient in one launch (the training loss always needs both, main_train_dimo.py:343 + :415): ssim_sum as
 * dimo_ssim_forward, dL_dimg1 as dimo_ssim_backward with upstream dL_dmean -- the derivative planes never leave the
 * chip (each workgroup recomputes them on its tile's halo).  clamp_img1 bit 1 (value 2): *ssim_sum is already zero
 * (the caller cleared it with its other accumulators), skip the memset. */
int dimo_ssim_forward_backward(int B, int C, int H, int W, int clamp_img1, const float *img1, const float *img2,
                               const float *dL_dmean /* 1 float, device */, float *ssim_sum, float *dL_dimg1,
                               void *stream);
/* the same with img2 as B separate [C,H,W] tensors (HOST array of B <= 32 device pointers): the targets of a batch
 * need not be stacked */
int dimo_ssim_forward_backward_images(int B, int C, int H, int W, int clamp_img1, const float *img1,
                                      const float *const *img2_images_host, const float *dL_dmean, float *ssim_sum,
                                      float *dL_dimg1, void *stream);

/* ------------------------------------------------------------------ fused image losses + their gradients
 * One motion's batch of B <= 64 renders (main_train_dimo.py:331-372, src/loss.py:64-106):
 *   loss += sum_b w_mse[b] |clamp(image_b,0,1) - gt_b|^2  + w_mask |alpha - mask|^2
 *         + w_smooth_{x,y} |d depth| exp(-mean_c |d rgb|)  + w_bilat_{x,y} sqrt(1 + (|d n| exp(-3 mean_c |d rgb|))^2)
 * (the caller folds lambda, the 1/numel of each mean and the data-parallel share into the weights), and
 * g_image/g_depth/g_normal/g_alpha receive dloss/d(raw rasterizer outputs); ssim_grad (optional, [B,3,H,W],
 * w.r.t. the clamped image) is added before the clamp mask.  image[B,3,H,W] depth[B,1,H,W]|NULL
 * normal[B,3,H,W]|NULL alpha[B,1,H,W] gt[B,3,H,W] mask[B,1,H,W] (mask_per_image != 0) or [1,H,W] shared.
 * w_mse_host: B floats on the HOST (passed by value to the kernel).  loss_accum: DIMO_LOSS_WORDS device floats, added
 * to; the loss is their SUM (the workgroups spread their atomic adds over 16 cache lines: same-address atomics
 * serialise in the L2 at ~7 ns each, which was a fifth of this kernel's time).
 * g_dot (optional, [B,1,H,W]): per pixel sum over the channels of gradient x rendered value -- the rasterizer backward's
 * "S" (dimo_render_desc.g_dot): with it the blend backward reads 4 bytes per pixel instead of the nine final
 * accumulator planes.  gt_images_host / mask_images_host (optional HOST arrays of B device pointers, [3,H,W] /
 * [1,H,W] each): the batch's targets / masks as separate tensors instead of the contiguous gt / mask (which may then
 * be NULL). */
#define DIMO_LOSS_WORDS 512
int dimo_image_loss(int B, int H, int W, const float *image, const float *depth, const float *normal,
                    const float *alpha, const float *gt, const float *mask, int mask_per_image,
                    const float *w_mse_host, float w_mask, float w_smooth_x, float w_smooth_y, float w_bilat_x,
                    float w_bilat_y, const float *ssim_grad, float *loss_accum, float *g_image, float *g_depth,
                    float *g_normal, float *g_alpha, float *g_dot, const float *const *gt_images_host,
                    const float *const *mask_images_host, void *stream);

/* SSIM + the image losses above in ONE tile pass (csrc/ssim.hip: ssim_loss_tile_kernel): what
 * dimo_ssim_forward_backward_images(B, 3, H, W, clamp | prezeroed, image, gt images, ssim_coef, ssim_sum, ssim_grad)
 * followed by dimo_image_loss(..., ssim_grad, ...) computes -- main_train_dimo.py:331-372 with src/loss.py:64-106,
 * 132-175 -- without the SSIM gradient image in between.  ssim_coef: ONE device float, dL/d(mean SSIM) (the trainer
 * passes -lambda_ssim x share); *ssim_sum (device, zeroed by the caller) receives the sum of the SSIM map over the
 * B x 3 planes.  Everything else as dimo_image_loss. */
int dimo_ssim_image_loss(int B, int H, int W, const float *image, const float *depth, const float *normal,
                         const float *alpha, const float *gt, const float *mask, int mask_per_image,
                         const float *w_mse_host, float w_mask, float w_smooth_x, float w_smooth_y, float w_bilat_x,
                         float w_bilat_y, const float *ssim_coef, float *ssim_sum, float *loss_accum, float *g_image,
                         float *g_depth, float *g_normal, float *g_alpha, float *g_dot,
                         const float *const *gt_images_host, const float *const *mask_images_host, void *stream);

/* ------------------------------------------------------------------ fused skinning (stage s2 of Renderer.render)
 * One kernel for renderer/latent_gs_renderer.py:1187-1219: LBS weights w_k = L1norm(exp(-d_k^2/(2 r_k^2)) + 1e-7),
 * out_xyz = sum_k w_k (R(dq_k/|dq_k|)(x - c_k) + c_k + dc_k)   (local_frame != 0; else x + sum_k w_k dc_k),
 * out_rot = normalize(quat_mul(sum_k w_k dq_k, rotation)), out_scales = exp(scaling), out_opacity = sigmoid(opacity).
 *   xyz[N,3] rotation[N,4] scaling[N,3] opacity[N]  : raw canonical-Gaussian parameters
 *   c_xyz[M,3] c_log_radius[M] (= _c_radius)        : control points;  d_xyz[M,3] d_rot[M,4] : TimeNet outputs
 *   nn_dist[N,4] nn_idx[N,4] (int64)                : dimo_knn results (k = 4)
 * M <= dimo_deform_max_ctrl_points() (the control-point table and its gradient accumulators live in LDS).
 * Backward writes (does not accumulate) all eight gradients; control-point gradients are reduced without
 * global atomics in a fixed order (deterministic). scratch: dimo_deform_backward_scratch_bytes(N, M). */
int dimo_deform_max_ctrl_points(void);
size_t dimo_deform_backward_scratch_bytes(int N, int M);
int dimo_deform_forward(int N, int M, int local_frame, const float *xyz, const float *rotation, const float *scaling,
                        const float *opacity, const float *c_xyz, const float *c_log_radius, const float *d_xyz,
                        const float *d_rot, const float *nn_dist, const int64_t *nn_idx, float *out_xyz,
                        float *out_rot, float *out_scales, float *out_opacity, void *stream);
/* accumulate != 0: all eight gradient outputs are added to (a training step sums its renders' gradients
 * directly in the flat gradient bucket) instead of overwritten. */
int dimo_deform_backward(int N, int M, int local_frame, int accumulate, const float *xyz, const float *rotation,
                         const float *scaling, const float *opacity, const float *c_xyz, const float *c_log_radius,
                         const float *d_xyz, const float *d_rot, const float *nn_dist, const int64_t *nn_idx,
                         const float *g_out_xyz, const float *g_out_rot, const float *g_out_scales,
                         const float *g_out_opacity, float *dL_dxyz, float *dL_drotation, float *dL_dscaling,
                         float *dL_dopacity, float *dL_dc_xyz, float *dL_dc_log_radius, float *dL_dd_xyz,
                         float *dL_dd_rot, void *scratch, size_t scratch_bytes, void *stream);

/* ------------------------------------------------------------------ Adam over the flat parameter bucket
 * torch.optim.Adam semantics (no weight decay / amsgrad; renderer/latent_gs_renderer.py:475: eps = 1e-15) for all
 * parameter groups in one launch.  params/grads/exp_avg/exp_avg_sq: n floats each, 16-byte aligned; the groups are
 * n_segments contiguous ranges [segment_end[k-1], segment_end[k]) with learning rate segment_lr[k] (HOST arrays,
 * passed by value, n_segments <= 32, segment_end[n_segments-1] == n).  step = 1-based step count (bias correction).
 * skip_flags: optional device ints (n_flags of them, flag_stride ints apart); if any is non-zero the update is a
 * no-op (used with the rasterizer's capacity-overflow word).  zero_grad != 0 clears grads in the same pass.
 * skipped_launches: optional 2 device ints, zero-initialised by the caller and owned by this function afterwards:
 * the number of no-op launches so far, kept on the device so that the bias corrections use the number of updates
 * actually applied (step - skipped) without a host read-back; `step` must then count EVERY launch (1, 2, 3, ...).
 * report_src / report_words / report_dst_host / report_seq: optional -- the launch copies report_words device words
 * (the step's per-render (R, overflow) instance counts) to report_dst_host[1 ..] and then stores report_seq to
 * report_dst_host[0] (system-scope release); report_dst_host is HOST-VISIBLE pinned memory (hipHostMalloc) of
 * >= 1 + report_words words that the host polls a step later instead of a device-to-host copy + event.
 * zero_extra / zero_n: optional device floats cleared by the same launch (the next step's accumulators).
 * range_begin / range_end / final_part: ONE optimizer step taken as two launches over [0, split) and [split, n)
 * (offsets in floats, multiples of 4; 0, 0 = the whole bucket): both are given the same `step`; the launch with
 * final_part != 0 is the one that counts the step in skipped_launches, writes the report and clears zero_extra
 * (it must be the LAST of the two to run). */
int dimo_flat_adam_step(int64_t n, float *params, float *grads, float *exp_avg, float *exp_avg_sq, int n_segments,
                        const int64_t *segment_end_host, const float *segment_lr_host, float beta1, float beta2,
                        float eps, int64_t step, const int *skip_flags, int n_flags, int flag_stride, int zero_grad,
                        int *skipped_launches, const uint32_t *report_src, int report_words,
                        uint32_t *report_dst_host, uint32_t report_seq, float *zero_extra, int64_t zero_n,
                        int64_t range_begin, int64_t range_end, int final_part, void *stream);

/* Diagnostic (no reference counterpart): the 64-lane x 16-value wave reduction the rasterizer backward uses
 * (csrc/wave_ops.hpp), run on caller data.  in: 64 x 16 floats (lane-major), out: 3 x 16 floats = the column sums by
 * the 16-value form, then by the 13-value and the 10-value forms (their columns 13..15 / 10..15 are unspecified). */
int dimo_selftest_wave_reduce16(const float *in, float *out, void *stream);
/* Diagnostic: per-work-item trace of the blend backward.  buffer: device memory for `capacity` records of 4 x uint64
 * (s_memrealtime = 100 MHz ticks at start, at end, XCC << 56 | render << 48 | item code, records << 48 | quadrant visits << 32 | HW_ID), or NULL to
 * switch it off.  Synchronises the device; returns the number of records written since the previous call, or
 * DIMO_E_ARG when the library was built without -DDIMO_BWD_TRACE (the default). */
int64_t dimo_debug_blend_trace(void *buffer, int64_t capacity);

/* ------------------------------------------------------------------ TimeNet (the deformation MLP)
 * renderer/latent_gs_renderer.py:184-245 (`TimeNet.forward` with t_apply: one time per batch entry) for a whole
 * step's batch of P (motion, frame) pairs x M control points, forward and backward as fp32 MFMA GEMM chains.
 * Input row (p, m) = [posenc(c_xyz[m]; pts_freqs) | posenc(time[p]; time_freqs) | latent_table[latent_rows[p]]]
 * (src/pos_enc.py:6-54 ordering: per frequency sin(all dims), cos(all dims)); deformnet = D Linear+ReLU layers of
 * width W, the output of layer `skip` is concatenated BEHIND the input row (skip < 0: no skip); heads
 * pts_layers / rot_layers = Linear(W,W)+ReLU+Linear(W,3|4).
 * weight[l]/bias[l] (torch.nn.Linear layout [out,in] row-major), l = 0..D-1: deformnet.l; D: pts_layers.0;
 * D+1: pts_layers.2; D+2: rot_layers.0; D+3: rot_layers.2.  g_weight/g_bias: gradient buffers of the same shapes,
 * ADDED to (hardware fp32 atomics: summation order is not fixed run to run).
 * times_host[P], latent_rows_host[P] (NULL: row p) are HOST arrays (passed by value to the kernels), P <=
 * DIMO_TIMENET_MAX_PAIRS.  workspace (dimo_timenet_workspace_bytes) carries the activations from forward to
 * backward.  backward: g_d_xyz[P,M,3] g_d_rot[P,M,4] in; g_c_xyz[M,3] (NULL ok) and g_latent_table (NULL ok) are
 * ADDED to. */
#define DIMO_TIMENET_MAX_LAYERS 20
#define DIMO_TIMENET_MAX_PAIRS 256
typedef struct {
  int D, W, skip, pts_freqs, time_freqs, latent_dim;
  const float *weight[DIMO_TIMENET_MAX_LAYERS], *bias[DIMO_TIMENET_MAX_LAYERS];
  float *g_weight[DIMO_TIMENET_MAX_LAYERS], *g_bias[DIMO_TIMENET_MAX_LAYERS];
} dimo_timenet_desc;
size_t dimo_timenet_workspace_bytes(const dimo_timenet_desc *net, int P, int M);
int dimo_timenet_forward(const dimo_timenet_desc *net, int P, int M, const float *c_xyz, const float *times_host,
                         const float *latent_table, const int *latent_rows_host, float *d_xyz, float *d_rot,
                         void *workspace, size_t workspace_bytes, void *stream);
int dimo_timenet_backward(const dimo_timenet_desc *net, int P, int M, const float *g_d_xyz, const float *g_d_rot,
                          const float *times_host, const int *latent_rows_host, float *g_c_xyz,
                          float *g_latent_table, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ farthest point sampling (stage s1, every FPS_iter)
 * pytorch3d.ops.sample_farthest_points(points[1,N,3], K) with the default fixed start (main_train_dimo.py:511-515):
 * out_idx[0] = 0, then K-1 times the point with the largest squared distance to the selected set (lowest index among
 * equals).  min_dist_scratch: N floats.  One workgroup; K sequential rounds. */
int dimo_farthest_point_sample(int N, int K, const float *xyz, float *min_dist_scratch, int64_t *out_idx,
                               void *stream);

/* ------------------------------------------------------------------ native step executor
 * Runs the per-render kernel chains of one training step (main_train_dimo.py:276-318 forward, the mirrored
 * backward of :415) from one host call each: render i runs on private stream i % n_streams, joined with events
 * against `main_stream`.  Every pointer is caller-owned, persistent device memory; nothing is allocated.
 *   forward : dimo_deform_forward -> dimo_raster_preprocess_forward (no host read-back) -> dimo_raster_render_forward
 *   backward: dimo_raster_backward on the render's stream; then, ordered on main_stream, g_f_dc += g_shs and
 *             dimo_deform_backward(accumulate = 1) into the shared gradient views of dimo_step_common.
 * Degree-0 colour (f_dc [N,1,3]) and scale/rotation covariance only -- DIMO's training configuration. */
#define DIMO_EXECUTOR_TYPES 1
typedef struct {
  int N, M, H, W, with_normal, local_frame;
  int64_t R_cap;
  const float *xyz, *rotation, *scaling, *opacity, *f_dc;
  const float *c_xyz, *c_log_radius;
  const float *nn_dist;
  const int64_t *nn_idx;
  const float *bg;
  float scale_modifier;
  float *g_xyz, *g_rotation, *g_scaling, *g_opacity, *g_f_dc, *g_c_xyz, *g_c_log_radius;
  void *lbs_scratch;
  size_t lbs_scratch_bytes, geom_bytes, bin_bytes, img_bytes, bwd_scratch_bytes;
  /* stage s1 (renderer/latent_gs_renderer.py:1176-1177,1211-1212): the TimeNet moves every Gaussian itself --
   * d_xyz of a render is [N,3] (d_rot unused), pts = xyz + d_xyz, rotation = normalize(rotation), every scale =
   * exp(log_r[0]) (the shared radius `_r`, get_scaling :341-351); no control points, no KNN.  0 = stage s2 skinning */
  int stage1;
  const float *log_r;
  float *g_log_r;
} dimo_step_common;

typedef struct {
  const float *view, *proj, *campos;
  float tanfovx, tanfovy;
  const float *d_xyz, *d_rot;   /* TimeNet outputs of this render [M,3] [M,4] */
  float *g_d_xyz, *g_d_rot;     /* their gradient rows (accumulated) */
  float *out_color, *out_depth, *out_normal, *out_alpha;
  const float *g_color, *g_depth, *g_normal, *g_alpha;
  float *pts, *rot, *scales, *opac;   /* per-slot persistent workspaces from here on */
  int32_t *radii;
  void *geom, *bin, *img, *bwd_scratch;
  float *g_means3D, *g_means2D, *g_shs, *g_opac, *g_scales, *g_rot;
  /* optional [H,W]: sum over the output channels of (gradient image x rendered image) per pixel, as dimo_image_loss
   * emits it; NULL: the blend backward forms it from the forward's final accumulators */
  const float *g_dot;
  /* optional: two words that receive this render's (instance count R, overflow flag) when its tile lists are complete
   * -- the renders of a step can point into ONE array that the capacity policy and the optimizer's skip flag read
   * without gathering the per-render workspaces' words */
  uint32_t *totals_out;
} dimo_render_desc;

/* n_streams > 0: per-render chains on that many private streams (render i on stream i % n).
 * n_streams == 0: batched -- every stage is ONE launch over all renders of the call (blockIdx.y = render), on the
 *                 caller's stream.
 * n_streams < 0: batched ranges -- each [first, first+count) range is one batch; ranges go round-robin over
 *                |n_streams| private streams (one motion's renders per range: its losses overlap the other
 *                motions' rendering). */
void *dimo_executor_create(int n_streams);
void dimo_executor_destroy(void *executor);
/* forward chains of renders [0, n_renders) on the private streams (after everything enqueued on main_stream so
 * far); main_stream does NOT wait -- dimo_executor_join makes it wait for a sub-range, so the losses of one motion
 * can run while the other motions are still rendering */
int dimo_executor_forward(void *executor, const dimo_step_common *common, int n_renders,
                          const dimo_render_desc *renders, void *main_stream);
/* the same for renders [first, first+count) only (`renders` is still the whole array) */
int dimo_executor_forward_range(void *executor, const dimo_step_common *common, int first, int count,
                                const dimo_render_desc *renders, void *main_stream);
int dimo_executor_join(void *executor, int first, int count, void *main_stream);
/* rasterizer backward of renders [first, first+count) on their streams (after main_stream's current tail) ... */
int dimo_executor_backward_launch(void *executor, const dimo_step_common *common, int first, int count,
                                  const dimo_render_desc *renders, void *main_stream);
/* batched ranges (n_streams < 0): the private stream of the range that starts at render `first` (NULL if none), so
 * that the caller can enqueue the range's loss kernels behind its forward without a cross-stream join, and the
 * rasterizer backward continuing on that stream (no fork from a caller stream) */
void *dimo_executor_range_stream(void *executor, int first);
/* Batched ranges only: the rasterizer backward of every range inside [first, first + count) in launches of up to 8
 * renders on the CALLER's stream, ordered behind what the ranges' private streams hold at the time of the call. */
int dimo_executor_backward_launch_joint(void *executor, const dimo_step_common *common, int first, int count,
                                        const dimo_render_desc *descs, void *main_stream);
/* Batched ranges only: the rasterizer backward of the range that starts at `first`, IN ORDER on the stream that ran
 * its forward chain (and its loss kernels), its private stream.  Per-motion backward: one motion's backward overlaps the other motion's
 * loss kernels (each motion of main_train_dimo.py:276-318 is one range). */
int dimo_executor_backward_launch_in_order(void *executor, const dimo_step_common *common, int first, int count,
                                           const dimo_render_desc *renders, void *main_stream);
/* Batched ranges only.  Side work: the caller enqueues something ITSELF on private stream `which` (returned as a
 * hipStream_t, made to wait for main_stream's tail), concurrently with what it goes on enqueueing on main_stream, and
 * then calls dimo_executor_side_done(which); every forward chain started afterwards waits for it (a chain on stream
 * `which` follows in order).  Used for the step's KNN (main_train_dimo.py:257-258) next to the TimeNet forward. */
void *dimo_executor_side_stream(void *executor, int which, void *main_stream);
int dimo_executor_side_done(void *executor, int which);
/* main_stream waits for the side work last marked on private stream `which`; the private stream itself (no dependency
 * added), for work that CONTINUES what it holds -- the fold of the skinning backward and the optimizer's update of the
 * per-Gaussian parameters behind one motion's chain, next to the TimeNet backward on the caller's stream -- and the
 * waits of dimo_executor_backward_accumulate without its kernels, for a stream that only needs the ranges' results. */
int dimo_executor_wait_side(void *executor, int which, void *main_stream);
void *dimo_executor_private_stream(void *executor, int which);
int dimo_executor_join_ranges(void *executor, int first, int count, void *stream);
/* Batched ranges only, after dimo_executor_backward_launch_in_order of the same range: the range's skinning backward
 * on the same stream: per-Gaussian gradients in place in the deformation groups' leader buffers; what it adds to
 * SHARED words -- the control-point sums into g_c_xyz / g_c_log_radius, the TimeNet-row gradients -- it adds atomically
 * (round 6: no partial tables, no staging, lbs_scratch is unused by this path).
 * dimo_executor_backward_accumulate over the step's renders then only folds those into the gradient views (one
 * launch, fixed order), instead of running every motion's skinning backward on main_stream one after the other.
 * (The LBS block of latent_gs_renderer.py:1191-1219, backward, per motion of main_train_dimo.py:276-318.) */
int dimo_executor_backward_skinning_in_order(void *executor, const dimo_step_common *common, int first, int count,
                                             const dimo_render_desc *renders, void *main_stream);
/* ... and, on main_stream, per render: wait for it, g_f_dc += g_shs, skinning backward (accumulate) */
int dimo_executor_backward_accumulate(void *executor, const dimo_step_common *common, int first, int count,
                                      const dimo_render_desc *renders, void *main_stream);

#ifdef __cplusplus
}
#endif
#endif /* DIMO_HIP_H */
