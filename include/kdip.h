/* libkdip_hip -- C ABI of the MI355X-native guided-diffusion inverse-problem hot path.
 *
 * The reference (xypeng9903/k-diffusion-inverse-problems) is pure Python: its plug-in
 * surface is duck typing + two decorator registries and it has no FFI.  This header is the
 * boundary a maintainer would bind (ctypes stub in INTEGRATION.md); every entry point
 * names the reference interface it replaces (file:line under the reference root).
 *
 * Conventions
 *   - every function returns an int status: 0 = OK, < 0 = error (enum below); the message of
 *     the last error on the calling thread is returned by kdip_last_error().  No exceptions,
 *     no aborts across the ABI.
 *   - pointers named *_dev are device pointers the CALLER owns (e.g. torch tensors'
 *     data_ptr()); the library never frees or retains them beyond the call.  Work is
 *     enqueued on `stream` (a hipStream_t passed as void*); the caller keeps buffers alive
 *     until the stream is synchronised.  Pointers named *_host are host memory and are
 *     consumed before the call returns.
 *   - image tensors are fp32, NCHW, contiguous: [B,3,H,W] (measurement of the SR operator
 *     [B,3,H/sf,W/sf]); B independent problems that share one sigma per call.
 *   - handles are not thread-safe; one per (process, device).
 */
#ifndef KDIP_H
#define KDIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum { KDIP_OK = 0, KDIP_ERR_ARG = -1, KDIP_ERR_HIP = -2, KDIP_ERR_STATE = -3, KDIP_ERR_NOMEM = -4,
       KDIP_ERR_UNSUPPORTED = -5 };
enum { KDIP_F32 = 0, KDIP_BF16 = 1,                   /* storage / MFMA input type of the UNet */
       KDIP_F16X3 = 3,      /* as KDIP_BF16X3 with an fp16 HEAD as well: a*b = f16(a)*f16(b) + f16(a)*f16(b - f16(b)) + f16(a - f16(a))*f16(b), three fp16
                             * MFMAs, 11 + 11-bit operands (operand error ~2^-22; two staged operand planes instead of three, no weight re-encode:
                             * 3.6 % faster end to end).  The head has fp16's exponent range only: an activation / gradient operand beyond +-65504
                             * after its power-of-two scaling SATURATES its products instead of degrading them.  Every such launch raises bit 0 of
                             * kdip_unet_x3_saturated, the handle carries the KDIP_BF16X3 weights as well, and kdip_unet_x3_head(u, 1) makes the
                             * next calls run on them: callers poll the flag per call and redo a flagged call bf16-headed (kdip_amd/unet.py does),
                             * so the fast arithmetic never decides a result outside its window.  Deterministic like KDIP_BF16X3. */
       KDIP_BF16X3 = 2 };   /* fp32 storage, split-precision convs: every product a*b as three 16-bit MFMAs into one fp32 accumulator --
                             * bf16(a)*bf16(b) + f16(a)*f16(b - bf16(b)) + f16(a - bf16(a))*f16(bf16(b)) (bf16 head, fp16 tails with
                             * power-of-two range scaling; csrc/conv.hip Mma<f32x3_t>): operand error ~2^-21 of the tensor's scale instead
                             * of bf16's 2^-9 -- the fast mode that meets the 1e-3 dB tolerance against the reference's fp32 arithmetic
                             * (utils_model.py:364 use_fp16=False).
                             * KDIP_F32 and KDIP_BF16X3 handles are DETERMINISTIC: every cross-block reduction (GroupNorm statistics,
                             * split-K partial sums) is made in a fixed order (csrc/det.h), two runs of one call are bitwise equal, as
                             * the reference's CPU path is.  KDIP_BF16 (throughput mode) uses floating-point atomics. */
enum { KDIP_OP_INPAINT = 0, KDIP_OP_BLUR = 1, KDIP_OP_SR = 2 };
enum { KDIP_OT_NONE = 0, KDIP_OT_DWT = 1, KDIP_OT_DCT = 2 };

const char* kdip_last_error(void);
int kdip_version(void);

/* ------------------------------------------------------------------ UNet (rows A4-A6, A14)
 * Replaces guided_diffusion.unet.UNetModel (guided_diffusion/unet.py:396-668) as built by
 * script_util.create_model (guided_diffusion/script_util.py:130-184) with the defaults of
 * condition/diffpir_utils/utils_model.py:353-387, and -- for kdip_unet_vjp -- the
 * torch.autograd.grad(..., x) calls of condition/condition.py:136,146,155,172. */
typedef struct kdip_unet kdip_unet;

/* attention_ds: downsample rates with attention (image_size // attention_resolution). */
int kdip_unet_create(int device, int dtype, int image_size, int in_channels, int model_channels, int out_channels,
                     int num_res_blocks, const int* attention_ds, int n_attention_ds, const int* channel_mult,
                     int n_channel_mult, int num_head_channels, kdip_unet** out);
void kdip_unet_destroy(kdip_unet* u);
/* One parameter of the reference state_dict, by its key (e.g. "input_blocks.3.0.in_layers.2.weight",
 * plus optional "out_cov.weight"/"out_cov.bias" of OpenAIDenoiserV2, k_diffusion/external.py:141);
 * fp32 host data in PyTorch layout.  Replaces nn.Module.load_state_dict
 * (sample_condition_openai.py:130-132). */
int kdip_unet_load(kdip_unet* u, const char* name, const float* data_host, const long* shape, int ndim);
/* Repack all weights into MFMA fragment order (forward + dgrad copies) and upload. */
int kdip_unet_finalize(kdip_unet* u);
/* out = UNet(x * in_scale, t).  x_dev [B,in_ch,S,S], t_dev [B] (fractional allowed),
 * out_dev [B,out_ch,S,S]; optional cov_out_dev [B,6,S,S] = out_cov(h) and feature_dev
 * [B,model_ch,S,S] = h (UNetModel.forward(..., return_feature=True), unet.py:665-666).
 * Activations needed by kdip_unet_vjp are stashed inside the handle. */
int kdip_unet_forward(kdip_unet* u, void* stream, const float* x_dev, const float* t_dev, int B, float in_scale,
                      float* out_dev, float* cov_out_dev, float* feature_dev);
/* gx = (d out / d (x*in_scale))^T cot for the last kdip_unet_forward.  cot_dev [B,out_ch,S,S] fp32,
 * gx_dev [B,in_ch,S,S] fp32; B must equal the batch of that forward (KDIP_ERR_ARG otherwise).  May be called repeatedly. */
int kdip_unet_vjp(kdip_unet* u, void* stream, const float* cot_dev, int B, float* gx_dev);
/* bytes of device workspace currently held for batch B (allocated lazily, grows monotonically). */
long kdip_unet_workspace_bytes(kdip_unet* u, int B);
/* Counter that changes whenever the handle re-allocates a workspace arena (a call at a batch whose plan does not fit): every
 * device pointer a captured hipGraph of an earlier call baked in is dangling after that -- re-capture (kdip_amd/graphs.py). */
long kdip_unet_workspace_generation(kdip_unet* u);
/* KDIP_BF16X3 handles: how the dgrad convs of kdip_unet_vjp place the fp16 window of their gradient operand (gradients have no natural
 * scale).  per_launch = 0 (default): one power-of-two scale per VJP from max |cotangent| -- right for networks whose backward gains
 * keep the gradient tensors of one VJP within +-4 decades of the cotangent (elements outside lose their fp16 tails: the error falls
 * back towards bf16's, never to inf).  per_launch = 1: every dgrad launch derives its scale from a sampled max of its own input (one
 * extra ~4 us launch each, +2 % per guided call): f32-grade whatever the network's backward gains. */
int kdip_unet_x3_window(kdip_unet* u, int per_launch);
/* (no reference counterpart) KDIP_BF16X3 handles: *flags_host = bit 0: since the last reset some conv launch staged an activation /
 * gradient operand outside the fp16 window of its tail planes (|a * 2^sa| > 65504: that product fell back towards bf16 accuracy);
 * bit 1: a weight of this handle was outside the window when it was packed (|w| > 255.9).  0 = every product of every conv carried its
 * full split precision.  KDIP_F16X3 handles also: bit 2: some conv launch's whole operand tensor (non-zero) sat below 2^-12 of its scale
 * reference, under the fp16 head's normal range (the low side of the window; found per launch, checked at the end of each forward / VJP);
 * bit 3: a weight tensor of the handle sits under that range, the handle runs bf16-headed throughout.  Bits 0 and 2 = "redo this call
 * with kdip_unet_x3_head(u, 1)".  Synchronises `stream`. */
int kdip_unet_x3_saturated(kdip_unet* u, void* stream, int reset, int* flags_host);
/* (no reference counterpart) KDIP_F16X3 handles: bf16_head = 1 -> the following kdip_unet_forward / kdip_unet_vjp / kdip_guided_call_v1 run in the
 * KDIP_BF16X3 arithmetic on the bf16-headed weights the handle carries (same workspace plan, same activation stash: a VJP may be redone
 * after a forward made in the other arithmetic); 0 -> back to the fp16-headed arithmetic (default).  Returns the previous setting or < 0. */
int kdip_unet_x3_head(kdip_unet* u, int bf16_head);
/* (no reference counterpart) A/B switch of the fixed-order reductions of a KDIP_F32 / KDIP_BF16X3 handle (default on; off = the
 * floating-point atomics of the KDIP_BF16 mode: measures what reproducibility costs).  Returns the previous setting (0 | 1) or < 0. */
int kdip_unet_deterministic(kdip_unet* u, int on);
/* Debug / test aid: 64-bit word sum of the activation stash kdip_unet_vjp reads (unchanged between a forward and its VJPs). */
int kdip_unet_debug_stash_checksum(kdip_unet* u, void* stream, unsigned long long* sum_host);

/* ------------------------------------------------- operators / solvers (rows A9-A13)
 * One context per measurement operator instance (condition/measurements.py:86-244). */
typedef struct kdip_op kdip_op;
int kdip_op_create(int device, int kind, int image_size, int scale_factor, float sigma_s, kdip_op** out);
void kdip_op_destroy(kdip_op* op);
/* PSF -> OTF on device: p2o + fft2 (condition/diffpir_utils/utils_sisr.py:22-41,79-96). */
int kdip_op_set_psf(kdip_op* op, const float* psf_host, int kh, int kw);
/* Optional rank-1 factors of the PSF (psf = krow (x) kcol): enables the LDS-staged separable
 * stencil for A / A^T instead of the FFT (same circular convolution, measurements.py:139-148). */
int kdip_op_set_separable(kdip_op* op, const float* krow_host, const float* kcol_host, int taps);
/* mask [3,S,S] in {0,1} (InpaintingOperator.mask, measurements.py:209). */
int kdip_op_set_mask(kdip_op* op, const float* mask_host);
/* OrthoTransform type for the posterior covariance basis (condition/utils.py:50-67). */
int kdip_op_set_ortho(kdip_op* op, int ortho_type);
/* copies the complex OTF [S,S] (float2) to otf_dev: operator.pre_calculated[0]. */
int kdip_op_get_otf(kdip_op* op, void* stream, float* otf_dev);
/* A x / A^T y of the solver model: circular blur (BLUR), blur then ::sf decimation (SR),
 * mask multiply (INPAINT).  SR adjoint = operator.transpose (measurements.py:114-120). */
int kdip_op_apply(kdip_op* op, void* stream, const float* x_dev, int B, int adjoint, float* out_dev);
/* mat = A^T (sigma_s^2 I + A C A^T)^-1 (y - A x0)   (condition/condition.py:317-439).
 * var_tensor_dev == NULL: C = var_scalar * I, closed form.  Otherwise C = W^-1 diag(var) W,
 * batched per-sample CG (scipy cg tol=1e-4, maxiter=1000 semantics); cg_iters_host/cg_info_host
 * (may be NULL) receive per-sample iteration counts and scipy-style info (0 = converged). */
int kdip_op_solve(kdip_op* op, void* stream, const float* y_dev, const float* x0_dev, float var_scalar,
                  const float* var_tensor_dev, int B, float* mat_dev, int* cg_iters_host, int* cg_info_host);
/* CG trip control of kdip_op_solve's tensor-variance branch.  trips = 0 (default): adaptive -- the per-sample convergence flags are
 * read back every second iteration (a host synchronisation; cg_iters / cg_info are filled).  trips > 0: exactly `trips` iterations,
 * no host read at all (capture-safe: samples freeze themselves on the device when they converge, surplus iterations change
 * nothing); cg_iters / cg_info come back as -1 and kdip_op_cg_unconverged tells afterwards whether `trips` was enough. */
int kdip_op_set_cg_fixed_trips(kdip_op* op, int trips);
/* *count_host = number of fixed-trip CG solves on this operator, since the last query, that stopped with an unconverged sample
 * (sticky device counter, read + cleared here; synchronises the stream).  Callers that replay captured guided calls check it once
 * per sampler run instead of reading flags every second CG iteration. */
int kdip_op_cg_unconverged(kdip_op* op, void* stream, int* count_host);
/* Bumped whenever the operator re-allocates its solver workspace (a call at a larger batch): hipGraphs captured before hold the old
 * buffers (same contract as kdip_unet_workspace_generation). */
long kdip_op_workspace_generation(kdip_op* op);
/* OrthoTransform.__call__ / .inv on [B,3,S,S] (condition/utils.py:59-67,88-139). */
int kdip_op_ortho(kdip_op* op, void* stream, const float* x_dev, int B, int inverse, float* out_dev);

/* ---- stand-alone operator kernels ---- */
/* out[b, j] = x[b].flat[idx[j]] and its adjoint (InpaintingOperator flatten paths, measurements.py:217-236). */
int kdip_gather(void* stream, const float* x_dev, const long* idx_dev, long nidx, long per_sample, int B, float* out_dev);
int kdip_scatter(void* stream, const float* y_dev, const long* idx_dev, long nidx, long per_sample, int B, float* out_dev);
/* out = x * mask (mask [3,S,S] broadcast over the batch). */
int kdip_mask_mul(void* stream, const float* x_dev, const float* mask_dev, int B, long chw, float* out_dev);
/* One axis of Resizer.forward (condition/dps_utils/resizer.py:55-74) and its adjoint:
 * weights/fov [n_out, taps]; axis 1 = W (x [planes, other, n_in]), axis 0 = H (x [planes, n_in, other]). */
int kdip_resize_axis(void* stream, const float* x_dev, const float* w_dev, const int* fov_dev, int taps, int n_in,
                     int n_out, int other, int axis, long planes, int adjoint, float* out_dev);
/* dense circular PSF convolution / correlation in the spatial domain (LDS-staged tile + halo). */
int kdip_blur_dense(void* stream, const float* x_dev, const float* psf_dev, int ks, int S, long planes, int adjoint,
                    float* out_dev);
/* raw 2-D FFT of [planes,S,S] (complex as interleaved float2): torch.fft.fft2 / ifft2. */
int kdip_fft2(void* stream, int S, const float* in_dev, int real_in, float* out_dev, int real_out, long planes,
              int inverse, float* tmp_dev);

/* ------------------------------------------------------ guidance algebra (rows A3-A8) */
/* p_mean_variance epilogue of ConditionOpenAIDenoiser.uncond_pred (condition/condition.py:231-248,
 * gaussian_diffusion.py:262-276,293-333).  tables7 = {c_in, sqrt(1/ac_t), sqrt(1/ac_t - 1), log beta_t,
 * log post_var_clipped_t, post_var_t, post_mean_coef1_t}.  var_dev may be NULL (scalar-variance branch). */
int kdip_x0_epilogue_v1(void* stream, const float* unet_out_dev, const float* x_dev, int B, long HW,
                        const float* tables7_host, float* x0_mean_dev, float* x0_raw_dev, float* var_dev);
/* ConditionOpenAIDenoiserV2.uncond_pred (condition/condition.py:287-300). */
int kdip_x0_epilogue_v2(void* stream, const float* unet_out_dev, const float* cov_out_dev, const float* x_dev, int B,
                        long HW, float sigma, int want_var, float* x0_mean_dev, float* x0_var_dev, float* theta_var_dev);
/* cotangent on the 6 UNet output channels for d<ghat, x0_mean>/dx. */
int kdip_vjp_cotangent_v1(void* stream, const float* ghat_dev, const float* x0_raw_dev, int B, long HW,
                          float sqrt_recipm1, float* cot6_dev, float* g_raw_dev);
int kdip_vjp_cotangent_v2(void* stream, const float* ghat_dev, int B, long HW, float* cot6_dev);
/* hat = clamp(x0 + coef * (a * g_direct + b * unet_vjp), -1, 1); either gradient may be NULL. */
int kdip_guidance_combine(void* stream, const float* x0_mean_dev, const float* g_direct_dev, float a,
                          const float* unet_vjp_dev, float b, float coef, long n, float* hat_dev);
/* out = a*x + b*y (y may be NULL); out = x*y; clamp to [-1,1]. */
int kdip_axpby(void* stream, const float* x_dev, float a, const float* y_dev, float b, long n, float* out_dev);
int kdip_mul(void* stream, const float* x_dev, const float* y_dev, long n, float* out_dev);
int kdip_clamp(void* stream, const float* x_dev, long n, float* out_dev);
/* DPS: out[b] = zeta * x[b] / ||r[b]||_2 (per-sample norm; condition/condition.py:140-148).
 * x [B, per_x], r [B, per_r]; norm_out_dev [B] fp32, tmp_dev [B] fp64 scratch. */
int kdip_dps_normalize(void* stream, const float* x_dev, long per_x, const float* r_dev, long per_r, float zeta, int B,
                       float* out_dev, float* norm_out_dev, double* tmp_dev);

/* ------------------------------------------------------------- sampler updates (row A2)
 * k_diffusion/sampling.py:46-48,118-135,159-184. */
int kdip_sampler_add_noise(void* stream, const float* x_dev, const float* eps_dev, float s, long n, float* out_dev);
int kdip_sampler_euler(void* stream, const float* x_dev, const float* denoised_dev, float sigma_hat, float dt, long n,
                       float* out_dev);
int kdip_sampler_heun(void* stream, const float* x_dev, const float* denoised_dev, const float* x2_dev,
                      const float* denoised2_dev, float sigma_hat, float sigma_next, float dt, long n, float* out_dev);

/* --------------------------------------------------------------------------- profiling
 * Opt-in per-launch timing of the implicit-GEMM conv kernels with HIP events recorded on the
 * launch stream (bench.py's roofline leg).  enable(1) clears and starts recording, enable(0)
 * stops; report() synchronises the recorded events and fills per-class totals: milliseconds,
 * algorithmic FLOPs (2*B*H*W*Cin*Cout*taps, un-padded), minimal HBM bytes, launch count. */
int kdip_profile_enable(int on);
int kdip_profile_num_classes(void);
const char* kdip_profile_class_name(int cls);
int kdip_profile_report(double* ms, double* flops, double* bytes, long* launches);
/* writes one CSV row per recorded launch (class, shape, algorithmic GFLOP / MB, microseconds). */
int kdip_profile_dump(const char* path);

/* out[b] (+)= mean_i ((pred - target)^2 * exp(-logvar) + logvar): one half of OpenAIDenoiserV2.loss (k_diffusion/external.py:145-159);
 * accumulate = 0 zeroes out[] first. */
int kdip_gauss_nll_mean(void* stream, const float* pred_dev, const float* target_dev, const float* logvar_dev, int B, long per,
                        int accumulate, float* out_dev);

/* ------------------------------------------------------------------ LPIPS
 * The perceptual metric of compute_metrics (sample_condition_openai.py:41-49,161: lpips.LPIPS(net='vgg')): a stand-alone conv
 * layer handle (the 13 VGG-16 convs of lpips/pretrained_networks.py:vgg16 run on the implicit-GEMM kernel), ReLU / 2x2 max
 * pool on fp32 NCHW planes, and the per-layer distance  out[b] += mean_px sum_c w_c (f0_c/|f0| - f1_c/|f1|)^2
 * (lpips.normalize_tensor, NetLinLayer, spatial_average).  Host side: kdip_amd/lpips.py. */
typedef struct kdip_conv kdip_conv;
int kdip_conv_create(int device, int dtype, const float* w_host /*[Cout][Cin][kh][kw]*/, const float* bias_host, int Cout, int Cin,
                     int ntaps /*9 | 1*/, kdip_conv** out);
void kdip_conv_destroy(kdip_conv* c);
long kdip_conv_workspace_bytes(kdip_conv* c, int B, int H, int W);
int kdip_conv_apply(kdip_conv* c, void* stream, const float* x_nchw_dev, int B, int H, int W, float* y_nchw_dev, void* workspace_dev);
int kdip_relu_maxpool(void* stream, const float* x_dev, long planes, int H, int W, int pool, float* y_dev);
int kdip_lpips_layer(void* stream, const float* f0_dev, const float* f1_dev, const float* lin_w_dev, int B, int C, long HW,
                     float* out_accum_dev /*[B], += */);

/* ------------------------------------------------------------- CU-masked streams
 * A HIP stream restricted to the compute units whose bits are set in mask_words (hipExtStreamCreateWithCUMask).  Two part-batches
 * of a GPU's batch on two such streams with disjoint masks share the chip by SPACE: one stream's persistent conv launches (every
 * CU's LDS full) no longer lock the other stream's small-map kernels out.  (include/kdip_internal.h: kdip_debug_cu_census reports where the
 * blocks of such a stream run.) */
int kdip_stream_create_cu_mask(int device, const unsigned* mask_words, int nwords, void** stream_out);
int kdip_stream_destroy(void* stream);

/* ------------------------------------------------------------- one guided call (kdip_guided_step)
 * ConditionOpenAIDenoiser._type_I_guidance_impl (condition/condition.py:167-174) with uncond_pred (:231-274) in ONE entry point:
 * UNet forward -> p_mean_variance epilogue -> mat-solver (closed form, or CG when tensor_var != 0) -> cotangent -> UNet VJP ->
 * hat = clamp(x0_mean + sigma^2 * c_in * (a_t * g_raw + unet_vjp), -1, 1).  x_dev [B,3,S,S], t_dev [B] (floored timestep),
 * y_dev = the measurement in the solver's layout, tables7_host as kdip_x0_epilogue_v1, var_scalar = the scalar x0 variance
 * (ignored when tensor_var: the learned per-pixel variance is used), ws_dev = kdip_guided_ws_floats(B, S) floats, hat_dev [B,3,S,S].
 * Capture-safe when the operator is in fixed-trip CG mode (or tensor_var == 0).  Leaves the UNet stash valid for kdip_unet_vjp. */
long kdip_guided_ws_floats(int B, int S);
/* Regions of that workspace after a call, as float offsets from ws_dev: offsets_host[KDIP_GWS_*] (count must be KDIP_GWS_COUNT).
 * x0_raw (the un-clamped x0 prediction) is what a stepwise continuation needs for another VJP through the clamp
 * (condition.py:231 `pred_xstart.clamp(-1, 1)`; tmpd :268-269, STSL :185-208). */
enum { KDIP_GWS_OUT6 = 0, KDIP_GWS_X0_MEAN, KDIP_GWS_X0_RAW, KDIP_GWS_VAR, KDIP_GWS_MAT, KDIP_GWS_COT, KDIP_GWS_G_RAW, KDIP_GWS_UG,
       KDIP_GWS_SCORE, KDIP_GWS_COUNT };
int kdip_guided_ws_layout(int B, int S, long* offsets_host, int count);
int kdip_guided_call_v1(kdip_unet* u, kdip_op* op, void* stream, const float* x_dev, const float* t_dev, const float* y_dev, int B,
                        const float* tables7_host, float sigma, float var_scalar, int tensor_var, float* ws_dev, float* hat_dev,
                        int* cg_iters_host, int* cg_info_host);

#ifdef __cplusplus
}
#endif
#endif /* KDIP_H */
