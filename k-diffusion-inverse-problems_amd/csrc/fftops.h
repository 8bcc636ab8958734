// Internal launch API: FFT, frequency-domain algebra, operators, transforms, guidance, sampler, CG.
#pragma once
#include "common.h"

namespace kdip {

// ---- fft.hip ----------------------------------------------------------------------------
void make_twiddles256(float2* host);   // 128 entries exp(-2 pi i k / 256)
int fft2(hipStream_t st, const float2* tw256, int N, const void* in, int real_in, float2* tmp, void* out, int real_out,
         long planes, int inverse);
int cmul_otf(hipStream_t st, float2* X, const float2* FB, long nn, long planes, int conj);
int otf_solve(hipStream_t st, float2* R, const float2* FB, long nn, long planes, float s2, float v);
int sr_invw(hipStream_t st, const float2* FB, int N, int sf, float* invW);
int sr_solve_tile(hipStream_t st, const float2* Rs, const float* invW, const float2* FB, int N, int sf, long planes,
                  float s2, float v, float2* out);
int psf_embed(hipStream_t st, const float* psf, int kh, int kw, int N, float* plane);

// ---- ops.hip ----------------------------------------------------------------------------
// generic fp32 point-wise:  out = a*x + b*y (y may be null)
int axpby(hipStream_t st, const float* x, float a, const float* y, float b, long n, float* out);
int mul_planes(hipStream_t st, const float* x, const float* m, long n, long mn, float* out);       // out = x * m[i % mn]
int mul_elem(hipStream_t st, const float* x, const float* y, long n, float* out);
int strided_down(hipStream_t st, const float* x, int N, int sf, long planes, float* out);          // out = x[::sf, ::sf]
int zero_fill_up(hipStream_t st, const float* x, int n, int sf, long planes, float* out);          // phase-0 zero fill
int gather_idx(hipStream_t st, const float* x, const long* idx, long nidx, long per_sample, int B, float* out);
int scatter_idx(hipStream_t st, const float* y, const long* idx, long nidx, long per_sample, int B, float* out);
// separable / dense circular blur in the spatial domain (LDS staged)
int blur_sep_circ(hipStream_t st, const float* x, const float* k1d, int taps, int N, long planes, int axis, float* out);
int blur_dense_circ(hipStream_t st, const float* x, const float* k2d, int ks, int N, long planes, int adjoint, float* out);
// antialiased cubic resize along one axis (gather form) and its adjoint (scatter-add)
int resize_axis(hipStream_t st, const float* x, const float* w, const int* fov, int taps, int n_in, int n_out,
                int other, int axis, long planes, float* out);
int resize_axis_adj(hipStream_t st, const float* g, const float* w, const int* fov, int taps, int n_in, int n_out,
                    int other, int axis, long planes, float* out);
// Haar level-3 DWT in Mallat (pywt coeffs_to_array) layout
int dwt_haar3(hipStream_t st, const float* x, int N, long planes, float* out);
int idwt_haar3(hipStream_t st, const float* c, int N, long planes, float* out);
// 3-point orthonormal DCT across the channel planes of each sample (in place allowed)
int dct3_channels(hipStream_t st, const float* x, long HW, int B, int inverse, float* out);

// guidance epilogues (fp32 NCHW)
struct X0Params {
  float c_in, sqrt_recip, sqrt_recipm1, log_beta, log_post_var, post_var, coef1;
  int want_var;      // 1: convert per-pixel variance (Eq. 22)
};
int x0_epilogue_v1(hipStream_t st, const float* unet_out, const float* x, int B, long HW, X0Params p, float* x0_mean,
                   float* x0_raw, float* var);
int x0_epilogue_v2(hipStream_t st, const float* unet_out, const float* cov_out, const float* x, int B, long HW,
                   float sigma, int want_var, float* x0_mean, float* x0_var, float* theta_var);
int vjp_cotangent_v1(hipStream_t st, const float* ghat, const float* x0_raw, int B, long HW, float sqrt_recipm1,
                     float* cot6, float* g_raw);
int vjp_cotangent_v2(hipStream_t st, const float* ghat, int B, long HW, float* cot6);
// hat = clamp(x0 + coef * (a * g_direct + b * unet_vjp), -1, 1)
int guidance_combine(hipStream_t st, const float* x0_mean, const float* g_direct, float a, const float* unet_vjp,
                     float b, float coef, long n, float* hat);
int clamp_pm1(hipStream_t st, const float* x, long n, float* out);
// per-sample 2-norm: out[b] = sqrt(sum x^2)
int norm_per_sample(hipStream_t st, const float* x, int B, long per, float* out, double* tmp);
int scale_per_sample_inv(hipStream_t st, const float* x, const float* nrm, float zeta, int B, long per, float* out);

// sampler fusions (k_diffusion/sampling.py:46-48,118-135,159-184)
int sampler_add_noise(hipStream_t st, const float* x, const float* eps, float s, long n, float* out);
int sampler_euler(hipStream_t st, const float* x, const float* den, float sigma_hat, float dt, long n, float* out);
int sampler_heun(hipStream_t st, const float* x, const float* den1, const float* x2, const float* den2,
                 float sigma_hat, float sigma_next, float dt, long n, float* out);

// batched per-sample CG state machine
struct CgState {      // device arrays of length B
  double *rr, *pq, *rho_prev, *atol2;
  float *alpha, *beta;
  int *active, *iters, *any_active;
  int* unconverged;      // sticky: number of fixed-trip solves that stopped with an active sample (kdip_op_cg_unconverged reads + clears it)
};
int cg_dot(hipStream_t st, const float* a, const float* b, int B, long per, double* out);
int cg_init(hipStream_t st, CgState s, int B, float tol);                     // uses rr = b.b
int cg_step_a(hipStream_t st, CgState s, int B, int it, int count_unconverged = 0);   // active/beta from rr; count_unconverged: unconverged += any_active
int cg_update_p(hipStream_t st, CgState s, const float* r, float* p, int B, long per);
int cg_step_b(hipStream_t st, CgState s, int B);                              // alpha from rho/pq
int cg_update_xr(hipStream_t st, CgState s, float* x, float* r, const float* p, const float* q, int B, long per);

}  // namespace kdip
