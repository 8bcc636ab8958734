// Operator context + mat-solvers  mat = A^T (sigma_s^2 I + A C A^T)^-1 (y - A x0)
// (condition/condition.py:307-439) fully on device: closed forms through the LDS FFT,
// tensor-variance branch through a batched per-sample CG (scipy cg semantics, no host
// round trips per matvec -- the reference moves every operand to the CPU per call).
#include <functional>
#include <math.h>
#include "kernels.h"
#include "opctx.h"

namespace kdip {

#define CK(call) do { int _rc = (call); if (_rc) return _rc; } while (0)

OpCtx::~OpCtx() {
  for (void* p : allocs) (void)hipFree(p);
  if (h_any) (void)hipHostFree(h_any);
}

static int dmalloc(OpCtx* c, void** p, size_t bytes) {
  if (hipMalloc(p, bytes) != hipSuccess) return set_error(KDIP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
  c->allocs.push_back(*p);
  return KDIP_OK;
}

int OpCtx::init() {
  KDIP_HIP_CHECK(hipSetDevice(device));
  KDIP_REQUIRE(N == 256 || N == 64, "operator context: image size %d unsupported (64 or 256)", N);
  float2 host[128];
  make_twiddles256(host);
  CK(dmalloc(this, (void**)&tw, sizeof(host)));
  KDIP_HIP_CHECK(hipMemcpy(tw, host, sizeof(host), hipMemcpyHostToDevice));
  KDIP_HIP_CHECK(hipHostMalloc((void**)&h_any, sizeof(int) * 4));
  // allocated ONCE: captured fixed-trip graphs keep incrementing this word whatever the CG workspace does later
  CK(dmalloc(this, (void**)&cg.unconverged, sizeof(int)));
  KDIP_HIP_CHECK(hipMemset(cg.unconverged, 0, sizeof(int)));
  return KDIP_OK;
}

int OpCtx::ensure_ws(int B) {
  if (B <= wsB) return KDIP_OK;
  KDIP_HIP_CHECK(hipSetDevice(device));
  // (re)allocate; old buffers stay in `allocs` until destruction (B grows rarely)
  size_t nn = (size_t)N * N, P = (size_t)3 * B;
  CK(dmalloc(this, (void**)&c0, sizeof(float2) * P * nn));
  CK(dmalloc(this, (void**)&c1, sizeof(float2) * P * nn));
  CK(dmalloc(this, (void**)&ctmp, sizeof(float2) * P * nn));
  for (int i = 0; i < 10; ++i) CK(dmalloc(this, (void**)&rbuf[i], sizeof(float) * P * nn));
  CK(dmalloc(this, (void**)&cg.rr, sizeof(double) * B));
  CK(dmalloc(this, (void**)&cg.pq, sizeof(double) * B));
  CK(dmalloc(this, (void**)&cg.rho_prev, sizeof(double) * B));
  CK(dmalloc(this, (void**)&cg.atol2, sizeof(double) * B));
  CK(dmalloc(this, (void**)&cg.alpha, sizeof(float) * B));
  CK(dmalloc(this, (void**)&cg.beta, sizeof(float) * B));
  CK(dmalloc(this, (void**)&cg.active, sizeof(int) * B));
  CK(dmalloc(this, (void**)&cg.iters, sizeof(int) * B));
  CK(dmalloc(this, (void**)&cg.any_active, sizeof(int)));
  CK(dmalloc(this, (void**)&dtmp, sizeof(double) * B));
  wsB = B;
  ++ws_generation;                 // captured graphs hold the old buffers' addresses (graphs.py re-captures)
  return KDIP_OK;
}

int OpCtx::set_psf(const float* psf_host, int kh, int kw) {
  KDIP_HIP_CHECK(hipSetDevice(device));
  KDIP_REQUIRE(kh == kw && kh <= N, "psf %dx%d unsupported", kh, kw);
  ks = kh;
  CK(dmalloc(this, (void**)&psf, sizeof(float) * kh * kw));
  KDIP_HIP_CHECK(hipMemcpy(psf, psf_host, sizeof(float) * kh * kw, hipMemcpyHostToDevice));
  CK(dmalloc(this, (void**)&FB, sizeof(float2) * N * N));
  float* plane; float2* t;
  CK(dmalloc(this, (void**)&plane, sizeof(float) * N * N));
  CK(dmalloc(this, (void**)&t, sizeof(float2) * N * N));
  CK(psf_embed(0, psf, kh, kw, N, plane));                 // p2o: pad + roll (utils_sisr.py:33-36)
  CK(fft2(0, tw, N, plane, 1, t, FB, 0, 1, 0));            // OTF = fft2 (utils_sisr.py:38)
  if (kind == OP_SR) {
    CK(dmalloc(this, (void**)&invW, sizeof(float) * (N / sf) * (N / sf)));
    CK(sr_invw(0, FB, N, sf, invW));
  }
  KDIP_HIP_CHECK(hipDeviceSynchronize());
  return KDIP_OK;
}

int OpCtx::set_separable(const float* kr, const float* kc, int taps) {
  KDIP_HIP_CHECK(hipSetDevice(device));
  ktaps = taps;
  std::vector<float> rf(taps), cf(taps);
  for (int i = 0; i < taps; ++i) { rf[i] = kr[taps - 1 - i]; cf[i] = kc[taps - 1 - i]; }
  float** dst[4] = {&krow, &kcol, &krow_f, &kcol_f};
  const float* src[4] = {kr, kc, rf.data(), cf.data()};
  for (int i = 0; i < 4; ++i) {
    CK(dmalloc(this, (void**)dst[i], sizeof(float) * taps));
    KDIP_HIP_CHECK(hipMemcpy(*dst[i], src[i], sizeof(float) * taps, hipMemcpyHostToDevice));
  }
  return KDIP_OK;
}

int OpCtx::set_mask(const float* mask_host) {
  KDIP_HIP_CHECK(hipSetDevice(device));
  CK(dmalloc(this, (void**)&mask, sizeof(float) * 3 * N * N));
  KDIP_HIP_CHECK(hipMemcpy(mask, mask_host, sizeof(float) * 3 * N * N, hipMemcpyHostToDevice));
  return KDIP_OK;
}

int OpCtx::set_ortho(int type) {
  KDIP_REQUIRE(type >= 0 && type <= 2, "ortho type %d", type);
  ortho = type;
  if (type == OT_DCT && !dctD) {
    KDIP_HIP_CHECK(hipSetDevice(device));
    std::vector<float> D((size_t)N * N);
    for (int k = 0; k < N; ++k)
      for (int n = 0; n < N; ++n) {
        double ck = k == 0 ? sqrt(1.0 / N) : sqrt(2.0 / N);
        D[(size_t)k * N + n] = (float)(ck * cos(M_PI * (n + 0.5) * k / N));
      }
    CK(dmalloc(this, (void**)&dctD, sizeof(float) * N * N));
    KDIP_HIP_CHECK(hipMemcpy(dctD, D.data(), sizeof(float) * N * N, hipMemcpyHostToDevice));
  }
  return KDIP_OK;
}

// A x = Re ifft2(FB . fft2 x)  |  A^T y = Re ifft2(conj(FB) . fft2 y); separable PSFs use the LDS stencil.
int OpCtx::apply_A(hipStream_t st, const float* x, float* out, int B, int adjoint) {
  KDIP_REQUIRE(FB, "operator has no PSF");
  CK(ensure_ws(B));
  const long P = 3L * B, nn = (long)N * N;
  if (ktaps) {
    float* t = rbuf[9];
    CK(blur_sep_circ(st, x, adjoint ? kcol_f : kcol, ktaps, N, P, 1, t));
    CK(blur_sep_circ(st, t, adjoint ? krow_f : krow, ktaps, N, P, 0, out));
    return KDIP_OK;
  }
  CK(fft2(st, tw, N, x, 1, ctmp, c0, 0, P, 0));
  CK(cmul_otf(st, c0, FB, nn, P, adjoint));
  CK(fft2(st, tw, N, c0, 0, ctmp, out, 1, P, 1));
  return KDIP_OK;
}

static int dct2_planes(OpCtx* c, hipStream_t st, const float* x, float* out, float* tmp, long P, int inverse) {
  const int N = c->N;
  BGemm g;
  // forward: T = X D^T ; Y = D T.   inverse: T = Y D ; X = D^T T.
  g = BGemm(); g.A = x; g.sam = N; g.sak = 1; g.sab1 = (long)N * N; g.sab2 = 0;
  g.Bm = c->dctD; g.sbb1 = 0; g.sbb2 = 0;
  if (!inverse) { g.sbk = 1; g.sbn = N; } else { g.sbk = N; g.sbn = 1; }
  g.C = tmp; g.scm = N; g.scn = 1; g.scb1 = (long)N * N; g.scb2 = 0;
  g.M = N; g.N = N; g.K = N; g.nb1 = (int)P; g.nb2 = 1; g.alpha = 1.f; g.c_f32 = 1;
  CK(bgemm(st, DT_F32, g));
  g = BGemm(); g.A = c->dctD; g.sab1 = 0; g.sab2 = 0;
  if (!inverse) { g.sam = N; g.sak = 1; } else { g.sam = 1; g.sak = N; }
  g.Bm = tmp; g.sbk = N; g.sbn = 1; g.sbb1 = (long)N * N; g.sbb2 = 0;
  g.C = out; g.scm = N; g.scn = 1; g.scb1 = (long)N * N; g.scb2 = 0;
  g.M = N; g.N = N; g.K = N; g.nb1 = (int)P; g.nb2 = 1; g.alpha = 1.f; g.c_f32 = 1;
  CK(bgemm(st, DT_F32, g));
  return KDIP_OK;
}

int OpCtx::ortho_fwd(hipStream_t st, const float* x, float* out, int B) {
  const long P = 3L * B, nn = (long)N * N;
  if (ortho == OT_NONE) { if (out != x) KDIP_HIP_CHECK(hipMemcpyAsync(out, x, sizeof(float) * P * nn, hipMemcpyDeviceToDevice, st)); return KDIP_OK; }
  if (ortho == OT_DWT) return dwt_haar3(st, x, N, P, out);
  CK(ensure_ws(B));
  CK(dct2_planes(this, st, x, rbuf[8], rbuf[9], P, 0));
  return dct3_channels(st, rbuf[8], nn, B, 0, out);
}
int OpCtx::ortho_inv(hipStream_t st, const float* x, float* out, int B) {
  const long P = 3L * B, nn = (long)N * N;
  if (ortho == OT_NONE) { if (out != x) KDIP_HIP_CHECK(hipMemcpyAsync(out, x, sizeof(float) * P * nn, hipMemcpyDeviceToDevice, st)); return KDIP_OK; }
  if (ortho == OT_DWT) return idwt_haar3(st, x, N, P, out);
  CK(ensure_ws(B));
  CK(dct3_channels(st, x, nn, B, 1, rbuf[8]));
  return dct2_planes(this, st, rbuf[8], out, rbuf[9], P, 1);
}
int OpCtx::cov_apply(hipStream_t st, const float* x, const float* var, float* out, int B) {
  const long n = 3L * B * N * N;
  if (ortho == OT_NONE) return mul_elem(st, x, var, n, out);
  CK(ensure_ws(B));
  float* t = rbuf[7];
  CK(ortho_fwd(st, x, t, B));
  CK(mul_elem(st, t, var, n, t));
  return ortho_inv(st, t, out, B);
}

int OpCtx::sr_transpose(hipStream_t st, const float* y, float* out, int B) {
  CK(ensure_ws(B));
  const long P = 3L * B;
  float* up = rbuf[6];
  CK(zero_fill_up(st, y, N / sf, sf, P, up));
  return apply_A(st, up, out, B, 1);
}

// ---- batched per-sample CG driver (scipy.sparse.linalg.cg legacy-tol semantics) ----
static int cg_solve(OpCtx* c, hipStream_t st, const std::function<int(const float*, float*)>& matvec, const float* b,
                    float* x, float* r, float* p, float* q, int B, long per, int* iters_host, int* info_host) {
  const int maxiter = 1000;
  const long n = (long)B * per;
  KDIP_HIP_CHECK(hipMemsetAsync(x, 0, sizeof(float) * n, st));
  KDIP_HIP_CHECK(hipMemsetAsync(p, 0, sizeof(float) * n, st));
  KDIP_HIP_CHECK(hipMemcpyAsync(r, b, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
  CK(cg_dot(st, b, b, B, per, c->cg.rr));
  CK(cg_init(st, c->cg, B, 1e-4f));
  int it = 0;
  // fixed-trip mode (capture-safe, kdip_op_set_cg_fixed_trips): no convergence flag is read back; every sample freezes itself on
  // the device when it converges (cg_step_a), so surplus iterations change nothing.  One more residual check after the last
  // update leaves `any_active` = "some sample would have needed more iterations" for kdip_op_cg_unconverged.
  const int fixed = c->cg_fixed_trips;
  for (; it < (fixed > 0 ? fixed + 1 : maxiter); ++it) {
    CK(cg_dot(st, r, r, B, per, c->cg.rr));
    CK(cg_step_a(st, c->cg, B, it, fixed > 0 && it == fixed));
    if (fixed > 0) { if (it == fixed) break; }
    else if (it >= 2 && (it % 2) == 0) {
      KDIP_HIP_CHECK(hipMemcpyAsync(c->h_any, c->cg.any_active, sizeof(int), hipMemcpyDeviceToHost, st));
      KDIP_HIP_CHECK(hipStreamSynchronize(st));
      if (!c->h_any[0]) break;
    }
    CK(cg_update_p(st, c->cg, r, p, B, per));
    CK(matvec(p, q));
    CK(cg_dot(st, p, q, B, per, c->cg.pq));
    CK(cg_step_b(st, c->cg, B));
    CK(cg_update_xr(st, c->cg, x, r, p, q, B, per));
  }
  if (fixed > 0) {           // no host round trip: the counts are not known here
    for (int i = 0; i < B; ++i) { if (iters_host) iters_host[i] = -1; if (info_host) info_host[i] = -1; }
    return KDIP_OK;
  }
  if (iters_host || info_host) {
    std::vector<int> hi(B), ha(B);
    KDIP_HIP_CHECK(hipMemcpyAsync(hi.data(), c->cg.iters, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    KDIP_HIP_CHECK(hipMemcpyAsync(ha.data(), c->cg.active, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    KDIP_HIP_CHECK(hipStreamSynchronize(st));
    for (int i = 0; i < B; ++i) {
      if (iters_host) iters_host[i] = hi[i];
      if (info_host) info_host[i] = ha[i] ? maxiter : 0;     // scipy: info = maxiter when not converged
    }
  }
  return KDIP_OK;
}

int OpCtx::solve(hipStream_t st, const float* y, const float* x0, float v, const float* vt, int B, float* mat,
                 int* iters_host, int* info_host) {
  CK(ensure_ws(B));
  const long P = 3L * B, nn = (long)N * N, n = P * nn;
  float s = fmaxf(sigma_s, 1e-3f);                       // condition.py:321,353
  if (kind == OP_SR) s = fmaxf(s, 1e-2f);                // condition.py:404
  const float s2 = s * s;
  if (iters_host) for (int i = 0; i < B; ++i) iters_host[i] = 0;
  if (info_host) for (int i = 0; i < B; ++i) info_host[i] = 0;
  float *b = rbuf[0], *xw = rbuf[1], *r = rbuf[2], *p = rbuf[3], *q = rbuf[4], *t0 = rbuf[5];

  if (kind == OP_INPAINT) {
    KDIP_REQUIRE(mask, "inpainting operator has no mask");
    CK(axpby(st, y, 1.f, x0, -1.f, n, b));
    CK(mul_planes(st, b, mask, n, 3 * nn, b));            // m*y - m*x0  (condition.py:323,340)
    if (!vt) return axpby(st, b, 1.f / (s2 + v), nullptr, 0.f, n, mat);
    auto mv = [&](const float* in, float* out) -> int {
      CK(cov_apply(st, in, vt, t0, B));
      CK(mul_planes(st, t0, mask, n, 3 * nn, t0));
      return axpby(st, in, s2, t0, 1.f, n, out);
    };
    return cg_solve(this, st, mv, b, mat, r, p, q, B, 3 * nn, iters_host, info_host);
  }
  KDIP_REQUIRE(FB, "operator has no PSF/OTF");
  if (kind == OP_BLUR) {
    CK(apply_A(st, x0, t0, B, 0));
    CK(axpby(st, y, 1.f, t0, -1.f, n, b));                // y - A x0
    if (!vt) {
      CK(fft2(st, tw, N, b, 1, ctmp, c0, 0, P, 0));
      CK(otf_solve(st, c0, FB, nn, P, s2, v));            // / (s2 + v F2B) * conj(FB)  (condition.py:357)
      return fft2(st, tw, N, c0, 0, ctmp, mat, 1, P, 1);
    }
    auto mv = [&](const float* in, float* out) -> int {
      CK(apply_A(st, in, t0, B, 1));
      CK(cov_apply(st, t0, vt, t0, B));
      CK(apply_A(st, t0, out, B, 0));
      return axpby(st, in, s2, out, 1.f, n, out);
    };
    CK(cg_solve(this, st, mv, b, xw, r, p, q, B, 3 * nn, iters_host, info_host));
    return apply_A(st, xw, mat, B, 1);                    // mat = A^T u  (condition.py:384)
  }
  // ---- super-resolution: A = (stride-sf decimation) o (circular blur)
  const int ns = N / sf;
  const long nns = (long)ns * ns, nsm = P * nns;
  float *ax = rbuf[5], *d = rbuf[6];
  CK(apply_A(st, x0, ax, B, 0));
  CK(strided_down(st, ax, N, sf, P, d));
  CK(axpby(st, y, 1.f, d, -1.f, nsm, b));                 // y - down(A x0)
  if (!vt) {
    KDIP_REQUIRE(ns == 16 || ns == 64 || ns == 256, "SR: low-res size %d unsupported", ns);
    CK(fft2(st, tw, ns, b, 1, ctmp, c0, 0, P, 0));
    CK(sr_solve_tile(st, c0, invW, FB, N, sf, P, s2, v, c1));      // condition.py:409-410
    return fft2(st, tw, N, c1, 0, ctmp, mat, 1, P, 1);
  }
  float *up = rbuf[7], *full = rbuf[8];
  auto mv = [&](const float* in, float* out) -> int {
    CK(zero_fill_up(st, in, ns, sf, P, up));
    CK(apply_A(st, up, full, B, 1));
    // cov_apply uses rbuf[7] as scratch when a transform is active: keep `up` out of its way
    CK(cov_apply(st, full, vt, full, B));
    CK(apply_A(st, full, up, B, 0));
    CK(strided_down(st, up, N, sf, P, out));
    return axpby(st, in, s2, out, 1.f, nsm, out);
  };
  CK(cg_solve(this, st, mv, b, xw, r, p, q, B, 3 * nns, iters_host, info_host));
  CK(zero_fill_up(st, xw, ns, sf, P, up));
  return apply_A(st, up, mat, B, 1);
}

}  // namespace kdip
