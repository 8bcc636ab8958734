// Operator / solver context: device-resident OTF, PSF, mask, basis and CG workspace.
#pragma once
#include <vector>
#include "common.h"
#include "fftops.h"

namespace kdip {

enum OpKind { OP_INPAINT = 0, OP_BLUR = 1, OP_SR = 2 };
enum OrthoKind { OT_NONE = 0, OT_DWT = 1, OT_DCT = 2 };

struct OpCtx {
  int device = 0;
  int kind = OP_BLUR, N = 256, sf = 1;
  float sigma_s = 0.05f;
  int ortho = OT_NONE;
  float2* tw = nullptr;            // 128 twiddles
  float2* FB = nullptr;            // [N][N] OTF
  float* invW = nullptr;           // SR: [N/sf][N/sf]
  float* psf = nullptr; int ks = 0;        // dense PSF [ks][ks]
  float *krow = nullptr, *kcol = nullptr, *krow_f = nullptr, *kcol_f = nullptr; int ktaps = 0;   // separable factors (+flipped)
  float* mask = nullptr;           // [3][N][N]
  float* dctD = nullptr;           // [N][N] orthonormal DCT-II matrix
  // workspace
  int wsB = 0;
  long ws_generation = 0;          // bumped whenever ensure_ws re-allocates the buffers below (captured hipGraphs hold their addresses)
  float2 *c0 = nullptr, *c1 = nullptr, *ctmp = nullptr;
  float* rbuf[10] = {nullptr};
  CgState cg{};
  double* dtmp = nullptr;
  int* h_any = nullptr;            // pinned host flag
  int cg_fixed_trips = 0;          // > 0: every CG solve runs exactly this many iterations with no host read (hipGraph capture); converged samples stay frozen
  std::vector<void*> allocs;

  ~OpCtx();
  int init();
  int ensure_ws(int B);
  int set_psf(const float* psf_host, int kh, int kw);
  int set_separable(const float* krow_host, const float* kcol_host, int taps);
  int set_mask(const float* mask_host);
  int set_ortho(int type);
  // linear maps on [B,3,N,N] fp32 (model A of the solvers)
  int apply_A(hipStream_t st, const float* x, float* out, int B, int adjoint);
  int ortho_fwd(hipStream_t st, const float* x, float* out, int B);
  int ortho_inv(hipStream_t st, const float* x, float* out, int B);
  int cov_apply(hipStream_t st, const float* x, const float* var, float* out, int B);   // W^-1 diag(var) W x
  int solve(hipStream_t st, const float* y, const float* x0, float var_scalar, const float* var_tensor, int B,
            float* mat, int* iters_host, int* info_host);
  int sr_transpose(hipStream_t st, const float* y, float* out, int B);
};

}  // namespace kdip
