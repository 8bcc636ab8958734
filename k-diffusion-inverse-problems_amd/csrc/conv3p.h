// Parameter block of the second-generation bf16 3x3 conv kernel (conv3.hip; also used by the archived ping-pong variant,
// tools/experiments/conv4/).
#pragma once
#include "common.h"

namespace kdip {

struct Conv3Params {
  const bf16_t* x; long ldx;          // staged tensor (TF 0/1: the conv input or GroupNorm input; TF 2: dy of the GroupNorm output)
  const bf16_t* x2; long ldx2;        // TF 2: the GroupNorm input
  const uint4* wp;                    // packed weights (pack_conv_weight, bf16)
  const float* bias;                  // [Cout] or null
  const bf16_t* res; long ldr;        // residual or null
  bf16_t* y; long ldy;
  int B, H, W, Cin, Cout;
  int ntilesN, tilesX, tilesY, mtiles, nblkN;
  int in_ups, res_ups;
  unsigned long long* dbg;            // C3_TIMING builds: [grid][8] stamps (start, first patch staged, K loop done, end) + XCC id
  const float* tf_coef;               // TF 1: [B][Cin][2] (a, b); TF 2: [B][Cin][4] (a, b, k0, k1)
  int tf_silu;
  int st_silu;
  double* st_sums;                    // [B][32][2]
  const bf16_t* st_x; long st_ldx;    // mode 2: GroupNorm input of the produced gradient
  const float* st_coef;               // mode 2: [B][Cout][2]
  const float* st_mr;                 // mode 2: [B][32][2]
  // GroupNorm-coefficient fold (Conv3Fuse::fold_*): non-null fold_stats switches the table load from tf_coef to on-the-fly coefficients
  const double* fold_stats; const double* fold_stats2; int fold_C1;
  const float *fold_gamma, *fold_beta, *fold_film; long fold_film_ld;
  long fold_HW; float fold_eps;
  float *fold_coef_out, *fold_mr_out;
  const float *fold_coef, *fold_mr;
  unsigned mg_nblk, mg_tpi, mg_tx;    // conv4: ceil(2^32 / d) of nblkN, tilesX * tilesY, tilesX (0: d == 1) -- tile index -> coordinates without divisions
};

#ifdef __HIPCC__
// (a, b) of channel c of image b for the GroupNorm + FiLM in front of a conv: the arithmetic of gn_coef_kernel (norm.hip), with the
// group sums optionally merged from two producers as gn_merge_stats_kernel does.  Also returns the group's (mean, rstd).
__device__ __forceinline__ void c3_fold_coef_fwd(const Conv3Params& p, int b, int c, float& a, float& bb, float& mean_f, float& rstd_f) {
  const int C = p.Cin, cpg = C >> 5, g = c / cpg;
  double s1, s2;
  if (p.fold_stats2) {
    const int C1 = p.fold_C1, C2 = C - C1, c0 = g * cpg;
    const double* src = c0 < C1 ? p.fold_stats + (long)b * 64 : p.fold_stats2 + (long)b * 64;
    const int cpgs = (c0 < C1 ? C1 : C2) >> 5, g0 = (c0 < C1 ? c0 : c0 - C1) / cpgs;
    s1 = 0; s2 = 0;
    for (int k = 0; k < cpg / cpgs; ++k) { s1 += src[(g0 + k) * 2]; s2 += src[(g0 + k) * 2 + 1]; }
  } else {
    s1 = p.fold_stats[((long)b * 32 + g) * 2]; s2 = p.fold_stats[((long)b * 32 + g) * 2 + 1];
  }
  const double n = (double)p.fold_HW * cpg;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)p.fold_eps));
  const float m = (float)mean;
  a = rstd * p.fold_gamma[c];
  bb = p.fold_beta[c] - m * a;
  if (p.fold_film) {
    const float sc = 1.f + p.fold_film[(long)b * p.fold_film_ld + c], sh = p.fold_film[(long)b * p.fold_film_ld + C + c];
    a *= sc;
    bb = bb * sc + sh;
  }
  mean_f = m; rstd_f = rstd;
}
// (a, b, k0, k1) of channel c for the GroupNorm backward applied while staging: the arithmetic of gn_bwd_coef_kernel (norm.hip)
__device__ __forceinline__ float4 c3_fold_coef_bwd(const Conv3Params& p, int b, int c) {
  const int C = p.Cin, cpg = C >> 5, g = c / cpg;
  const float invN = 1.f / ((float)p.fold_HW * (float)cpg);
  const float mean = p.fold_mr[((long)b * 32 + g) * 2], rstd = p.fold_mr[((long)b * 32 + g) * 2 + 1];
  const float t1 = (float)p.fold_stats[((long)b * 32 + g) * 2] * invN, t2 = (float)p.fold_stats[((long)b * 32 + g) * 2 + 1] * invN;
  const float k1 = rstd * t2, k0 = t1 - mean * k1;
  const long i = (long)b * C + c;
  return make_float4(p.fold_coef[i * 2], p.fold_coef[i * 2 + 1], k0, k1);
}
#endif

}  // namespace kdip
