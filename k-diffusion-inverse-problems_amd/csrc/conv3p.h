// Parameter block shared by the second- (conv3.hip) and third-generation (conv4.hip) bf16 3x3 conv kernels.
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
  unsigned mg_nblk, mg_tpi, mg_tx;    // conv4: ceil(2^32 / d) of nblkN, tilesX * tilesY, tilesX (0: d == 1) -- tile index -> coordinates without divisions
};

// conv4.hip: third-generation kernel (one 8-wave block per CU, 16 x 32-pixel x 128-channel tiles, two wave groups in ping-pong).
// `p` is the block conv3_forward built (tile counts are recomputed for the 16-row tiles); returns KDIP_OK after the launch.
bool conv4_shape_ok(const Conv3Params& p, int tf, int stm, bool res);
int conv4_tf_max_cin(int tf);
int conv4_launch(const Conv3Params& p, int tf, int stm, bool res, hipStream_t st);

}  // namespace kdip
