// Internal C++ launch API shared by the translation units of libkdip_hip.
#pragma once
#include "common.h"
#include "det.h"

namespace kdip {

// ---- conv.hip ---------------------------------------------------------------------------
#ifndef KDIP_SPLITK_MAX
#define KDIP_SPLITK_MAX 16      // most K splits of an under-filled 3x3 launch = slabs of the deterministic split-K workspace (unet.hip sizes it with this)
#endif
// Optional GroupNorm statistics fused into the conv epilogue (see ConvParams::st_mode).
struct ConvStats {
  int mode = 0, silu = 0;
  double* sums = nullptr;          // [B][32][2], pre-zeroed by the caller
  const void* x = nullptr; long ldx = 0;
  const float* coef = nullptr; const float* mr = nullptr;
  // nearest-neighbour x2 upsampling folded into the reads (Upsample of an upsampling ResBlock, unet.py:232-235): the input
  // tensor / the residual tensor are HALF-resolution [B, H/2, W/2, C] and pixel (y, x) reads (y >> 1, x >> 1)
  int in_ups = 0, res_ups = 0;
  // split-precision mode (DT_F32X3): device word holding the bits of max |x| of the tensor family the INPUT belongs to (the VJP's
  // cotangent), or null for inputs of O(1) scale (forward activations): sets the fp16 window of the A operand (conv.hip, Mma<f32x3_t>)
  const unsigned* x3_amax = nullptr;
  unsigned* x3_sat = nullptr;      // ... and a device word whose bit 0 the kernel sets when a scaled operand leaves that window (|a| > 65504)
  // fp16-headed split (DT_F32H3) only: device word that receives (atomic max) the bits of the largest |scaled operand| this LAUNCH staged.  The
  // fp16 head loses its relative precision below 2^-14: a launch whose whole operand tensor sits below the window is found by
  // x3_lowpeak_check and flagged (bit 2 of x3_sat) -- the other side of the window watch
  unsigned* x3_lowpeak = nullptr;
  // split-precision 3x3 convs on maps with >= 128 pixels: the input is a GroupNorm INPUT and silu?(a*x + b), (a, b) = tf_coef [B][Cin][2]
  // (gn_coef), is applied while the patch is staged -- the activated tensor never exists in HBM (as Conv3Fuse::tf 1 does for bf16)
  const float* tf_coef = nullptr; int tf_silu = 0;
  // tf_mode 2 (split precision, 3x3 dgrad): GroupNorm BACKWARD apply while staging -- the input is dz = dy * silu'(z) as a store_dz epilogue left it,
  // tf_x2 [same shape and channel stride] is that GroupNorm's input, tf_coef [B][Cin][4] = (a, b, k0, k1) (gn_bwd_coef): A = a*dz - (k0 + k1*x2).
  // The gradient w.r.t. the GroupNorm input is never written (what Conv3Fuse::tf 2 does for bf16).  nn.py:17-19 / unet.py:237-253 backward
  int tf_mode = 1; const void* tf_x2 = nullptr;
  // deterministic modes (det.h): fused statistics reduced in a fixed order through det->slab (+ a finish kernel); sk_det:
  // sk_ws is NOT pre-zeroed and holds one slab [B*H*W][Cout] per K split (sk_ws_floats bounds the number of splits), summed in
  // split order by the finalize pass
  const DetWs* det = nullptr; int sk_det = 0;
  DetPending* defer = nullptr;     // (with det, modes 1 / 2) the finish pass is left to the consumer of the sums: *defer receives what it needs (det.h)
  // fp32 storage (f32 / split-precision modes), mode 2: the epilogue's backward-statistics sweep also WRITES dz = dy * silu'(a*x + b) over the
  // dy it stored (the lines are still L2-hot), as conv3's st_mode-2 epilogue does for bf16: every consumer of the tensor -- the
  // GroupNorm-backward apply pass, the gnb epilogue below, the TFM-2 staging of the next dgrad conv -- is then transcendental-free
  int store_dz = 0;
  // mode 3 (fp32 storage, one image per tile, Cout % tile width == 0): GroupNorm-BACKWARD APPLY folded into this conv's epilogue --
  //   y = conv(x) + [ a*dz - (k0 + k1*gx) ] (+ add),   (a, b, k0, k1) = gnb_coef [B][Cout][4] (gn_bwd_coef), dz / gx / add [B,H,W,Cout]-shaped
  // i.e. the ResBlock input gradient  gn_bwd_apply(conv1-dgrad output) + skip-dgrad (+ concat-skip gradient)  leaves the 1x1 skip dgrad
  // conv directly: the skip gradient is never written to / re-read from HBM and the separate gn_bwd_apply pass disappears
  // (guided_diffusion/unet.py:215-222,256 backward; nn.py:17-19)
  const float* gnb_coef = nullptr;
  const void* gnb_dz = nullptr; long gnb_lddz = 0;
  const void* gnb_x = nullptr; long gnb_ldx = 0;
  const void* gnb_add = nullptr; long gnb_lda = 0;
  int gnb_silu = 0;                // gnb_dz holds dy, not dz: the epilogue applies silu'(a*gx + b) itself (the 1x1 conv is HBM-bound: the arithmetic is free there)
};
inline bool conv_tf_eligible(DType cdt, int ntaps, int H, int W, int Cin_pad) { return is_x3(cdt) && ntaps == 9 && (long)H * W >= 128 && Cin_pad % 32 == 0; }
// true iff conv_forward can fuse statistics for an output of this shape
inline bool conv_stats_eligible(int H, int W, int Cout) { return (long)H * W >= 128 && Cout % 128 == 0; }
int conv_forward(hipStream_t st, DType dt, int ntaps, const void* x, long ldx, int B, int H, int W, int Cin,
                 const void* wp, const float* bias, int Cout, void* y, long ldy, const void* res, long ldr,
                 int out_f32, float alpha, int cin_real = 0, const ConvStats* stt = nullptr, float* sk_ws = nullptr, long sk_ws_floats = 0);
// sk_ws: optional fp32 split-K workspace of sk_ws_floats floats, all zero on entry and on return; when given, under-filled
// launches (small-spatial layers) split their K range over blockIdx.y and reduce through it
size_t packed_weight_bytes(DType dt, int ntaps, int Cin_pad, int Cout);
long x3_weight_subwindow();         // ... DT_F32H3: weight TENSORS packed so far whose largest scaled magnitude is non-zero and below 2^-12 (under the fp16 head's normal range)
int x3_lowpeak_check(hipStream_t st, const unsigned* peaks, int n, unsigned* flag);      // flag |= 4 when any launch's peak word is non-zero and below 2^-12
long x3_weight_saturations();       // running count (of the calling host thread) of DT_F32X3 weights packed so far whose scaled value left the fp16 window (|w| > 255.9)
int conv_debug_timing(void* buf, int H, int cin, int cout, int st_mode);   // -DKDIP_TIMING=1 diagnostic builds only
void pack_conv_weight(DType dt, const float* w, int Cout, int Cin, int ntaps, int transpose_flip, int Cin_pad_out,
                      void* out);

// ---- conv3.hip: second-generation bf16 3x3 conv for the large maps (W % 32 == 0, H % 8 == 0, Cout % 128 == 0), with the
// preceding GroupNorm(+FiLM+SiLU) forward apply or GroupNorm backward apply fused into the input staging ---------------
struct Conv3Fuse {
  int in_ups = 0, res_ups = 0;       // input / residual are half-resolution tensors read at (y >> 1, x >> 1)
  // staging transform of the input patch:
  //   tf 1: A = silu?(a*x + b)                       tf_coef [B][Cin][2] = (a, b)        (gn_coef)
  //   tf 2: A = a*x - (k0 + k1*x2), x = dz = dy * silu'(z) as a st_mode-2 epilogue stores it   tf_coef [B][Cin][4] = (a, b, k0, k1)   (gn_bwd_coef); x2 = GroupNorm input
  int tf = 0, tf_silu = 0;
  const float* tf_coef = nullptr;
  const void* x2 = nullptr; long ldx2 = 0;
  // GroupNorm-coefficient fold (instead of tf_coef): the conv computes the coefficients while it loads its table, so the tiny
  // gn_coef / gn_merge_stats / gn_bwd_coef launches leave the dependency chain between two convs.
  //   tf 1: (a, b) from the input's sums `fold_stats` [B][32][2] (optionally merged on the fly from two producers' sums: channels
  //         [0, fold_C1) from fold_stats, the rest from fold_stats2), gamma / beta / FiLM rows; written to fold_coef_out [B][Cin][2]
  //         and fold_mr_out [B][32][2] for the VJP (every block that loads an image's table writes the same values)
  //   tf 2: (a, b, k0, k1) from fold_coef [B][Cin][2], fold_mr [B][32][2] and the backward sums fold_stats [B][32][2]
  const double* fold_stats = nullptr; const double* fold_stats2 = nullptr; int fold_C1 = 0;
  const float *fold_gamma = nullptr, *fold_beta = nullptr, *fold_film = nullptr; long fold_film_ld = 0;
  long fold_HW = 0; float fold_eps = 1e-5f;
  float *fold_coef_out = nullptr, *fold_mr_out = nullptr;
  const float *fold_coef = nullptr, *fold_mr = nullptr;
  // statistics of the OUTPUT accumulated in the epilogue (same meaning as ConvStats::mode 1 / 2); with st_mode 2 the tensor
  // written is dz = dy * silu'(a*st_x + b), NOT dy: run the GroupNorm backward of it with silu = 0
  int st_mode = 0, st_silu = 0;
  double* st_sums = nullptr;
  const void* st_x = nullptr; long st_ldx = 0;
  const float* st_coef = nullptr; const float* st_mr = nullptr;
};
int conv3_debug_timing(void* buf);   // -DC3_TIMING=1 builds: [grid][8] uint64 per-block phase stamps of every later conv3 launch
int conv3_tf_max_cin(int tf);          // most input channels the staging-transform table of mode tf holds
bool conv3_eligible(DType dt, int ntaps, int H, int W, int Cin_pad, int Cout, long ldx, long ldy);
int conv3_forward(hipStream_t st, const void* x, long ldx, int B, int H, int W, int Cin, const void* wp, const float* bias, int Cout,
                  void* y, long ldy, const void* res, long ldr, const Conv3Fuse* fu, int cin_real = 0);

// ---- gemm.hip: strided batched GEMM  C[b] = alpha * A[b] (MxK) * B[b] (KxN) ----------------
// element strides; batch index b = b1*nb2 + b2 with separate strides per level.
struct BGemm {
  const void* A; long sam, sak, sab1, sab2;
  const void* Bm; long sbk, sbn, sbb1, sbb2;
  void* C; long scm, scn, scb1, scb2;
  int M, N, K, nb1, nb2;
  float alpha;
  int c_f32;       // C stored as fp32 regardless of dt
  int a_f32;       // A stored as fp32 regardless of dt
};
int bgemm(hipStream_t st, DType dt, const BGemm& g);

// ---- attention.hip: fused QKVAttentionLegacy forward / VJP (bf16, head width 64, T % 64 == 0); qkv [B][T][ld] with head h at
// channels 192 h + (q | k | v), output / cotangent [B][T][ld*] with head h at channels 64 h
bool attn_fused_eligible(DType dt, int T, int head_channels, long ld_qkv);
int f32_to_bf16(hipStream_t st, const float* x, long n, void* y);      // (test hook helper)
int attn_fused_forward(hipStream_t st, const void* qkv, long ld, int B, int T, int heads, void* vt_ws, void* o, long ldo, float* lse);
int attn_fused_backward(hipStream_t st, const void* qkv, long ld, const void* dO, long lddo, const void* o, long ldo, const float* lse,
                        int B, int T, int heads, void* ws, float* D, void* dqkv, long ldg);

// ---- norm.hip ---------------------------------------------------------------------------
// GroupNorm(32) over NHWC [B, HW, C] (ld = channel stride). stats: double [B][32][2] (sum, sumsq), zeroed by callee.
int gn_stats(hipStream_t st, DType dt, const void* x, long ldx, int B, long HW, int C, double* stats, int prezeroed = 0, const DetWs* det = nullptr);
// det (fp32 storage only): the sums are reduced in a fixed order through det->slab and WRITTEN (det.h) -- run-to-run bit-reproducible
// coef[B][C][2] = (a, b) with y = a*x + b  [then SiLU]; a = rstd*gamma*(1+scale), b = (beta - mean*rstd*gamma)*(1+scale)+shift
// film: [B][2C] fp32 (scale | shift) or null.  Also writes mr[B][32][2] = (mean, rstd) fp32.
// deferred finish passes of the fused conv statistics (conv.hip): sums [B][32][2] <- slab; the second form also computes the GroupNorm (+ FiLM)
// coefficients from them (the arithmetic of gn_coef_kernel): coef [B][C][2], mr [B][32][2]
int conv_stats_finish(hipStream_t st, const DetPending& pd, int B, double* sums);
int conv_stats_finish_coef(hipStream_t st, const DetPending& pd, int B, double* sums, const float* gamma, const float* beta, const float* film, long film_ld,
                           long HW, int C, float eps, float* coef, float* mr);
int gn_coef(hipStream_t st, const double* stats, const float* gamma, const float* beta, const float* film,
            int B, long HW, int C, float eps, float* coef, float* mr, long film_ld = 0);
int gn_apply(hipStream_t st, DType dt, const void* x, long ldx, const float* coef, int B, long HW, int C, int silu,
             void* y, long ldy);
// backward: dy wrt apply output -> dx (+ optional addend), two passes.
int gn_bwd_stats(hipStream_t st, DType dt, const void* x, long ldx, const void* dy, long lddy, const float* coef,
                 const float* mr, int B, long HW, int C, int silu, double* sums, int prezeroed = 0, int half_lgW = -1, const DetWs* det = nullptr);
int gn_bwd_apply(hipStream_t st, DType dt, const void* x, long ldx, const void* dy, long lddy, const float* coef,
                 const float* mr, const double* sums, int B, long HW, int C, int silu, const void* addend, long lda,
                 void* dx, long lddx, const void* addend2 = nullptr, long lda2 = 0, int half_lgW = -1);
// (a, b, k0, k1) per (image, channel) for a dgrad conv that applies this GroupNorm backward while staging (Conv3Fuse::tf 2)
int gn_bwd_coef(hipStream_t st, const float* coef, const float* mr, const double* sums, int B, long HW, int C, float* out);
// half_lgW >= 0 (W = 1 << half_lgW): dy / addend are HALF-resolution tensors standing for 0.25 * nearest-upsample (the adjoint
// of the 2x2 average pool of a downsampling ResBlock), read in place instead of being materialised at full resolution.
// GN apply (+SiLU) fused with the 2x2 average pool of both the activated tensor (yp) and the raw input (xp)
int gn_apply_pool2(hipStream_t st, DType dt, const void* x, long ldx, const float* coef, int B, int H, int W, int C, int silu,
                   void* yp, long ldy, void* xp, long ldxp);
// GroupNorm statistics of a channel concat from its two producers' statistics (group boundaries must nest)
bool gn_merge_eligible(int C1, int C2);
int gn_merge_stats(hipStream_t st, const double* s1, int C1, const double* s2, int C2, int B, double* out);
// small feature maps (HW <= 256): stats + coefficients + apply in one launch, one block per (image, group)
bool gn_small_eligible(DType dt, long HW, int C);
int gn_fwd_small(hipStream_t st, DType dt, const void* x, long ldx, int B, long HW, int C, const float* gamma,
                 const float* beta, const float* film, long film_ld, float eps, int silu, void* y, long ldy, float* coef,
                 float* mr);
int gn_bwd_small(hipStream_t st, DType dt, const void* x, long ldx, const void* dy, long lddy, const float* coef,
                 const float* mr, int B, long HW, int C, int silu, const void* addend, long lda, void* dx, long lddx,
                 const void* addend2 = nullptr, long lda2 = 0);

// ---- elementwise.hip --------------------------------------------------------------------
int avgpool2(hipStream_t st, DType dt, const void* x, long ldx, int B, int H, int W, int C, void* y, long ldy, float scale);
// sum of each 2x2 block (adjoint of nearest upsample): implemented as avgpool2 with scale 4.
int softmax_rows(hipStream_t st, DType dt, const float* s, long rows, int cols, void* p);
// dS = P * (dP - rowsum(dP*P)); dP fp32 in, P dtype T, dS out dtype T
int softmax_bwd_rows(hipStream_t st, DType dt, const void* p, const float* dp, long rows, int cols, void* ds);
// LPIPS glue on fp32 NCHW planes: y = relu(maxpool2?(x)); out[b] += LPIPS distance of one layer (see elementwise.hip)
int relu_maxpool_planes(hipStream_t st, const float* x, long planes, int H, int W, int pool, float* y);
int lpips_layer(hipStream_t st, const float* f0, const float* f1, const float* w, int B, int C, long HW, float* out);
int gauss_nll_mean(hipStream_t st, const float* pred, const float* target, const float* logvar, int B, long per, int accumulate, float* out);
int silu_f32(hipStream_t st, const float* x, long n, float* y);
// *out = bits of max |x| (fp32 bit patterns of non-negative floats order like unsigned integers); capture-safe (memset + one kernel)
int amax_bits(hipStream_t st, const float* x, long n, unsigned* out);
// sampled variant (every k-th 16-byte vector, k chosen so that ~64 K vectors are read); *out_zeroed must be zero on entry; n % 4 == 0, x 16-byte aligned
int amax_bits_sampled(hipStream_t st, const float* x, long n, unsigned* out_zeroed);
int timestep_embedding(hipStream_t st, const float* t, int B, int dim, float* out);
int f32_to_T(hipStream_t st, DType dt, const float* x, long n, void* y);
int T_to_f32(hipStream_t st, DType dt, const void* x, long n, float* y);
// NCHW fp32 [B,C,H,W] * scale -> NHWC T [B,H,W,ld] (channels >= C zero-filled up to Cpad)
int nchw_to_nhwc(hipStream_t st, DType dt, const float* x, int B, int C, int H, int W, float scale, void* y, long ld, int Cpad);
int nhwc_to_nchw_f32(hipStream_t st, const float* x, long ld, int B, int C, int H, int W, float* y);
// few-channel 3x3 convs as 1x1 convs (elementwise.hip): taps folded into K (im2col of the NCHW input) or into N (per-tap partial products + gather)
int im2col3_nchw(hipStream_t st, const float* x, int B, int C, int H, int W, float scale, float* y, long ld);
int tap_gather_nchw(hipStream_t st, const float* P, long ld, int B, int Co, int H, int W, const float* bias, float* out);
int nhwc_T_to_nchw_f32(hipStream_t st, DType dt, const void* x, long ld, int B, int C, int H, int W, float* y);

}  // namespace kdip
