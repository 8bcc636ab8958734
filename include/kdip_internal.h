/* libkdip_hip -- test and diagnostic hooks.  NOT part of the drop-in boundary.
 *
 * include/kdip.h is exactly the export list SURVEY.md section 8(b) asks for (what a maintainer of the reference binds).  The entry
 * points below exist so that tests/ can localise a kernel bug to one kernel (kdip_test_*) and so that tools/ can time the phases of a
 * block or switch an A/B knob (kdip_debug_*).  They are exported from the same shared library, are covered by the same
 * header <-> export <-> ctypes consistency test (tests/test_host_cpu.py::test_library_exports_every_declared_symbol), and may change
 * without notice. */
#ifndef KDIP_INTERNAL_H
#define KDIP_INTERNAL_H
#include "kdip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Where the blocks of a (CU-masked) stream run: out_host[2b] = HW_ID, out_host[2b+1] = XCC_ID of block b. */
int kdip_debug_cu_census(void* stream, int blocks, unsigned* out_host);

/* ------------------------------------------------------------------ low-level test hooks
 * (exercised by tests/ to localise kernel bugs; NHWC tensors of the UNet storage dtype) */
int kdip_test_conv(void* stream, int dtype, int ntaps, const float* x_nchw_dev, int B, int Cin, int H, int W,
                   const float* w_host, const float* bias_host, int Cout, int transpose_flip, float* y_nchw_dev,
                   int storage_out /* 0: fp32 NHWC epilogue (output heads); 1: storage-dtype epilogue (UNet-internal) */);
/* Test hook of the fused attention forward + VJP (replaces QKVAttentionLegacy.forward, guided_diffusion/unet.py:339-356, and its
 * autograd backward): device fp32 qkv [B][T][3C] (C = 64 * heads, head h at channels 192 h + q | k | v), dO [B][T][C]; inputs are
 * rounded to bf16; outputs o [B][T][C], dqkv [B][T][3C] as fp32.  T must be a multiple of 64. */
int kdip_test_attention(void* stream, const float* qkv_dev, const float* dO_dev, int B, int T, int heads, float* o_dev, float* dqkv_dev);

/* Second-generation bf16 3x3 conv (csrc/conv3.hip) with its fused GroupNorm staging transforms and epilogue statistics.
 * Tensor arguments are device fp32 NCHW; tf 1: tf_coef [B][Cin][2] = (a, b); tf 2: x = dy, x2 = GroupNorm input,
 * tf_coef [B][Cin][4] = (a, b, k0, k1); st_mode 1 / 2: sums_dev [B][32][2] (fp64) receives the GroupNorm forward / backward
 * sums of the output (mode 2: stx = GroupNorm input of the output, st_coef [B][Cout][2], st_mr [B][32][2]).
 * reps > 1: mean HIP-event microseconds per launch in *avg_us_host. */
int kdip_test_conv3(void* stream, const float* x_nchw_dev, const float* x2_nchw_dev, int B, int Cin, int H, int W,
                    const float* w_host, const float* bias_host, int Cout, int transpose_flip, int tf, const float* tf_coef_dev,
                    const float* res_nchw_dev, int in_ups, int res_ups, int st_mode, const float* stx_nchw_dev,
                    const float* st_coef_dev, const float* st_mr_dev, float* y_nchw_dev, double* sums_dev, int reps,
                    float* avg_us_host);
int kdip_test_groupnorm(void* stream, int dtype, const float* x_nchw_dev, int B, int C, int H, int W,
                        const float* gamma_host, const float* beta_host, const float* film_host, int silu,
                        float* y_nchw_dev, const float* dy_nchw_dev, float* dx_nchw_dev);
/* Diagnostic (libraries built with -DKDIP_TIMING=1 only; KDIP_ERR_UNSUPPORTED otherwise): every later 3x3 conv launch
 * matching (H, real Cin, Cout, fused-statistics mode) writes per-block phase timestamps (100 MHz ticks: start, first patch
 * staged, K loop done, end, + 3 epilogue sub-phases) to dev_buf[grid][8] (uint64).  dev_buf = NULL switches it off.  tools/conv_phases.py. */
int kdip_debug_conv_timing(void* dev_buf, int H, int cin, int cout, int st_mode);
/* Same for csrc/conv3.hip (-DC3_TIMING=1): dev_buf[grid][8] = start, first patch staged, K loop done, end (100 MHz ticks), XCC id. */
int kdip_debug_conv3_timing(void* dev_buf);
/* Test / A-B aid: 1 (default) = the large-map bf16 convs compute their GroupNorm staging coefficients from the statistics themselves
 * (no gn_coef / gn_merge_stats / gn_bwd_coef launches between two convs); 0 = separate coefficient kernels.  Results are bit-identical. */
int kdip_debug_gn_fold(int on);
/* Test / A-B aid (deterministic modes): 1 (default) = a forward conv with fused GroupNorm statistics leaves their fixed-order finish pass to the consuming
 * GroupNorm, which runs it inside its coefficient kernel (one launch less between two convs); 0 = finish pass behind the conv.  Results are bit-identical. */
int kdip_debug_defer_finish(int on);
/* Diagnostic (KDIP_F16X3 handles): the per-launch peak words of the LAST fp16-headed pass (forward or VJP) -- largest |scaled operand| each conv
 * launch staged, in launch order; *n_host = number of launches (<= max copied).  Synchronises `stream`.  tools/f16x3_check.py prints the
 * distribution: how far real workloads sit from the low side of the fp16 window. */
int kdip_debug_x3_peaks(kdip_unet* u, void* stream, float* peaks_host, int max, int* n_host);

#ifdef __cplusplus
}
#endif
#endif /* KDIP_INTERNAL_H */
