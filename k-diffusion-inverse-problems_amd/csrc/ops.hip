// Measurement operators, basis transforms, guidance epilogues, sampler updates and the
// batched on-device CG vector kernels.  Everything here is fp32 NCHW planes and HBM-bound;
// kernels are written as coalesced streaming passes (LDS staging where a stencil re-reads).
//
// Reference call sites: condition/measurements.py:86-244 (operators), utils_sisr.py:44-61
// (zero-fill up / strided down), dps_utils/resizer.py:55-74 (Resizer.forward),
// condition/utils.py:88-139 (DCT / Haar DWT), condition/condition.py:133-183,231-300
// (guidance algebra), guided_diffusion/gaussian_diffusion.py:262-276,293-333 (x0 / variance),
// k_diffusion/sampling.py:46-48,118-135,159-184 (sampler), scipy cg at condition.py:343,379,432.
#include "common.h"
#include "fftops.h"

namespace kdip {

static inline int pw_grid(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ----------------------------------------------------------------------- point-wise ----
__global__ void axpby_kernel(const float* x, float a, const float* y, float b, long n, float* out) {
  GRID_STRIDE(i, n) out[i] = y ? a * x[i] + b * y[i] : a * x[i];
}
int axpby(hipStream_t st, const float* x, float a, const float* y, float b, long n, float* out) {
  ProfScope ps_(st, PC_OP_POINTWISE, (double)n * sizeof(float) * (y ? 3 : 2), "axpby", n);
  hipLaunchKernelGGL(axpby_kernel, dim3(pw_grid(n)), dim3(256), 0, st, x, a, y, b, n, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void mul_planes_kernel(const float* x, const float* m, long n, long mn, float* out) {
  GRID_STRIDE(i, n) out[i] = x[i] * m[i % mn];
}
int mul_planes(hipStream_t st, const float* x, const float* m, long n, long mn, float* out) {
  ProfScope ps_(st, PC_OP_GATHER, (double)n * sizeof(float) * 2 + (double)mn * sizeof(float), "mask_mul", n);
  hipLaunchKernelGGL(mul_planes_kernel, dim3(pw_grid(n)), dim3(256), 0, st, x, m, n, mn, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void mul_elem_kernel(const float* x, const float* y, long n, float* out) { GRID_STRIDE(i, n) out[i] = x[i] * y[i]; }
int mul_elem(hipStream_t st, const float* x, const float* y, long n, float* out) {
  ProfScope ps_(st, PC_OP_POINTWISE, (double)n * sizeof(float) * 3, "mul", n);
  hipLaunchKernelGGL(mul_elem_kernel, dim3(pw_grid(n)), dim3(256), 0, st, x, y, n, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void strided_down_kernel(const float* x, int N, int sf, long planes, float* out) {
  const int n = N / sf;
  GRID_STRIDE(i, planes * n * n) {
    long p = i / (n * n); int rc = (int)(i % (n * n)); int r = rc / n, c = rc % n;
    out[i] = x[p * (long)N * N + (long)(r * sf) * N + c * sf];
  }
}
int strided_down(hipStream_t st, const float* x, int N, int sf, long planes, float* out) {
  hipLaunchKernelGGL(strided_down_kernel, dim3(pw_grid(planes * (N / sf) * (N / sf))), dim3(256), 0, st, x, N, sf, planes, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void zero_fill_up_kernel(const float* x, int n, int sf, long planes, float* out) {
  const int N = n * sf;
  GRID_STRIDE(i, planes * (long)N * N) {
    long p = i / ((long)N * N); int rc = (int)(i % ((long)N * N)); int R = rc / N, C = rc % N;
    out[i] = (R % sf == 0 && C % sf == 0) ? x[p * n * n + (long)(R / sf) * n + C / sf] : 0.f;
  }
}
int zero_fill_up(hipStream_t st, const float* x, int n, int sf, long planes, float* out) {
  hipLaunchKernelGGL(zero_fill_up_kernel, dim3(pw_grid(planes * (long)n * sf * n * sf)), dim3(256), 0, st, x, n, sf, planes, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

// gather / scatter at precomputed (c,h,w)-lexicographic flat indices (measurements.py:217-236)
__global__ void gather_idx_kernel(const float* x, const long* idx, long nidx, long per, int B, float* out) {
  GRID_STRIDE(i, (long)B * nidx) { long b = i / nidx, j = i % nidx; out[i] = x[b * per + idx[j]]; }
}
int gather_idx(hipStream_t st, const float* x, const long* idx, long nidx, long per, int B, float* out) {
  ProfScope ps_(st, PC_OP_GATHER, (double)B * nidx * (2 * sizeof(float)) + (double)nidx * sizeof(long), "gather", B, nidx, per);
  hipLaunchKernelGGL(gather_idx_kernel, dim3(pw_grid((long)B * nidx)), dim3(256), 0, st, x, idx, nidx, per, B, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void scatter_idx_kernel(const float* y, const long* idx, long nidx, long per, int B, float* out) {
  GRID_STRIDE(i, (long)B * nidx) { long b = i / nidx, j = i % nidx; out[b * per + idx[j]] = y[i]; }
}
int scatter_idx(hipStream_t st, const float* y, const long* idx, long nidx, long per, int B, float* out) {
  ProfScope ps_(st, PC_OP_GATHER, (double)B * nidx * sizeof(float) + (double)B * per * sizeof(float) + (double)nidx * sizeof(long), "scatter", B, nidx, per);
  KDIP_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float) * B * per, st));
  hipLaunchKernelGGL(scatter_idx_kernel, dim3(pw_grid((long)B * nidx)), dim3(256), 0, st, y, idx, nidx, per, B, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

// ---------------------------------------------------------------- circular blur ----
// Separable pass: out[n] = sum_t k[t] * x[(n - (t - taps/2)) mod N] along `axis`.
// A block stages LPB full lines in LDS (same line<->memory mapping as the FFT passes).
template <int LPB>
__global__ __launch_bounds__(256) void blur_sep_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                        int taps, int N, int axis, float* __restrict__ out) {
  extern __shared__ float sm[];               // [LPB][N+1] lines, then taps
  float* kk = sm + LPB * (N + 1);
  const int tid = threadIdx.x;
  const long pbase = (long)blockIdx.y * N * N;
  const int line0 = blockIdx.x * LPB;
  for (int t = tid; t < taps; t += 256) kk[t] = k[t];
  for (int e = tid; e < LPB * N; e += 256) {
    int l, i; long g;
    if (axis == 1) { l = e / N; i = e % N; g = pbase + (long)(line0 + l) * N + i; }
    else { i = e / LPB; l = e % LPB; g = pbase + (long)i * N + (line0 + l); }
    sm[l * (N + 1) + i] = x[g];
  }
  __syncthreads();
  const int half = taps / 2, mask = N - 1;
  for (int e = tid; e < LPB * N; e += 256) {
    int l, i; long g;
    if (axis == 1) { l = e / N; i = e % N; g = pbase + (long)(line0 + l) * N + i; }
    else { i = e / LPB; l = e % LPB; g = pbase + (long)i * N + (line0 + l); }
    const float* ln = sm + l * (N + 1);
    float s = 0.f;
    for (int t = 0; t < taps; ++t) s += kk[t] * ln[(i - t + half) & mask];
    out[g] = s;
  }
}
// Register-blocked variant for taps <= 63 (the 61-tap Gaussian PSF): a thread produces 8 consecutive outputs of a line
// from one sliding pass over 70 staged samples with the (zero-padded, centred) 63 taps in registers -- 16 LDS reads per
// output instead of 122, which turns the pass from LDS-bound (130 us for 192 planes of 256^2) into a streaming one.
template <int LPB>
__global__ __launch_bounds__(256) void blur_sep63_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                          int taps, int N, int axis, float* __restrict__ out) {
  constexpr int KT = 63, R = 8;
  extern __shared__ float sm[];               // [LPB] lines of N samples, one pad word after every 8 (and one per line): in the row pass a wave's
                                              // threads sit on consecutive 8-sample segments of ONE line -- a stride of 8 words is an 8-way bank
                                              // conflict on every read (100 vs 61 us per 192-plane pass), a stride of 9 is none
  const int LS = N + (N >> 3) + 1;
  const int tid = threadIdx.x;
  const long pbase = (long)blockIdx.y * N * N;
  const int line0 = blockIdx.x * LPB;
  float kr[KT];
  const int shift = (KT - taps) / 2;          // taps odd <= 63: k'[t + shift] = k[t], half' = 31
#pragma unroll
  for (int t = 0; t < KT; ++t) kr[t] = (t >= shift && t - shift < taps) ? k[t - shift] : 0.f;
  if (axis == 1) {       // rows: 16-byte loads (N % 8 == 0, planes 16-byte aligned: checked by the launcher)
    for (int e = tid; e < LPB * (N / 4); e += 256) {
      const int l = e / (N / 4), i = (e % (N / 4)) * 4;
      const float4 q = *(const float4*)(x + pbase + (long)(line0 + l) * N + i);
      float* d = sm + l * LS + i + (i >> 3);       // (4 consecutive samples never straddle a pad: i % 4 == 0)
      d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
    }
  } else {
    for (int e = tid; e < LPB * N; e += 256) {
      const int i = e / LPB, l = e % LPB;
      sm[l * LS + i + (i >> 3)] = x[pbase + (long)i * N + (line0 + l)];
    }
  }
  __syncthreads();
  const int mask = N - 1, segs = N / R;
  for (int w = tid; w < LPB * segs; w += 256) {
    // axis 1: consecutive threads -> consecutive segments of a line; axis 0: consecutive threads -> consecutive lines
    const int l = axis == 1 ? w / segs : w % LPB, i0 = (axis == 1 ? w % segs : w / LPB) * R;
    const float* ln = sm + l * LS;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    // out[i0 + r] = sum_t k'[t] ln[(i0 + r - t + 31) & mask]; with j = r - t + 62: sample ln[(i0 + j - 31) & mask].
    // i0 is a multiple of 8: sample index = (b + k) & mask with b = (i0 - 32) & mask (a multiple of 8) and k = j + 1, so the padded LDS
    // index is 9 * (((b >> 3) + (k >> 3)) & (N / 8 - 1)) + (k & 7): one wrap per group of 8 samples instead of three VALU per sample
    const int b8 = ((i0 - 32) & mask) >> 3, m8 = (N >> 3) - 1;
#pragma unroll
    for (int kq = 0; kq <= (KT + R - 1) / 8; ++kq) {
      const float* grp = ln + ((b8 + kq) & m8) * 9;
#pragma unroll
      for (int k7 = 0; k7 < 8; ++k7) {
        const int j = kq * 8 + k7 - 1;        // compile-time after unrolling
        if (j < 0 || j >= KT + R - 1) continue;
        const float v = grp[k7];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int t = r + (KT - 1) - j;
          if (t >= 0 && t < KT) acc[r] += kr[t] * v;
        }
      }
    }
    if (axis == 1) {     // 8 consecutive outputs of a row: two 16-byte stores
      float4* o4 = (float4*)(out + pbase + (long)(line0 + l) * N + i0);
      o4[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      o4[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) out[pbase + (long)(i0 + r) * N + (line0 + l)] = acc[r];
    }
  }
}

int blur_sep_circ(hipStream_t st, const float* x, const float* k1d, int taps, int N, long planes, int axis, float* out) {
  ProfScope ps_(st, PC_OP_BLUR, (double)planes * N * N * sizeof(float) * 2, "blur_sep", planes, N, taps, axis);
  KDIP_REQUIRE((N & (N - 1)) == 0 && N >= 16, "blur: N=%d must be a power of two", N);
  constexpr int LPB = 16;
  if ((taps & 1) && taps <= 63 && N % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0) {
#ifndef KDIP_BLUR_LPB_WIDE
#define KDIP_BLUR_LPB_WIDE 32
#endif
#ifndef KDIP_BLUR_LPB
#define KDIP_BLUR_LPB 8             // 8 lines per block = one 8-output segment per thread and 2 x the blocks: 20.7 -> 16.9 us per pass of 24 planes (16: 20.7, 4: 19.7)
#endif
    constexpr int L63 = KDIP_BLUR_LPB;       // lines per block of the register-blocked kernel
    // the column pass (axis 0) touches LPB consecutive floats of every row: 8 lines = 32-byte pieces of the 128-byte lines.  Once there
    // are enough planes to fill the chip with 32-line blocks, take whole lines (KDIP_BLUR_LPB_WIDE); few planes keep the many small blocks
    // (both passes: a block also loads the 63 taps once, amortised over 4 x the work)
    // (dynamic LDS stays under the 64 KB a launch gets without raising the kernel's cap: the 32-line block needs 32 (N + N / 8 + 1) floats
    // = 73.9 KB at N = 512 -- larger images take the 8-line block, 18.5 KB there)
    constexpr size_t LDS_CAP = 64 * 1024;
    if (N % KDIP_BLUR_LPB_WIDE == 0 && planes * (N / KDIP_BLUR_LPB_WIDE) >= 1024 && sizeof(float) * KDIP_BLUR_LPB_WIDE * (N + N / 8 + 1) <= LDS_CAP) {
      constexpr int LW = KDIP_BLUR_LPB_WIDE;
      hipLaunchKernelGGL(blur_sep63_kernel<LW>, dim3(N / LW, (unsigned)planes), dim3(256), sizeof(float) * LW * (N + N / 8 + 1), st, x, k1d, taps, N, axis, out);
      KDIP_LAUNCH_CHECK(); return KDIP_OK;
    }
    KDIP_REQUIRE(sizeof(float) * L63 * (N + N / 8 + 1) <= LDS_CAP, "blur: N=%d needs more than 64 KB of LDS per block", N);
    hipLaunchKernelGGL(blur_sep63_kernel<L63>, dim3(N / L63, (unsigned)planes), dim3(256), sizeof(float) * L63 * (N + N / 8 + 1), st, x, k1d, taps, N,
                       axis, out);
    KDIP_LAUNCH_CHECK(); return KDIP_OK;
  }
  size_t lds = sizeof(float) * (LPB * (N + 1) + taps);
  KDIP_REQUIRE(lds <= 64 * 1024, "blur: N=%d needs more than 64 KB of LDS per block", N);
  hipLaunchKernelGGL(blur_sep_kernel<LPB>, dim3(N / LPB, (unsigned)planes), dim3(256), lds, st, x, k1d, taps, N, axis, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

// Dense ks x ks circular convolution (adjoint=0: true convolution centred at ks/2; adjoint=1: correlation).
// 32x32 output tile + halo staged in LDS; zero taps are skipped (motion PSF has 217 of 3721 non-zero).
__global__ __launch_bounds__(256) void blur_dense_kernel(const float* __restrict__ x, const float* __restrict__ k2d,
                                                          int ks, int N, int adjoint, float* __restrict__ out) {
  extern __shared__ float sm[];
  const int half = ks / 2, TS = 32, HS = TS + 2 * half;
  float* tile = sm;                  // [HS][HS+1]
  float* kk = sm + HS * (HS + 1);    // [ks*ks]
  const int tid = threadIdx.x, mask = N - 1;
  const long pbase = (long)blockIdx.z * N * N;
  const int r0 = blockIdx.y * TS, c0 = blockIdx.x * TS;
  for (int t = tid; t < ks * ks; t += 256) kk[t] = k2d[t];
  for (int e = tid; e < HS * HS; e += 256) {
    int hr = e / HS, hc = e % HS;
    tile[hr * (HS + 1) + hc] = x[pbase + (long)((r0 + hr - half) & mask) * N + ((c0 + hc - half) & mask)];
  }
  __syncthreads();
  for (int e = tid; e < TS * TS; e += 256) {
    int r = e / TS, c = e % TS;
    float s = 0.f;
    for (int i = 0; i < ks; ++i)
      for (int j = 0; j < ks; ++j) {
        float w = kk[i * ks + j];
        if (w == 0.f) continue;
        // conv: x[r - (i-half)][c - (j-half)] ; corr: x[r + (i-half)][c + (j-half)]
        int hr = adjoint ? r + i : r + 2 * half - i;
        int hc = adjoint ? c + j : c + 2 * half - j;
        s += w * tile[hr * (HS + 1) + hc];
      }
    out[pbase + (long)(r0 + r) * N + (c0 + c)] = s;
  }
}
int blur_dense_circ(hipStream_t st, const float* x, const float* k2d, int ks, int N, long planes, int adjoint, float* out) {
  ProfScope ps_(st, PC_OP_BLUR, (double)planes * N * N * sizeof(float) * 2, "blur_dense", planes, N, ks, adjoint);
  KDIP_REQUIRE((N & (N - 1)) == 0 && N >= 32 && (ks & 1), "blur: bad N=%d / ks=%d", N, ks);
  const int HS = 32 + 2 * (ks / 2);
  size_t lds = sizeof(float) * (HS * (HS + 1) + ks * ks);
  hipLaunchKernelGGL(blur_dense_kernel, dim3(N / 32, N / 32, (unsigned)planes), dim3(256), lds, st, x, k2d, ks, N, adjoint, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

// ---------------------------------------------------------------------- Resizer ----
__global__ void resize_axis_kernel(const float* __restrict__ x, const float* __restrict__ w, const int* __restrict__ fov,
                                   int taps, int n_in, int n_out, int other, int axis, long planes, float* __restrict__ out) {
  // axis 1: x[p][other][n_in] -> out[p][other][n_out];  axis 0: x[p][n_in][other] -> out[p][n_out][other]
  GRID_STRIDE(i, planes * (long)n_out * other) {
    long p = i / ((long)n_out * other); int rc = (int)(i % ((long)n_out * other));
    int o, q;
    if (axis == 1) { q = rc / n_out; o = rc % n_out; } else { o = rc / other; q = rc % other; }
    const float* xp = x + p * (long)n_in * other;
    float s = 0.f;
    for (int t = 0; t < taps; ++t) {
      int src = fov[o * taps + t];
      float xv = (axis == 1) ? xp[(long)q * n_in + src] : xp[(long)src * other + q];
      s += xv * w[o * taps + t];
    }
    out[i] = s;
  }
}
// axis 1 (the contiguous axis) through LDS: a block stages ROWS whole input rows with coalesced 16-byte loads and keeps the (weight,
// source index) tables in LDS; every thread then gathers its outputs from LDS.  (The direct form reads the tables and the samples with
// per-lane addresses, three dependent loads per tap: 217 us per 192-plane pass against 21 us for the axis-0 pass of the same data.)
template <int ROWS, int MAXT>
__global__ __launch_bounds__(256) void resize_axis1_lds_kernel(const float* __restrict__ x, const float* __restrict__ w, const int* __restrict__ fov,
                                                                int taps, int n_in, int n_out, long rows_total, float* __restrict__ out) {
  extern __shared__ float rs_sm[];             // [ROWS] rows of n_in samples, one pad word after every 4 (neighbouring outputs read samples 1 / scale
                                               // = 4 apart: a stride of 5 words spreads them over the banks) + one per row
  const int LS = n_in + (n_in >> 2) + 1;
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * ROWS;
  const int nv = n_in / 4;                     // n_in % 4 == 0 (checked by the launcher)
  for (int e = tid; e < ROWS * nv; e += 256) {
    const int r = e / nv, v = e - r * nv;
    if (row0 + r < rows_total) {
      const float4 q = *(const float4*)(x + (row0 + r) * n_in + v * 4);
      float* d = rs_sm + r * LS + v * 5;
      d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
    }
  }
  // thread -> (output column o, row group): the column's taps stay in registers for all its rows (padded to MAXT with weight 0)
  const int ncol = n_out < 256 ? n_out : 256, ngrp = 256 / ncol;
  const int oc = tid % ncol, grp = tid / ncol;
  __syncthreads();
  for (int o = oc; o < n_out; o += ncol) {
    float wr[MAXT]; int fr[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const bool live = t < taps;
      wr[t] = live ? w[o * taps + t] : 0.f;
      const int si = live ? fov[o * taps + t] : 0;
      fr[t] = si + (si >> 2);
    }
    if (grp < ngrp) {
      for (int r = grp; r < ROWS; r += ngrp) {
        if (row0 + r >= rows_total) break;
        const float* xr = rs_sm + r * LS;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) acc += xr[fr[t]] * wr[t];
        out[(row0 + r) * n_out + o] = acc;
      }
    }
  }
}

int resize_axis(hipStream_t st, const float* x, const float* w, const int* fov, int taps, int n_in, int n_out, int other,
                int axis, long planes, float* out) {
  ProfScope ps_(st, PC_OP_RESIZE, (double)planes * other * ((double)n_in + n_out) * sizeof(float), "resize", planes, n_in, n_out, axis);
  constexpr int RROWS = 16, RMAXT = 16;
  const size_t lds = sizeof(float) * (size_t)RROWS * (n_in + n_in / 4 + 1);
  if (axis == 1 && n_in % 4 == 0 && ((uintptr_t)x % 16) == 0 && taps <= RMAXT && lds <= 60 * 1024) {
    const long rows = planes * other;
    hipLaunchKernelGGL((resize_axis1_lds_kernel<RROWS, RMAXT>), dim3((unsigned)((rows + RROWS - 1) / RROWS)), dim3(256), lds, st, x, w, fov, taps, n_in, n_out, rows, out);
    KDIP_LAUNCH_CHECK(); return KDIP_OK;
  }
  hipLaunchKernelGGL(resize_axis_kernel, dim3(pw_grid(planes * (long)n_out * other)), dim3(256), 0, st, x, w, fov, taps, n_in,
                     n_out, other, axis, planes, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
// adjoint of resize_axis as a GATHER: input element `src` collects g[o] * w[o][t] over every (o, t) whose field of view hits it, in
// (o, t) order -- no atomics, so the result does not depend on the order blocks arrive in (the scatter form added with fp32 atomics)
__global__ void resize_axis_adj_kernel(const float* __restrict__ g, const float* __restrict__ w, const int* __restrict__ fov,
                                       int taps, int n_in, int n_out, int other, int axis, long planes, float* __restrict__ out) {
  GRID_STRIDE(i, planes * (long)n_in * other) {
    long p = i / ((long)n_in * other); int rc = (int)(i % ((long)n_in * other));
    int src, q;
    if (axis == 1) { q = rc / n_in; src = rc % n_in; } else { src = rc / other; q = rc % other; }
    const float* gp = g + p * (long)n_out * other;
    float s = 0.f;
    for (int o = 0; o < n_out; ++o) {
      // the field of view of output o is a run of consecutive input indices, folded back at the borders (mirror padding): away from
      // the borders [first tap, last tap] bounds it and most outputs are skipped after two loads
      const int f0 = fov[o * taps], f1 = fov[o * taps + taps - 1];
      const int lo = f0 < f1 ? f0 : f1, hi = f0 < f1 ? f1 : f0;
      if (lo >= taps && hi < n_in - taps && (src < lo || src > hi)) continue;
      const float gv = (axis == 1) ? gp[(long)q * n_out + o] : gp[(long)o * other + q];
      for (int t = 0; t < taps; ++t)
        if (fov[o * taps + t] == src) s += gv * w[o * taps + t];
    }
    out[i] = s;
  }
}
int resize_axis_adj(hipStream_t st, const float* g, const float* w, const int* fov, int taps, int n_in, int n_out, int other,
                    int axis, long planes, float* out) {
  ProfScope ps_(st, PC_OP_RESIZE, (double)planes * other * ((double)n_in + n_out) * sizeof(float), "resize_adj", planes, n_in, n_out, axis);
  hipLaunchKernelGGL(resize_axis_adj_kernel, dim3(pw_grid(planes * (long)n_in * other)), dim3(256), 0, st, g, w, fov, taps,
                     n_in, n_out, other, axis, planes, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

// --------------------------------------------------------------------- Haar DWT ----
// One thread = one 8x8 input block = all three levels in registers; Mallat layout
// (pywt coeffs_to_array places by key: cA3 top-left; per level 'ad' TOP-RIGHT, 'da' BOTTOM-LEFT, 'dd' bottom-right -- put_sub / get_sub below;
// pinned against PyWavelets 1.1.1, tests/test_thirdparty_pins.py).
#define HS_ 0.70710678118654752440f
template <int S>
__device__ inline void haar_step(const float (&x)[2 * S][2 * S], float (&aa)[S][S], float (&da)[S][S], float (&ad)[S][S],
                                 float (&dd)[S][S]) {
#pragma unroll
  for (int i = 0; i < S; ++i)
#pragma unroll
    for (int j = 0; j < S; ++j) {
      float ah0 = (x[2 * i][2 * j] + x[2 * i + 1][2 * j]) * HS_, ah1 = (x[2 * i][2 * j + 1] + x[2 * i + 1][2 * j + 1]) * HS_;
      float dh0 = (x[2 * i][2 * j] - x[2 * i + 1][2 * j]) * HS_, dh1 = (x[2 * i][2 * j + 1] - x[2 * i + 1][2 * j + 1]) * HS_;
      aa[i][j] = (ah0 + ah1) * HS_; ad[i][j] = (ah0 - ah1) * HS_;
      da[i][j] = (dh0 + dh1) * HS_; dd[i][j] = (dh0 - dh1) * HS_;
    }
}
template <int S>
__device__ inline void haar_istep(const float (&aa)[S][S], const float (&da)[S][S], const float (&ad)[S][S],
                                  const float (&dd)[S][S], float (&x)[2 * S][2 * S]) {
#pragma unroll
  for (int i = 0; i < S; ++i)
#pragma unroll
    for (int j = 0; j < S; ++j) {
      float ah0 = (aa[i][j] + ad[i][j]) * HS_, ah1 = (aa[i][j] - ad[i][j]) * HS_;
      float dh0 = (da[i][j] + dd[i][j]) * HS_, dh1 = (da[i][j] - dd[i][j]) * HS_;
      x[2 * i][2 * j] = (ah0 + dh0) * HS_; x[2 * i + 1][2 * j] = (ah0 - dh0) * HS_;
      x[2 * i][2 * j + 1] = (ah1 + dh1) * HS_; x[2 * i + 1][2 * j + 1] = (ah1 - dh1) * HS_;
    }
}
template <int S>
__device__ inline void put_sub(float* o, int N, int size, int bi, int bj, const float (&da)[S][S], const float (&ad)[S][S],
                               const float (&dd)[S][S]) {
#pragma unroll
  for (int i = 0; i < S; ++i)
#pragma unroll
    for (int j = 0; j < S; ++j) {
      int r = bi * S + i, c = bj * S + j;
      // pywt.coeffs_to_array places a detail block by its dwtn KEY: first letter = axis -2 (rows), second = axis -1 (columns); 'a' -> the
      // leading half of that axis, 'd' -> the trailing half.  So 'ad' (approximation down the rows, detail along the columns: pywt's cV) sits
      // TOP-RIGHT and 'da' (pywt's cH) BOTTOM-LEFT -- pinned against PyWavelets 1.1.1 (tests/golden/thirdparty_pins.npz).  Rounds 1 - 4 had
      // the two blocks exchanged (restated from the documentation's cH / cV figure without the package at hand).
      o[(long)r * N + size + c] = ad[i][j];
      o[(long)(size + r) * N + c] = da[i][j];
      o[(long)(size + r) * N + size + c] = dd[i][j];
    }
}
template <int S>
__device__ inline void get_sub(const float* o, int N, int size, int bi, int bj, float (&da)[S][S], float (&ad)[S][S],
                               float (&dd)[S][S]) {
#pragma unroll
  for (int i = 0; i < S; ++i)
#pragma unroll
    for (int j = 0; j < S; ++j) {
      int r = bi * S + i, c = bj * S + j;
      ad[i][j] = o[(long)r * N + size + c];          // top-right: 'ad' (see put_sub)
      da[i][j] = o[(long)(size + r) * N + c];          // bottom-left: 'da'
      dd[i][j] = o[(long)(size + r) * N + size + c];
    }
}
__global__ void dwt_haar3_kernel(const float* __restrict__ x, int N, long planes, float* __restrict__ out) {
  const int nb = N / 8;
  GRID_STRIDE(i, planes * nb * nb) {
    long p = i / (nb * nb); int b = (int)(i % (nb * nb)); int bi = b / nb, bj = b % nb;
    const float* xp = x + p * (long)N * N; float* op = out + p * (long)N * N;
    float v[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float4 a = *(const float4*)(xp + (long)(bi * 8 + r) * N + bj * 8), c = *(const float4*)(xp + (long)(bi * 8 + r) * N + bj * 8 + 4);
      v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w; v[r][4] = c.x; v[r][5] = c.y; v[r][6] = c.z; v[r][7] = c.w;
    }
    float a1[4][4], da1[4][4], ad1[4][4], dd1[4][4];
    haar_step<4>(v, a1, da1, ad1, dd1);
    put_sub<4>(op, N, N / 2, bi, bj, da1, ad1, dd1);
    float a2[2][2], da2[2][2], ad2[2][2], dd2[2][2];
    haar_step<2>(a1, a2, da2, ad2, dd2);
    put_sub<2>(op, N, N / 4, bi, bj, da2, ad2, dd2);
    float a3[1][1], da3[1][1], ad3[1][1], dd3[1][1];
    haar_step<1>(a2, a3, da3, ad3, dd3);
    put_sub<1>(op, N, N / 8, bi, bj, da3, ad3, dd3);
    op[(long)bi * N + bj] = a3[0][0];
  }
}
int dwt_haar3(hipStream_t st, const float* x, int N, long planes, float* out) {
  ProfScope ps_(st, PC_OP_DWT, (double)planes * N * N * sizeof(float) * 2, "dwt", planes, N);
  KDIP_REQUIRE(N % 8 == 0, "dwt: N=%d must be a multiple of 8", N);
  hipLaunchKernelGGL(dwt_haar3_kernel, dim3(pw_grid(planes * (N / 8) * (N / 8))), dim3(256), 0, st, x, N, planes, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void idwt_haar3_kernel(const float* __restrict__ c, int N, long planes, float* __restrict__ out) {
  const int nb = N / 8;
  GRID_STRIDE(i, planes * nb * nb) {
    long p = i / (nb * nb); int b = (int)(i % (nb * nb)); int bi = b / nb, bj = b % nb;
    const float* cp = c + p * (long)N * N; float* op = out + p * (long)N * N;
    float a3[1][1], da3[1][1], ad3[1][1], dd3[1][1];
    a3[0][0] = cp[(long)bi * N + bj];
    get_sub<1>(cp, N, N / 8, bi, bj, da3, ad3, dd3);
    float a2[2][2], da2[2][2], ad2[2][2], dd2[2][2];
    haar_istep<1>(a3, da3, ad3, dd3, a2);
    get_sub<2>(cp, N, N / 4, bi, bj, da2, ad2, dd2);
    float a1[4][4], da1[4][4], ad1[4][4], dd1[4][4];
    haar_istep<2>(a2, da2, ad2, dd2, a1);
    get_sub<4>(cp, N, N / 2, bi, bj, da1, ad1, dd1);
    float v[8][8];
    haar_istep<4>(a1, da1, ad1, dd1, v);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      *(float4*)(op + (long)(bi * 8 + r) * N + bj * 8) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
      *(float4*)(op + (long)(bi * 8 + r) * N + bj * 8 + 4) = make_float4(v[r][4], v[r][5], v[r][6], v[r][7]);
    }
  }
}
int idwt_haar3(hipStream_t st, const float* c, int N, long planes, float* out) {
  ProfScope ps_(st, PC_OP_DWT, (double)planes * N * N * sizeof(float) * 2, "idwt", planes, N);
  KDIP_REQUIRE(N % 8 == 0, "idwt: N=%d must be a multiple of 8", N);
  hipLaunchKernelGGL(idwt_haar3_kernel, dim3(pw_grid(planes * (N / 8) * (N / 8))), dim3(256), 0, st, c, N, planes, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

// 3-point orthonormal DCT-II over the channel axis (scipy dctn runs over *all* axes, utils.py:94)
__global__ void dct3_kernel(const float* __restrict__ x, long HW, int B, int inverse, float* __restrict__ out) {
  const float s3 = 0.57735026918962576451f, s2 = 0.70710678118654752440f, s6 = 0.40824829046386301637f;
  GRID_STRIDE(i, (long)B * HW) {
    long b = i / HW, p = i % HW;
    const float* xb = x + b * 3 * HW + p; float* ob = out + b * 3 * HW + p;
    float a = xb[0], c = xb[HW], d = xb[2 * HW];
    if (!inverse) { ob[0] = (a + c + d) * s3; ob[HW] = (a - d) * s2; ob[2 * HW] = (a - 2.f * c + d) * s6; }
    else { ob[0] = a * s3 + c * s2 + d * s6; ob[HW] = a * s3 - 2.f * d * s6; ob[2 * HW] = a * s3 - c * s2 + d * s6; }
  }
}
int dct3_channels(hipStream_t st, const float* x, long HW, int B, int inverse, float* out) {
  hipLaunchKernelGGL(dct3_kernel, dim3(pw_grid((long)B * HW)), dim3(256), 0, st, x, HW, B, inverse, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

// --------------------------------------------------------------- guidance epilogues ----
__global__ void x0_v1_kernel(const float* __restrict__ uo, const float* __restrict__ x, int B, long HW, X0Params p,
                             float* __restrict__ x0, float* __restrict__ xraw, float* __restrict__ var) {
  GRID_STRIDE(i, (long)B * 3 * HW) {
    long b = i / (3 * HW); long r = i % (3 * HW);
    float eps = uo[b * 6 * HW + r], v = uo[b * 6 * HW + 3 * HW + r];
    float xin = x[i] * p.c_in;
    float raw = p.sqrt_recip * xin - p.sqrt_recipm1 * eps;
    x0[i] = fminf(fmaxf(raw, -1.f), 1.f);
    if (xraw) xraw[i] = raw;
    if (p.want_var) {
      float frac = (v + 1.f) / 2.f;
      float mv = expf(frac * p.log_beta + (1.f - frac) * p.log_post_var);
      var[i] = fmaxf((mv - p.post_var) / (p.coef1 * p.coef1), 1e-6f);
    }
  }
}
int x0_epilogue_v1(hipStream_t st, const float* unet_out, const float* x, int B, long HW, X0Params p, float* x0_mean,
                   float* x0_raw, float* var) {
  ProfScope ps_(st, PC_OP_POINTWISE, (double)B * HW * sizeof(float) * (6 + 3 + 3 + 3 + (var ? 3 : 0)), "x0_epilogue_v1", B, HW);
  hipLaunchKernelGGL(x0_v1_kernel, dim3(pw_grid((long)B * 3 * HW)), dim3(256), 0, st, unet_out, x, B, HW, p, x0_mean, x0_raw, var);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void x0_v2_kernel(const float* __restrict__ uo, const float* __restrict__ co, const float* __restrict__ x, int B,
                             long HW, float sigma, int want_var, float* __restrict__ x0, float* __restrict__ xv,
                             float* __restrict__ tv) {
  GRID_STRIDE(i, (long)B * 3 * HW) {
    long b = i / (3 * HW); long r = i % (3 * HW);
    float eps = uo[b * 6 * HW + r];
    x0[i] = eps * (-sigma) + x[i];
    if (want_var) {
      float s2 = sigma * sigma;
      xv[i] = expf(co[b * 6 * HW + r]) * s2;
      tv[i] = expf(co[b * 6 * HW + 3 * HW + r]) * s2;
    }
  }
}
int x0_epilogue_v2(hipStream_t st, const float* unet_out, const float* cov_out, const float* x, int B, long HW, float sigma,
                   int want_var, float* x0_mean, float* x0_var, float* theta_var) {
  hipLaunchKernelGGL(x0_v2_kernel, dim3(pw_grid((long)B * 3 * HW)), dim3(256), 0, st, unet_out, cov_out, x, B, HW, sigma,
                     want_var, x0_mean, x0_var, theta_var);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void cot_v1_kernel(const float* __restrict__ gh, const float* __restrict__ xraw, int B, long HW, float srm1,
                              float* __restrict__ cot, float* __restrict__ graw) {
  GRID_STRIDE(i, (long)B * 3 * HW) {
    long b = i / (3 * HW); long r = i % (3 * HW);
    float raw = xraw[i];
    float g = (raw >= -1.f && raw <= 1.f) ? gh[i] : 0.f;    // clamp backward, bounds inclusive
    graw[i] = g;
    cot[b * 6 * HW + r] = -srm1 * g;
    cot[b * 6 * HW + 3 * HW + r] = 0.f;
  }
}
int vjp_cotangent_v1(hipStream_t st, const float* ghat, const float* x0_raw, int B, long HW, float sqrt_recipm1, float* cot6,
                     float* g_raw) {
  ProfScope ps_(st, PC_OP_POINTWISE, (double)B * HW * sizeof(float) * (3 + 3 + 6 + 3), "cotangent_v1", B, HW);
  hipLaunchKernelGGL(cot_v1_kernel, dim3(pw_grid((long)B * 3 * HW)), dim3(256), 0, st, ghat, x0_raw, B, HW, sqrt_recipm1, cot6, g_raw);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void cot_v2_kernel(const float* __restrict__ gh, int B, long HW, float* __restrict__ cot) {
  GRID_STRIDE(i, (long)B * 3 * HW) {
    long b = i / (3 * HW); long r = i % (3 * HW);
    cot[b * 6 * HW + r] = gh[i];
    cot[b * 6 * HW + 3 * HW + r] = 0.f;
  }
}
int vjp_cotangent_v2(hipStream_t st, const float* ghat, int B, long HW, float* cot6) {
  hipLaunchKernelGGL(cot_v2_kernel, dim3(pw_grid((long)B * 3 * HW)), dim3(256), 0, st, ghat, B, HW, cot6);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void combine_kernel(const float* x0, const float* gd, float a, const float* uv, float b, float coef, long n, float* hat) {
  GRID_STRIDE(i, n) {
    float s = (gd ? a * gd[i] : 0.f) + (uv ? b * uv[i] : 0.f);
    hat[i] = fminf(fmaxf(x0[i] + coef * s, -1.f), 1.f);
  }
}
int guidance_combine(hipStream_t st, const float* x0_mean, const float* g_direct, float a, const float* unet_vjp, float b,
                     float coef, long n, float* hat) {
  ProfScope ps_(st, PC_OP_POINTWISE, (double)n * sizeof(float) * (3 + (unet_vjp ? 1 : 0)), "combine", n);
  hipLaunchKernelGGL(combine_kernel, dim3(pw_grid(n)), dim3(256), 0, st, x0_mean, g_direct, a, unet_vjp, b, coef, n, hat);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void clamp_kernel(const float* x, long n, float* out) { GRID_STRIDE(i, n) out[i] = fminf(fmaxf(x[i], -1.f), 1.f); }
int clamp_pm1(hipStream_t st, const float* x, long n, float* out) {
  hipLaunchKernelGGL(clamp_kernel, dim3(pw_grid(n)), dim3(256), 0, st, x, n, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

// --------------------------------------------------------------- per-sample reductions ----
// One block per sample, fixed thread -> element assignment, xor-butterfly wave sums, wave partials added in wave order: the same
// additions in the same order on every run (the CG trajectory and the DPS norm are reproducible bit for bit; a multi-block
// reduction through fp64 atomics was not).  ~10 us for 196 608 elements: 2 dots per CG iteration, < 0.5 % of a guided call.
__global__ __launch_bounds__(1024) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, long per, double* __restrict__ out) {
  __shared__ double sh[16];
  const int bidx = blockIdx.x;
  const float* ap = a + (long)bidx * per; const float* bp = b + (long)bidx * per;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const bool vec = (per % 4 == 0) && (((uintptr_t)ap | (uintptr_t)bp) % 16 == 0);
  if (vec) {
    const float4* a4 = (const float4*)ap; const float4* b4 = (const float4*)bp;
    for (long i = threadIdx.x; i < per / 4; i += 1024) {
      const float4 u = a4[i], v = b4[i];
      s0 += u.x * v.x; s1 += u.y * v.y; s2 += u.z * v.z; s3 += u.w * v.w;
    }
  } else {
    for (long i = threadIdx.x; i < per; i += 1024) s0 += ap[i] * bp[i];
  }
  double d = wave_sum_d(((double)s0 + (double)s1) + ((double)s2 + (double)s3));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sh[w];
    out[bidx] = t;
  }
}
int cg_dot(hipStream_t st, const float* a, const float* b, int B, long per, double* out) {
  ProfScope ps_(st, PC_OP_POINTWISE, (double)B * per * sizeof(float) * 2, "cg_dot", B, per);
  hipLaunchKernelGGL(dot_kernel, dim3(B), dim3(1024), 0, st, a, b, per, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void sqrt_d2f_kernel(const double* in, int B, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) out[i] = (float)sqrt(in[i]);
}
int norm_per_sample(hipStream_t st, const float* x, int B, long per, float* out, double* tmp) {
  int rc = cg_dot(st, x, x, B, per, tmp);
  if (rc) return rc;
  hipLaunchKernelGGL(sqrt_d2f_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, tmp, B, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void scale_inv_kernel(const float* x, const float* nrm, float zeta, long per, long n, float* out) {
  GRID_STRIDE(i, n) out[i] = x[i] * (zeta / nrm[i / per]);
}
int scale_per_sample_inv(hipStream_t st, const float* x, const float* nrm, float zeta, int B, long per, float* out) {
  hipLaunchKernelGGL(scale_inv_kernel, dim3(pw_grid((long)B * per)), dim3(256), 0, st, x, nrm, zeta, per, (long)B * per, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

// ----------------------------------------------------------------------- sampler ----
__global__ void add_noise_kernel(const float* x, const float* eps, float s, long n, float* out) { GRID_STRIDE(i, n) out[i] = x[i] + eps[i] * s; }
int sampler_add_noise(hipStream_t st, const float* x, const float* eps, float s, long n, float* out) {
  hipLaunchKernelGGL(add_noise_kernel, dim3(pw_grid(n)), dim3(256), 0, st, x, eps, s, n, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void euler_kernel(const float* x, const float* den, float sh, float dt, long n, float* out) {
  GRID_STRIDE(i, n) { float d = (x[i] - den[i]) / sh; out[i] = x[i] + d * dt; }
}
int sampler_euler(hipStream_t st, const float* x, const float* den, float sigma_hat, float dt, long n, float* out) {
  ProfScope ps_(st, PC_OP_POINTWISE, (double)n * sizeof(float) * 3, "sampler_euler", n);
  hipLaunchKernelGGL(euler_kernel, dim3(pw_grid(n)), dim3(256), 0, st, x, den, sigma_hat, dt, n, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void heun_kernel(const float* x, const float* d1, const float* x2, const float* d2, float sh, float sn, float dt, long n, float* out) {
  GRID_STRIDE(i, n) {
    float a = (x[i] - d1[i]) / sh, b = (x2[i] - d2[i]) / sn;
    out[i] = x[i] + (a + b) / 2.f * dt;
  }
}
int sampler_heun(hipStream_t st, const float* x, const float* den1, const float* x2, const float* den2, float sigma_hat,
                 float sigma_next, float dt, long n, float* out) {
  ProfScope ps_(st, PC_OP_POINTWISE, (double)n * sizeof(float) * 5, "sampler_heun", n);
  hipLaunchKernelGGL(heun_kernel, dim3(pw_grid(n)), dim3(256), 0, st, x, den1, x2, den2, sigma_hat, sigma_next, dt, n, out);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

// ---------------------------------------------------------------------------- CG ----
// scipy.sparse.linalg.cg (legacy tol): stop when ||r|| < tol*||b||, checked at the top of
// each iteration; every sample is its own solve (frozen once converged).
__global__ void cg_init_kernel(CgState s, int B, float tol) {
  int b = threadIdx.x;
  if (b == 0) *s.any_active = 1;
  if (b >= B) return;
  double bb = s.rr[b];                         // b.b
  s.atol2[b] = (double)tol * (double)tol * bb;
  s.active[b] = bb > 0.0 ? 1 : 0;
  s.iters[b] = 0; s.rho_prev[b] = 1.0; s.alpha[b] = 0.f; s.beta[b] = 0.f;
}
int cg_init(hipStream_t st, CgState s, int B, float tol) {
  KDIP_REQUIRE(B <= 1024, "cg: batch %d > 1024", B);
  hipLaunchKernelGGL(cg_init_kernel, dim3(1), dim3(1024), 0, st, s, B, tol);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void cg_step_a_kernel(CgState s, int B, int it, int count_unconverged) {
  __shared__ int any;
  int b = threadIdx.x;
  if (b == 0) any = 0;
  __syncthreads();
  if (b < B) {
    double rr = s.rr[b];
    int act = s.active[b] && !(rr < s.atol2[b]);
    s.active[b] = act;
    s.beta[b] = (act && it > 0) ? (float)(rr / s.rho_prev[b]) : 0.f;
    if (act) { s.iters[b] += 1; atomicOr(&any, 1); }
  }
  __syncthreads();
  if (b == 0) { *s.any_active = any; if (count_unconverged && any) *s.unconverged += 1; }
}
int cg_step_a(hipStream_t st, CgState s, int B, int it, int count_unconverged) {
  hipLaunchKernelGGL(cg_step_a_kernel, dim3(1), dim3(1024), 0, st, s, B, it, count_unconverged);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void cg_update_p_kernel(CgState s, const float* r, float* p, long per, long n) {
  GRID_STRIDE(i, n) { int b = (int)(i / per); if (s.active[b]) p[i] = r[i] + s.beta[b] * p[i]; }
}
int cg_update_p(hipStream_t st, CgState s, const float* r, float* p, int B, long per) {
  ProfScope ps_(st, PC_OP_POINTWISE, (double)B * per * sizeof(float) * 3, "cg_update_p", B, per);
  hipLaunchKernelGGL(cg_update_p_kernel, dim3(pw_grid((long)B * per)), dim3(256), 0, st, s, r, p, per, (long)B * per);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void cg_step_b_kernel(CgState s, int B) {
  int b = threadIdx.x;
  if (b >= B) return;
  if (s.active[b]) { s.alpha[b] = (float)(s.rr[b] / s.pq[b]); s.rho_prev[b] = s.rr[b]; }
  else s.alpha[b] = 0.f;
}
int cg_step_b(hipStream_t st, CgState s, int B) {
  hipLaunchKernelGGL(cg_step_b_kernel, dim3(1), dim3(1024), 0, st, s, B);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}
__global__ void cg_update_xr_kernel(CgState s, float* x, float* r, const float* p, const float* q, long per, long n) {
  GRID_STRIDE(i, n) {
    int b = (int)(i / per);
    if (s.active[b]) { float a = s.alpha[b]; x[i] += a * p[i]; r[i] -= a * q[i]; }
  }
}
int cg_update_xr(hipStream_t st, CgState s, float* x, float* r, const float* p, const float* q, int B, long per) {
  ProfScope ps_(st, PC_OP_POINTWISE, (double)B * per * sizeof(float) * 6, "cg_update_xr", B, per);
  hipLaunchKernelGGL(cg_update_xr_kernel, dim3(pw_grid((long)B * per)), dim3(256), 0, st, s, x, r, p, q, per, (long)B * per);
  KDIP_LAUNCH_CHECK(); return KDIP_OK;
}

}  // namespace kdip
