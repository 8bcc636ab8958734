// 2-D FFT (N in {64, 256}) in LDS and the frequency-domain point-wise kernels of the
// closed-form / CG mat-solvers.
//
// Replaces torch.fft.fft2/ifft2 at condition/measurements.py:118,142,155,181,195 and
// condition/condition.py:357,372-384,409-437, `p2o` (condition/diffpir_utils/utils_sisr.py:22-41)
// and the OTF algebra `F(r)/(sigma_s^2 + v|FB|^2) * conj(FB)`, `splits`-mean + `repeat`.
//
// Layout: planes of N x N, complex as float2.  One pass transforms one axis: a block
// stages LPB lines (rows, or LPB adjacent columns => each global row access is a
// contiguous LPB*8-byte run) in LDS, runs an in-LDS radix-2 DIT (bit-reversed scatter on
// load, twiddles from a host-computed double-precision table), and streams the result
// back.  Conventions follow torch: forward unnormalised, inverse scaled 1/(N*N).
// HBM-bound: 2 passes x (8 B read + 8 B write) per complex element per 2-D transform.
#include "common.h"
#include "fftops.h"

namespace kdip {

template <int N>
__device__ inline int bitrev(int i) {
  constexpr int LG = (N == 256) ? 8 : (N == 64 ? 6 : (N == 16 ? 4 : 0));
  return (int)(__brev((unsigned)i) >> (32 - LG));
}

// in/out: complex planes.  axis 1: lines are rows (element stride 1, line stride N);
// axis 0: lines are columns (element stride N, line stride 1).
// REAL_IN: input is a real plane (float).  REAL_OUT: write only the real part (float).
template <int N, int LPB, bool REAL_IN, bool REAL_OUT>
__global__ __launch_bounds__(256) void fft_axis_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                        const float2* __restrict__ tw, int axis, int inverse,
                                                        float scale) {
  __shared__ float2 buf[LPB][N + 1];
  const int tid = threadIdx.x;
  const long plane = blockIdx.y;
  const int line0 = blockIdx.x * LPB;
  const long pbase = plane * (long)N * N;
  // ---- load (bit-reversed along the line)
  for (int e = tid; e < LPB * N; e += 256) {
    int l, k;
    long g;
    if (axis == 1) { l = e / N; k = e % N; g = pbase + (long)(line0 + l) * N + k; }
    else { k = e / LPB; l = e % LPB; g = pbase + (long)k * N + (line0 + l); }
    float2 v;
    if (REAL_IN) { v.x = ((const float*)in)[g]; v.y = 0.f; }
    else v = ((const float2*)in)[g];
    buf[l][bitrev<N>(k)] = v;
  }
  __syncthreads();
  // ---- butterflies
  constexpr int HALF = N / 2;
  for (int m = 2; m <= N; m <<= 1) {
    const int hm = m >> 1, tstep = (256 / m) * (N == 256 ? 1 : 1);
    for (int e = tid; e < LPB * HALF; e += 256) {
      int l = e / HALF, j = e % HALF;
      int grp = j / hm, pos = j % hm;
      int i0 = grp * m + pos, i1 = i0 + hm;
      // twiddle exp(-2 pi i pos / m) from the 256-point table tw[k] = exp(-2 pi i k / 256)
      float2 w = tw[pos * (256 / m)];
      if (inverse) w.y = -w.y;
      float2 a = buf[l][i0], b = buf[l][i1];
      float2 t = make_float2(b.x * w.x - b.y * w.y, b.x * w.y + b.y * w.x);
      buf[l][i0] = make_float2(a.x + t.x, a.y + t.y);
      buf[l][i1] = make_float2(a.x - t.x, a.y - t.y);
    }
    (void)tstep;
    __syncthreads();
  }
  // ---- store
  for (int e = tid; e < LPB * N; e += 256) {
    int l, k;
    long g;
    if (axis == 1) { l = e / N; k = e % N; g = pbase + (long)(line0 + l) * N + k; }
    else { k = e / LPB; l = e % LPB; g = pbase + (long)k * N + (line0 + l); }
    float2 v = buf[l][k];
    if (REAL_OUT) ((float*)out)[g] = v.x * scale;
    else ((float2*)out)[g] = make_float2(v.x * scale, v.y * scale);
  }
}

template <int N>
static int fft2_N(hipStream_t st, const float2* tw, const void* in, int real_in, float2* tmp, void* out, int real_out,
                  long planes, int inverse) {
  constexpr int LPB = (N == 256) ? 8 : 16;
  dim3 grid(N / LPB, (unsigned)planes);
  const float scale = inverse ? 1.f / ((float)N * N) : 1.f;
  // pass 1: rows (axis 1): in -> tmp (complex)
  if (real_in)
    hipLaunchKernelGGL((fft_axis_kernel<N, LPB, true, false>), grid, dim3(256), 0, st, in, (void*)tmp, tw, 1, inverse, 1.f);
  else
    hipLaunchKernelGGL((fft_axis_kernel<N, LPB, false, false>), grid, dim3(256), 0, st, in, (void*)tmp, tw, 1, inverse, 1.f);
  KDIP_LAUNCH_CHECK();
  // pass 2: columns (axis 0): tmp -> out
  if (real_out)
    hipLaunchKernelGGL((fft_axis_kernel<N, LPB, false, true>), grid, dim3(256), 0, st, (const void*)tmp, out, tw, 0, inverse, scale);
  else
    hipLaunchKernelGGL((fft_axis_kernel<N, LPB, false, false>), grid, dim3(256), 0, st, (const void*)tmp, out, tw, 0, inverse, scale);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

int fft2(hipStream_t st, const float2* tw256, int N, const void* in, int real_in, float2* tmp, void* out, int real_out,
         long planes, int inverse) {
  ProfScope ps_(st, PC_OP_FFT, (double)planes * N * N * ((real_in ? 4.0 : 8.0) + 8.0 + 8.0 + (real_out ? 4.0 : 8.0)), "fft2", planes, N, real_in, real_out);      // two axis passes: in -> tmp -> out
  if (N == 256) return fft2_N<256>(st, tw256, in, real_in, tmp, out, real_out, planes, inverse);
  if (N == 64) return fft2_N<64>(st, tw256, in, real_in, tmp, out, real_out, planes, inverse);
  if (N == 16) return fft2_N<16>(st, tw256, in, real_in, tmp, out, real_out, planes, inverse);
  return set_error(KDIP_ERR_UNSUPPORTED, "fft2: N=%d (supported: 16, 64, 256)", N);
}

void make_twiddles256(float2* host) {
  for (int k = 0; k < 128; ++k) {
    double a = -2.0 * M_PI * (double)k / 256.0;
    host[k] = make_float2((float)cos(a), (float)sin(a));
  }
}

// --------------------------------------------------------------- point-wise kernels ----
static inline int pw_grid(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

// X[p] *= (conj? conj(FB) : FB), FB broadcast over planes
__global__ void cmul_otf_kernel(float2* __restrict__ X, const float2* __restrict__ FB, long nn, long total, int conj) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float2 x = X[i], f = FB[i % nn];
    if (conj) f.y = -f.y;
    X[i] = make_float2(x.x * f.x - x.y * f.y, x.x * f.y + x.y * f.x);
  }
}
int cmul_otf(hipStream_t st, float2* X, const float2* FB, long nn, long planes, int conj) {
  ProfScope ps_(st, PC_OP_OTF, (double)nn * planes * sizeof(float2) * 2 + (double)nn * sizeof(float2), "cmul_otf", planes, nn);
  hipLaunchKernelGGL(cmul_otf_kernel, dim3(pw_grid(nn * planes)), dim3(256), 0, st, X, FB, nn, nn * planes, conj);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// R = R / (s2 + v * F2B) * conj(FB)      (condition/condition.py:357)
__global__ void otf_solve_kernel(float2* __restrict__ R, const float2* __restrict__ FB, long nn, long total, float s2,
                                 float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float2 r = R[i], f = FB[i % nn];
    float f2 = f.x * f.x + f.y * f.y;
    float d = 1.f / (s2 + v * f2);
    r.x *= d; r.y *= d;
    R[i] = make_float2(r.x * f.x + r.y * f.y, r.y * f.x - r.x * f.y);
  }
}
int otf_solve(hipStream_t st, float2* R, const float2* FB, long nn, long planes, float s2, float v) {
  ProfScope ps_(st, PC_OP_OTF, (double)nn * planes * sizeof(float2) * 2 + (double)nn * sizeof(float2), "otf_solve", planes, nn);
  hipLaunchKernelGGL(otf_solve_kernel, dim3(pw_grid(nn * planes)), dim3(256), 0, st, R, FB, nn, nn * planes, s2, v);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// invW[n/sf][n/sf] = mean over the sf x sf aliases of |FB|^2   (utils_sisr.splits + mean, condition.py:409)
__global__ void sr_invw_kernel(const float2* __restrict__ FB, int N, int sf, float* __restrict__ invW) {
  int n = N / sf;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  int r = i / n, c = i % n;
  float s = 0.f;
  // splits(): stack of chunks along dim 2 then cat along dim 3 -> alias order irrelevant for the mean,
  // but keep the reference's summation order (row-chunk fastest within column-chunk)
  for (int cc = 0; cc < sf; ++cc)
    for (int rc = 0; rc < sf; ++rc) {
      float2 f = FB[(long)(rc * n + r) * N + (cc * n + c)];
      s += f.x * f.x + f.y * f.y;
    }
  invW[i] = s / (float)(sf * sf);
}
int sr_invw(hipStream_t st, const float2* FB, int N, int sf, float* invW) {
  int n = N / sf;
  hipLaunchKernelGGL(sr_invw_kernel, dim3(cdiv((long)n * n, 256)), dim3(256), 0, st, FB, N, sf, invW);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// out[p][R][C] = conj(FB[R][C]) * ( Rs[p][R % n][C % n] / (s2 + v * invW[R % n][C % n]) )   (condition.py:410)
__global__ void sr_solve_tile_kernel(const float2* __restrict__ Rs, const float* __restrict__ invW,
                                     const float2* __restrict__ FB, int N, int sf, long planes, float s2, float v,
                                     float2* __restrict__ out) {
  const int n = N / sf;
  long total = planes * (long)N * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long p = i / ((long)N * N);
    int rc = (int)(i % ((long)N * N));
    int R = rc / N, C = rc % N;
    int r = R % n, c = C % n;
    float2 q = Rs[p * n * n + (long)r * n + c];
    float d = 1.f / (s2 + v * invW[r * n + c]);
    q.x *= d; q.y *= d;
    float2 f = FB[rc];
    out[i] = make_float2(q.x * f.x + q.y * f.y, q.y * f.x - q.x * f.y);
  }
}
int sr_solve_tile(hipStream_t st, const float2* Rs, const float* invW, const float2* FB, int N, int sf, long planes,
                  float s2, float v, float2* out) {
  ProfScope ps_(st, PC_OP_OTF, (double)planes * N * N * sizeof(float2) + (double)planes * (N / sf) * (N / sf) * sizeof(float2), "sr_solve_tile", planes, N, sf);
  hipLaunchKernelGGL(sr_solve_tile_kernel, dim3(pw_grid(planes * (long)N * N)), dim3(256), 0, st, Rs, invW, FB, N, sf,
                     planes, s2, v, out);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// PSF -> padded, centred (rolled) real plane  (p2o before the FFT, utils_sisr.py:33-36)
__global__ void psf_embed_kernel(const float* __restrict__ psf, int kh, int kw, int N, float* __restrict__ plane) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * N) return;
  int r = i / N, c = i % N;
  // rolled[r][c] = padded[(r + kh/2) % N][(c + kw/2) % N]
  int pr = (r + kh / 2) % N, pc = (c + kw / 2) % N;
  plane[i] = (pr < kh && pc < kw) ? psf[pr * kw + pc] : 0.f;
}
int psf_embed(hipStream_t st, const float* psf, int kh, int kw, int N, float* plane) {
  hipLaunchKernelGGL(psf_embed_kernel, dim3(cdiv((long)N * N, 256)), dim3(256), 0, st, psf, kh, kw, N, plane);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

}  // namespace kdip
