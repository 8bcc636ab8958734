// GroupNorm(32) forward / backward for NHWC activations, fused with FiLM and SiLU.
//
// Replaces GroupNorm32 (fp32 statistics, guided_diffusion/nn.py:17-19,93-100) +
// nn.SiLU + the scale-shift FiLM `GN(h)*(1+scale)+shift` (guided_diffusion/unet.py:183-184,
// 207-208,249-253) and their backward.  All kernels are HBM-bound streaming passes:
// 16-byte loads, every lane on consecutive channels of a pixel (NHWC => fully coalesced),
// statistics accumulated fp32 per thread, combined in fp64 (LDS + one global atomic per
// group per block) so the E[x^2]-mean^2 form has no visible cancellation.
#include "common.h"
#include "kernels.h"
#include "det.h"

namespace kdip {

struct GnGeom {
  int VP;        // 16-byte vectors per pixel
  int lanes;     // pixel lanes per block
  int nthreads;  // block size
  int cpg;       // channels per group
};

template <typename T> static GnGeom gn_geom(int C) {
  GnGeom g;
  g.VP = C / TypeInfo<T>::EPV;
  g.nthreads = g.VP <= 256 ? 256 : (g.VP <= 512 ? 512 : 1024);
  g.lanes = g.nthreads / g.VP;
  g.cpg = C / 32;
  return g;
}

// ---------------------------------------------------------------- forward statistics ----
// Fixed-order combine of per-thread partial sums (deterministic modes): every thread parks its EPV (sum 1, sum 2) pairs in LDS as
// [pixel lane][channel][2]; thread (group g, k) then adds group g's lanes x cpg values in index order (fp64) -- block-size and
// arrival-order independent.  The block's 64 results go to slot blockIdx.x of image b's slab row; gn_det_finish_kernel adds the
// slots in a fixed tree and WRITES the image's statistics (det.h).
template <int EPV>
__device__ __forceinline__ void gn_det_combine(const float (&v1)[EPV], const float (&v2)[EPV], bool active, int vi, int pl, int lanes, int C,
                                               int cpg, float* lsh, double* slab) {
  const int tid = threadIdx.x, b = blockIdx.y;
  if (active) {
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      lsh[((long)pl * C + vi * EPV + e) * 2] = v1[e];
      lsh[((long)pl * C + vi * EPV + e) * 2 + 1] = v2[e];
    }
  }
  __syncthreads();
  double* row = slab + ((long)b * gridDim.x + blockIdx.x) * 64;
  if (tid < 64) {
    const int g = tid >> 1, k = tid & 1;
    double a = 0.0;
    for (int l = 0; l < lanes; ++l)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) a += (double)lsh[((long)l * C + c) * 2 + k];
    row[tid] = a;
  }
}
// out[b][64] = sum over the n chunk slots of image b (grid = B blocks of 256 threads): thread (part = tid >> 6, k = tid & 63) adds slots
// part, part + 4, ... (coalesced 512-byte rows, 8 loads in flight), the four parts are then added in order
__global__ __launch_bounds__(256) void gn_det_finish_kernel(const double* __restrict__ slab, int n, double* __restrict__ out) {
  __shared__ double dsh[256];
  const int tid = threadIdx.x, b = blockIdx.x, part = tid >> 6, k = tid & 63;
  const double* r0 = slab + (long)b * n * 64;
  double a = 0.0;
  int j = part;
  for (; j + 28 < n; j += 32) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = r0[(long)(j + 4 * u) * 64 + k];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  for (; j < n; j += 4) a += r0[(long)j * 64 + k];
  dsh[part * 64 + k] = a;
  __syncthreads();
  if (tid < 64) out[(long)b * 64 + tid] = ((dsh[tid] + dsh[64 + tid]) + dsh[128 + tid]) + dsh[192 + tid];
}

template <typename T, bool DET>
__global__ void gn_stats_kernel(const T* __restrict__ x, long ldx, long HW, int C, int VP, int lanes, int cpg,
                                long chunk, double* __restrict__ stats, double* det_slab) {
  constexpr int EPV = TypeInfo<T>::EPV;
  __shared__ double sh[32][2];
  extern __shared__ __attribute__((aligned(16))) float gn_lsh[];      // DET: [lanes][C][2]
  const int tid = threadIdx.x, b = blockIdx.y;
  if (tid < 64) sh[tid >> 1][tid & 1] = 0.0;
  __syncthreads();
  const int vi = tid % VP, pl = tid / VP;
  float s[EPV], ss[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) s[e] = ss[e] = 0.f;
  if (pl < lanes) {
    long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < HW ? p0 + chunk : HW;
    const T* base = x + ((long)b * HW) * ldx + (long)vi * EPV;
    for (long p = p0 + pl; p < p1; p += lanes) {
      uint4 v = *(const uint4*)(base + p * ldx);
      float f[EPV];
      unpack16<T>(v, f);
#pragma unroll
      for (int e = 0; e < EPV; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
    }
    if (DET) {
    } else if (cpg % EPV == 0) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int e = 0; e < EPV; ++e) { a += s[e]; q += ss[e]; }
      int g = (vi * EPV) / cpg;
      atomicAdd(&sh[g][0], (double)a);
      atomicAdd(&sh[g][1], (double)q);
    } else {
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        int g = (vi * EPV + e) / cpg;
        atomicAdd(&sh[g][0], (double)s[e]);
        atomicAdd(&sh[g][1], (double)ss[e]);
      }
    }
  }
  if (DET) {       // (block-uniform)
    gn_det_combine<EPV>(s, ss, pl < lanes, vi, pl, lanes, C, cpg, gn_lsh, det_slab);
    return;
  }
  __syncthreads();
  if (tid < 64) atomicAdd(&stats[((long)b * 32 + (tid >> 1)) * 2 + (tid & 1)], sh[tid >> 1][tid & 1]);
}

static long pick_chunk(long HW, int B) {
  // aim for >= ~1024 blocks, at least 32 pixels per block.  (The floor was 256 pixels until round 6: on the 32^2 / 64^2 maps that left 32 - 128
  // blocks, each walking its 256 pixels with one or two loads in flight per thread -- 56 us for gn_bwd_stats at 8 x 64^2 x 256 and 112 us for
  // gn_stats at 8 x 32^2 x 768, 220 - 1 200 GB/s.)  unet.hip's det_gn_bytes sizes the deterministic slab with the same rule.
  long want = (1024 + B - 1) / B;
  long chunk = (HW + want - 1) / want;
  if (chunk < 32) chunk = 32;
  return chunk;
}

// DetWs given (deterministic modes): fixed-order reduction through det->slab ([B][chunks][64] doubles), statistics WRITTEN by the
// last block of each image (no pre-zeroing needed)
static size_t gn_det_lds(int lanes, int C) { return (size_t)lanes * C * 2 * sizeof(float); }      // [lanes][C][2] floats
static int gn_det_check(const DetWs* det, int B, unsigned chunks) {
  KDIP_REQUIRE(det->slab && (size_t)B * chunks * 64 * sizeof(double) <= det->slab_bytes,
               "groupnorm: deterministic-reduction workspace too small (%d images x %u chunks, slab %zu bytes)", B, chunks, det->slab_bytes);
  return KDIP_OK;
}
int gn_stats(hipStream_t st, DType dt, const void* x, long ldx, int B, long HW, int C, double* stats, int prezeroed, const DetWs* det) {
  KDIP_REQUIRE(C % 32 == 0, "groupnorm: C=%d not a multiple of 32", C);
  if (!prezeroed && !det) KDIP_HIP_CHECK(hipMemsetAsync(stats, 0, sizeof(double) * B * 64, st));
  long chunk = pick_chunk(HW, B);
  dim3 grid(cdiv(HW, chunk), B);
  if (det) {
    KDIP_REQUIRE(dt != DT_BF16, "groupnorm: the deterministic reduction is instantiated for fp32 storage only");
    if (int rc = gn_det_check(det, B, grid.x)) return rc;
  }
  prof_begin(st, PC_GN_STATS, 0, (double)B * HW * C * (dt == DT_BF16 ? 2.0 : 4.0), "gn", B, HW, C, 0);
  if (dt == DT_BF16) {
    KDIP_REQUIRE(C % 8 == 0 && C / 8 <= 1024, "groupnorm: unsupported C=%d", C);
    GnGeom g = gn_geom<bf16_t>(C);
    hipLaunchKernelGGL((gn_stats_kernel<bf16_t, false>), grid, dim3(g.nthreads), 0, st, (const bf16_t*)x, ldx, HW, C, g.VP,
                       g.lanes, g.cpg, chunk, stats, nullptr);
  } else {
    KDIP_REQUIRE(C / 4 <= 1024, "groupnorm: unsupported C=%d", C);
    GnGeom g = gn_geom<float>(C);
    if (det)
      hipLaunchKernelGGL((gn_stats_kernel<float, true>), grid, dim3(g.nthreads), gn_det_lds(g.lanes, C), st, (const float*)x, ldx, HW, C, g.VP,
                         g.lanes, g.cpg, chunk, stats, (double*)det->slab);
    else
      hipLaunchKernelGGL((gn_stats_kernel<float, false>), grid, dim3(g.nthreads), 0, st, (const float*)x, ldx, HW, C, g.VP,
                         g.lanes, g.cpg, chunk, stats, nullptr);
  }
  if (det) hipLaunchKernelGGL(gn_det_finish_kernel, dim3(B), dim3(256), 0, st, (const double*)det->slab, (int)grid.x, stats);
  prof_end(st);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// ---------------------------------------------------------------- statistics of a concat ----
// GroupNorm over cat = [t1 (C1 ch) | t2 (C2 ch)]: when every merged group is a whole number of t1's or t2's groups,
// its (sum, sum of squares) is the sum of the producers' conv-fused statistics -- no pass over the concat buffer.
bool gn_merge_eligible(int C1, int C2) {
  const int C = C1 + C2;
  if (C1 <= 0 || C2 <= 0 || C1 % 32 || C2 % 32 || C % 32) return false;
  const int cpg = C / 32;
  return C1 % cpg == 0 && cpg % (C1 / 32) == 0 && cpg % (C2 / 32) == 0;
}
__global__ void gn_merge_stats_kernel(const double* __restrict__ s1, int C1, const double* __restrict__ s2, int C2, int B,
                                      double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 32) return;
  const int b = i / 32, g = i % 32, cpg = (C1 + C2) / 32, c0 = g * cpg;
  const double* src = c0 < C1 ? s1 + (long)b * 64 : s2 + (long)b * 64;
  const int cpgs = (c0 < C1 ? C1 : C2) / 32, g0 = (c0 < C1 ? c0 : c0 - C1) / cpgs;
  double a = 0, q = 0;
  for (int k = 0; k < cpg / cpgs; ++k) { a += src[(g0 + k) * 2]; q += src[(g0 + k) * 2 + 1]; }
  out[(long)i * 2] = a;
  out[(long)i * 2 + 1] = q;
}
int gn_merge_stats(hipStream_t st, const double* s1, int C1, const double* s2, int C2, int B, double* out) {
  KDIP_REQUIRE(gn_merge_eligible(C1, C2), "groupnorm: statistics of %d + %d channels cannot be merged", C1, C2);
  hipLaunchKernelGGL(gn_merge_stats_kernel, dim3(cdiv((long)B * 32, 256)), dim3(256), 0, st, s1, C1, s2, C2, B, out);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// ------------------------------------------------------------------------- coefficients ----
__global__ void gn_coef_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ film, long film_ld, int B,
                               long HW, int C, float eps, float* __restrict__ coef, float* __restrict__ mr) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  int b = i / C, c = i % C, cpg = C / 32, g = c / cpg;
  double n = (double)HW * cpg;
  double mean = stats[((long)b * 32 + g) * 2] / n;
  double var = stats[((long)b * 32 + g) * 2 + 1] / n - mean * mean;
  if (var < 0) var = 0;
  float rstd = (float)(1.0 / sqrt(var + (double)eps));
  float m = (float)mean;
  float a = rstd * gamma[c];
  float bb = beta[c] - m * a;
  if (film) {
    float sc = 1.f + film[(long)b * film_ld + c], sh = film[(long)b * film_ld + C + c];
    a *= sc;
    bb = bb * sc + sh;
  }
  coef[(long)i * 2] = a;
  coef[(long)i * 2 + 1] = bb;
  if (c % cpg == 0) { mr[((long)b * 32 + g) * 2] = m; mr[((long)b * 32 + g) * 2 + 1] = rstd; }
}

int gn_coef(hipStream_t st, const double* stats, const float* gamma, const float* beta, const float* film, int B,
            long HW, int C, float eps, float* coef, float* mr, long film_ld) {
  hipLaunchKernelGGL(gn_coef_kernel, dim3(cdiv((long)B * C, 256)), dim3(256), 0, st, stats, gamma, beta, film,
                     film_ld ? film_ld : 2L * C, B, HW, C, eps, coef, mr);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// ------------------------------------------------------------------------------ apply ----
// Thread = one fixed 16-byte channel vector (its 2*EPV coefficients stay in registers), walking
// pixels of one image chunk: lanes sweep consecutive channel vectors of a pixel => coalesced.
template <typename T>
__global__ void gn_apply_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ coef, long HW, int C,
                                int VP, int lanes, long chunk, int silu, T* __restrict__ y, long ldy) {
  constexpr int EPV = TypeInfo<T>::EPV;
  const int tid = threadIdx.x, b = blockIdx.y;
  const int vi = tid % VP, pl = tid / VP;
  if (pl >= lanes) return;
  float ca[EPV], cb[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    ca[e] = coef[((long)b * C + vi * EPV + e) * 2];
    cb[e] = coef[((long)b * C + vi * EPV + e) * 2 + 1];
  }
  long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < HW ? p0 + chunk : HW;
  const T* xb = x + ((long)b * HW) * ldx + (long)vi * EPV;
  T* yb = y + ((long)b * HW) * ldy + (long)vi * EPV;
  for (long p = p0 + pl; p < p1; p += lanes) {
    float f[EPV];
    unpack16<T>(*(const uint4*)(xb + p * ldx), f);
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      float z = ca[e] * f[e] + cb[e];
      f[e] = silu ? silu_T<T>(z) : z;
    }
    *(uint4*)(yb + p * ldy) = pack16<T>(f);
  }
}

static long pick_chunk_stream(long HW, int B) {
  long want = (4096 + B - 1) / B;              // ~4096 blocks in flight
  long chunk = (HW + want - 1) / want;
  if (chunk < 64) chunk = 64;
  return chunk;
}

int gn_apply(hipStream_t st, DType dt, const void* x, long ldx, const float* coef, int B, long HW, int C, int silu,
             void* y, long ldy) {
  long chunk = pick_chunk_stream(HW, B);
  dim3 grid(cdiv(HW, chunk), B);
  prof_begin(st, PC_GN_APPLY, 0, 2.0 * B * HW * C * (dt == DT_BF16 ? 2.0 : 4.0), "gn", B, HW, C, 0);
  if (dt == DT_BF16) {
    GnGeom g = gn_geom<bf16_t>(C);
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(g.nthreads), 0, st, (const bf16_t*)x, ldx, coef, HW, C, g.VP,
                       g.lanes, chunk, silu, (bf16_t*)y, ldy);
  } else {
    GnGeom g = gn_geom<float>(C);
    hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(g.nthreads), 0, st, (const float*)x, ldx, coef, HW, C, g.VP,
                       g.lanes, chunk, silu, (float*)y, ldy);
  }
  prof_end(st);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// Downsampling ResBlocks (guided_diffusion/unet.py:236-240: h = avg_pool(SiLU(GN(x))), x = avg_pool(x)):
// one pass over x produces both pooled tensors (instead of apply + two pool kernels = 4.5 tensor passes).
template <typename T>
__global__ void gn_apply_pool2_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ coef, int B, int H, int W,
                                      int C, int VP, int silu, T* __restrict__ yp, long ldy, T* __restrict__ xp, long ldxp) {
  constexpr int EPV = TypeInfo<T>::EPV;
  const int Ho = H / 2, Wo = W / 2;
  const long nvec = (long)B * Ho * Wo * VP;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
    const int vi = (int)(v % VP);
    const long op = v / VP;
    const int ox = (int)(op % Wo);
    const long t = op / Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    const T* p = x + (((long)b * H + 2 * oy) * W + 2 * ox) * ldx + (long)vi * EPV;
    float ca[EPV], cb[EPV], sy[EPV], sx[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      ca[e] = coef[((long)b * C + vi * EPV + e) * 2];
      cb[e] = coef[((long)b * C + vi * EPV + e) * 2 + 1];
      sy[e] = 0.f; sx[e] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float f[EPV];
      unpack16<T>(*(const uint4*)(p + (long)(q >> 1) * W * ldx + (long)(q & 1) * ldx), f);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        const float z = ca[e] * f[e] + cb[e];
        sy[e] += silu ? silu_T<T>(z) : z;
        sx[e] += f[e];
      }
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) { sy[e] *= 0.25f; sx[e] *= 0.25f; }
    *(uint4*)(yp + op * ldy + (long)vi * EPV) = pack16<T>(sy);
    *(uint4*)(xp + op * ldxp + (long)vi * EPV) = pack16<T>(sx);
  }
}

int gn_apply_pool2(hipStream_t st, DType dt, const void* x, long ldx, const float* coef, int B, int H, int W, int C, int silu,
                   void* yp, long ldy, void* xp, long ldxp) {
  const int VP = C / (dt == DT_BF16 ? 8 : 4);
  const long nvec = (long)B * (H / 2) * (W / 2) * VP;
  long g = (nvec + 255) / 256; if (g > 16384) g = 16384; if (g < 1) g = 1;
  prof_begin(st, PC_GN_APPLY, 0, 1.5 * B * H * W * C * (dt == DT_BF16 ? 2.0 : 4.0), "gn_pool", B, (long)H * W, C, 0);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(gn_apply_pool2_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)x, ldx, coef, B, H, W, C,
                       VP, silu, (bf16_t*)yp, ldy, (bf16_t*)xp, ldxp);
  else
    hipLaunchKernelGGL(gn_apply_pool2_kernel<float>, dim3((unsigned)g), dim3(256), 0, st, (const float*)x, ldx, coef, B, H, W, C, VP,
                       silu, (float*)yp, ldy, (float*)xp, ldxp);
  prof_end(st);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// --------------------------------------------------------------------------- backward ----
// z = a*x + b, y = silu(z) | z.  dz = dy*silu'(z) | dy.  xh = (x-mean)*rstd.
// T1 = sum_g a*dz, T2 = sum_g a*dz*xh;   dx = a*dz - T1/N - xh*T2/N   (N = HW*cpg)
template <typename T, bool DET>
__global__ void gn_bwd_stats_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                    const float* __restrict__ coef, const float* __restrict__ mr, long HW, int C,
                                    int VP, int lanes, int cpg, long chunk, int silu, double* __restrict__ sums,
                                    int half_lgW, double* det_slab) {
  extern __shared__ __attribute__((aligned(16))) float gn_lsh[];      // DET: [lanes][C][2]
  // half_lgW >= 0: dy is a half-resolution tensor (adjoint of the 2x2 average pool, unet.py:236-240 backward):
  // dy(p) = 0.25 * dy_half[(y >> 1, x >> 1)], W = 1 << half_lgW, dy batch stride HW / 4
  constexpr int EPV = TypeInfo<T>::EPV;
  __shared__ double sh[32][2];
  const int tid = threadIdx.x, b = blockIdx.y;
  if (tid < 64) sh[tid >> 1][tid & 1] = 0.0;
  __syncthreads();
  const int vi = tid % VP, pl = tid / VP;
  float t1[EPV], t2[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) t1[e] = t2[e] = 0.f;
  if (pl < lanes) {
    float a[EPV], bb[EPV], mean[EPV], rstd[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      int c = vi * EPV + e, g = c / cpg;
      a[e] = coef[((long)b * C + c) * 2];
      bb[e] = coef[((long)b * C + c) * 2 + 1];
      mean[e] = mr[((long)b * 32 + g) * 2];
      rstd[e] = mr[((long)b * 32 + g) * 2 + 1];
    }
    long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < HW ? p0 + chunk : HW;
    const T* xb = x + ((long)b * HW) * ldx + (long)vi * EPV;
    const T* db = dy + ((long)b * (half_lgW >= 0 ? HW >> 2 : HW)) * lddy + (long)vi * EPV;
    const float dscale = half_lgW >= 0 ? 0.25f : 1.f;
    for (long p = p0 + pl; p < p1; p += lanes) {
      float fx[EPV], fd[EPV];
      unpack16<T>(*(const uint4*)(xb + p * ldx), fx);
      const long pd = half_lgW >= 0 ? (((p >> half_lgW) >> 1) << (half_lgW - 1)) + ((p & ((1L << half_lgW) - 1)) >> 1) : p;
      unpack16<T>(*(const uint4*)(db + pd * lddy), fd);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        float z = a[e] * fx[e] + bb[e];
        float dz = (silu ? fd[e] * silu_grad_T<T>(z) : fd[e]) * dscale;
        float adz = a[e] * dz;
        t1[e] += adz;
        t2[e] += adz * (fx[e] - mean[e]) * rstd[e];
      }
    }
    if (!DET) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        int g = (vi * EPV + e) / cpg;
        atomicAdd(&sh[g][0], (double)t1[e]);
        atomicAdd(&sh[g][1], (double)t2[e]);
      }
    }
  }
  if (DET) {       // (block-uniform)
    gn_det_combine<EPV>(t1, t2, pl < lanes, vi, pl, lanes, C, cpg, gn_lsh, det_slab);
    return;
  }
  __syncthreads();
  if (tid < 64) atomicAdd(&sums[((long)b * 32 + (tid >> 1)) * 2 + (tid & 1)], sh[tid >> 1][tid & 1]);
}

int gn_bwd_stats(hipStream_t st, DType dt, const void* x, long ldx, const void* dy, long lddy, const float* coef,
                 const float* mr, int B, long HW, int C, int silu, double* sums, int prezeroed, int half_lgW, const DetWs* det) {
  if (!prezeroed && !det) KDIP_HIP_CHECK(hipMemsetAsync(sums, 0, sizeof(double) * B * 64, st));
  long chunk = pick_chunk(HW, B);
  dim3 grid(cdiv(HW, chunk), B);
  if (det) {
    KDIP_REQUIRE(dt != DT_BF16, "groupnorm: the deterministic reduction is instantiated for fp32 storage only");
    if (int rc = gn_det_check(det, B, grid.x)) return rc;
  }
  prof_begin(st, PC_GN_BWD_STATS, 0, 2.0 * B * HW * C * (dt == DT_BF16 ? 2.0 : 4.0), "gn", B, HW, C, 0);
  if (dt == DT_BF16) {
    GnGeom g = gn_geom<bf16_t>(C);
    hipLaunchKernelGGL((gn_bwd_stats_kernel<bf16_t, false>), grid, dim3(g.nthreads), 0, st, (const bf16_t*)x, ldx,
                       (const bf16_t*)dy, lddy, coef, mr, HW, C, g.VP, g.lanes, g.cpg, chunk, silu, sums, half_lgW, nullptr);
  } else {
    GnGeom g = gn_geom<float>(C);
    if (det)
      hipLaunchKernelGGL((gn_bwd_stats_kernel<float, true>), grid, dim3(g.nthreads), gn_det_lds(g.lanes, C), st, (const float*)x, ldx,
                         (const float*)dy, lddy, coef, mr, HW, C, g.VP, g.lanes, g.cpg, chunk, silu, sums, half_lgW, (double*)det->slab);
    else
      hipLaunchKernelGGL((gn_bwd_stats_kernel<float, false>), grid, dim3(g.nthreads), 0, st, (const float*)x, ldx,
                         (const float*)dy, lddy, coef, mr, HW, C, g.VP, g.lanes, g.cpg, chunk, silu, sums, half_lgW, nullptr);
  }
  if (det) hipLaunchKernelGGL(gn_det_finish_kernel, dim3(B), dim3(256), 0, st, (const double*)det->slab, (int)grid.x, sums);
  prof_end(st);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

#ifndef KDIP_GN_NT
#define KDIP_GN_NT 2             // gn_bwd_apply: non-temporal loads of dy / addends + store of dx (1), and of x (2): 3.87 -> 4.2-4.3 TB/s in the network; 0: plain
#endif
typedef __attribute__((ext_vector_type(4))) unsigned gn_u32x4;
__device__ __forceinline__ uint4 gn_ld(const void* p) {
  if (KDIP_GN_NT) { gn_u32x4 v = __builtin_nontemporal_load((const gn_u32x4*)p); return make_uint4(v[0], v[1], v[2], v[3]); }
  return *(const uint4*)p;
}
__device__ __forceinline__ void gn_st(void* p, uint4 o) {
  if (KDIP_GN_NT) { gn_u32x4 v = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(v, (gn_u32x4*)p); }
  else *(uint4*)p = o;
}

template <typename T>
__global__ void gn_bwd_apply_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                    const float* __restrict__ coef, const float* __restrict__ mr,
                                    const double* __restrict__ sums, long HW, int C, int VP, int lanes, int cpg,
                                    long chunk, int silu, const T* __restrict__ addend, long lda,
                                    const T* __restrict__ addend2, long lda2, T* __restrict__ dx, long lddx, int half_lgW) {
  // half_lgW >= 0: dy and addend are half-resolution tensors (see gn_bwd_stats_kernel); addend2 stays full resolution
  constexpr int EPV = TypeInfo<T>::EPV;
  const int tid = threadIdx.x, b = blockIdx.y;
  const int vi = tid % VP, pl = tid / VP;
  if (pl >= lanes) return;
  const float invN = 1.f / ((float)HW * (float)cpg);
  // dx = a*dz - t1 - xh*t2,  xh = (x-mean)*rstd   ==>   dx = a*dz - (k0 + k1*x)
  float ca[EPV], cb[EPV], k0[EPV], k1[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    int c = vi * EPV + e, g = c / cpg;
    ca[e] = coef[((long)b * C + c) * 2];
    cb[e] = coef[((long)b * C + c) * 2 + 1];
    float mean = mr[((long)b * 32 + g) * 2], rstd = mr[((long)b * 32 + g) * 2 + 1];
    float t1 = (float)sums[((long)b * 32 + g) * 2] * invN, t2 = (float)sums[((long)b * 32 + g) * 2 + 1] * invN;
    k1[e] = rstd * t2;
    k0[e] = t1 - mean * k1[e];
  }
  long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < HW ? p0 + chunk : HW;
  const T* xb = x + ((long)b * HW) * ldx + (long)vi * EPV;
  const long HWd = half_lgW >= 0 ? HW >> 2 : HW;
  const float dscale = half_lgW >= 0 ? 0.25f : 1.f;
  const T* db = dy + ((long)b * HWd) * lddy + (long)vi * EPV;
  const T* ab = addend ? addend + ((long)b * HWd) * lda + (long)vi * EPV : nullptr;
  const T* ab2 = addend2 ? addend2 + ((long)b * HW) * lda2 + (long)vi * EPV : nullptr;
  T* ob = dx + ((long)b * HW) * lddx + (long)vi * EPV;
  for (long p = p0 + pl; p < p1; p += lanes) {
    float fx[EPV], fd[EPV], fa[EPV], fa2[EPV], out[EPV];
    unpack16<T>(KDIP_GN_NT >= 2 ? gn_ld(xb + p * ldx) : *(const uint4*)(xb + p * ldx), fx);
    const long pd = half_lgW >= 0 ? (((p >> half_lgW) >> 1) << (half_lgW - 1)) + ((p & ((1L << half_lgW) - 1)) >> 1) : p;
    unpack16<T>(gn_ld(db + pd * lddy), fd);
    if (ab) unpack16<T>(gn_ld(ab + pd * lda), fa);
    if (ab2) unpack16<T>(gn_ld(ab2 + p * lda2), fa2);
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      float z = ca[e] * fx[e] + cb[e];
      float dz = (silu ? fd[e] * silu_grad_T<T>(z) : fd[e]) * dscale;
      float r = ca[e] * dz - (k0[e] + k1[e] * fx[e]);
      if (ab) r += fa[e] * dscale;
      if (ab2) r += fa2[e];
      out[e] = r;
    }
    gn_st(ob + p * lddx, pack16<T>(out));
  }
}

int gn_bwd_apply(hipStream_t st, DType dt, const void* x, long ldx, const void* dy, long lddy, const float* coef,
                 const float* mr, const double* sums, int B, long HW, int C, int silu, const void* addend, long lda,
                 void* dx, long lddx, const void* addend2, long lda2, int half_lgW) {
  long chunk = pick_chunk_stream(HW, B);
  dim3 grid(cdiv(HW, chunk), B);
  prof_begin(st, PC_GN_BWD_APPLY, 0, (3.0 + (addend ? 1 : 0) + (addend2 ? 1 : 0)) * B * HW * C * (dt == DT_BF16 ? 2.0 : 4.0), "gn", B, HW, C, (addend ? 1 : 0) + (addend2 ? 1 : 0));
  if (dt == DT_BF16) {
    GnGeom g = gn_geom<bf16_t>(C);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, grid, dim3(g.nthreads), 0, st, (const bf16_t*)x, ldx,
                       (const bf16_t*)dy, lddy, coef, mr, sums, HW, C, g.VP, g.lanes, g.cpg, chunk, silu,
                       (const bf16_t*)addend, lda, (const bf16_t*)addend2, lda2, (bf16_t*)dx, lddx, half_lgW);
  } else {
    GnGeom g = gn_geom<float>(C);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, dim3(g.nthreads), 0, st, (const float*)x, ldx,
                       (const float*)dy, lddy, coef, mr, sums, HW, C, g.VP, g.lanes, g.cpg, chunk, silu,
                       (const float*)addend, lda, (const float*)addend2, lda2, (float*)dx, lddx, half_lgW);
  }
  prof_end(st);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// Per-channel constants of the GroupNorm backward  dx = a*dz - (k0 + k1*x)  for the dgrad conv that applies it while staging
// its input patch (conv3.hip, tf 2): out[B][C][4] = (a, b, k0, k1), k1 = rstd*T2/N, k0 = T1/N - mean*k1 (as gn_bwd_apply_kernel).
__global__ void gn_bwd_coef_kernel(const float* __restrict__ coef, const float* __restrict__ mr, const double* __restrict__ sums,
                                   int B, long HW, int C, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C, cpg = C / 32, g = c / cpg;
  const float invN = 1.f / ((float)HW * (float)cpg);
  const float mean = mr[((long)b * 32 + g) * 2], rstd = mr[((long)b * 32 + g) * 2 + 1];
  const float t1 = (float)sums[((long)b * 32 + g) * 2] * invN, t2 = (float)sums[((long)b * 32 + g) * 2 + 1] * invN;
  const float k1 = rstd * t2, k0 = t1 - mean * k1;
  out[i] = make_float4(coef[(long)i * 2], coef[(long)i * 2 + 1], k0, k1);
}
int gn_bwd_coef(hipStream_t st, const float* coef, const float* mr, const double* sums, int B, long HW, int C, float* out) {
  hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(cdiv((long)B * C, 256)), dim3(256), 0, st, coef, mr, sums, B, HW, C, (float4*)out);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// ------------------------------------------------------------- small feature maps ----
// HW <= 256 (the 8x8 and 16x16 levels): the whole (image, group) slab is a few KB, so the
// stats / coef / apply chain (3 launches forward, 2 backward, each launch-latency bound at
// 10-40 us) collapses into one block per (image, group) that reads its slab twice (second
// read from L1/L2).  Same arithmetic as the streaming kernels above.
constexpr int GN_SMALL_MAX_CPG = 128;
#ifndef KDIP_GN_SMALL_HW
#define KDIP_GN_SMALL_HW 256
#endif
constexpr long GN_SMALL_MAX_HW = KDIP_GN_SMALL_HW;   // 32x32 maps (HW = 1024) measured faster on the streaming kernels + conv-fused statistics

__device__ inline void block_sum2_d(double& a, double& b, double (*red)[2]) {
  a = wave_sum_d(a); b = wave_sum_d(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = a; red[w][1] = b; }
  __syncthreads();
  a = 0; b = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a += red[i][0]; b += red[i][1]; }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_fwd_small_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ film,
                                                           long film_ld, int HW, int C, int cpg, float eps, int silu,
                                                           T* __restrict__ y, long ldy, float* __restrict__ coef,
                                                           float* __restrict__ mr) {
  constexpr int EPV = TypeInfo<T>::EPV;
  __shared__ double red[4][2];
  __shared__ float sa[GN_SMALL_MAX_CPG], sb[GN_SMALL_MAX_CPG];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int vpg = cpg / EPV, nvec = HW * vpg;
  const T* xb = x + (long)b * HW * ldx + (long)g * cpg;
  float s = 0.f, ss = 0.f;
  for (int i = tid; i < nvec; i += 256) {
    const int px = i / vpg, v = i - px * vpg;
    float f[EPV];
    unpack16<T>(*(const uint4*)(xb + (long)px * ldx + v * EPV), f);
#pragma unroll
    for (int e = 0; e < EPV; ++e) { s += f[e]; ss += f[e] * f[e]; }
  }
  double ds = s, dss = ss;
  block_sum2_d(ds, dss, red);
  const double n = (double)HW * cpg;
  const double mean = ds / n;
  double var = dss / n - mean * mean;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps)), m = (float)mean;
  if (tid < cpg) {
    const int c = g * cpg + tid;
    float a = rstd * gamma[c];
    float bb = beta[c] - m * a;
    if (film) {
      const float sc = 1.f + film[(long)b * film_ld + c], sh = film[(long)b * film_ld + C + c];
      a *= sc;
      bb = bb * sc + sh;
    }
    sa[tid] = a; sb[tid] = bb;
    coef[((long)b * C + c) * 2] = a;
    coef[((long)b * C + c) * 2 + 1] = bb;
  }
  if (tid == 0) { mr[((long)b * 32 + g) * 2] = m; mr[((long)b * 32 + g) * 2 + 1] = rstd; }
  __syncthreads();
  T* yb = y + (long)b * HW * ldy + (long)g * cpg;
  for (int i = tid; i < nvec; i += 256) {
    const int px = i / vpg, v = i - px * vpg;
    float f[EPV];
    unpack16<T>(*(const uint4*)(xb + (long)px * ldx + v * EPV), f);
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const float z = sa[v * EPV + e] * f[e] + sb[v * EPV + e];
      f[e] = silu ? silu_T<T>(z) : z;
    }
    *(uint4*)(yb + (long)px * ldy + v * EPV) = pack16<T>(f);
  }
}

bool gn_small_eligible(DType dt, long HW, int C) {
  const int cpg = C / 32, epv = dt == DT_BF16 ? 8 : 4;
  return HW <= GN_SMALL_MAX_HW && C % 32 == 0 && cpg % epv == 0 && cpg <= GN_SMALL_MAX_CPG;
}

int gn_fwd_small(hipStream_t st, DType dt, const void* x, long ldx, int B, long HW, int C, const float* gamma,
                 const float* beta, const float* film, long film_ld, float eps, int silu, void* y, long ldy, float* coef,
                 float* mr) {
  KDIP_REQUIRE(gn_small_eligible(dt, HW, C), "groupnorm(small): unsupported shape HW=%ld C=%d", HW, C);
  dim3 grid(32, B);
  prof_begin(st, PC_GN_APPLY, 0, 2.0 * B * HW * C * (dt == DT_BF16 ? 2.0 : 4.0), "gn_small", B, HW, C, 0);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(gn_fwd_small_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, gamma, beta, film,
                       film_ld ? film_ld : 2L * C, (int)HW, C, C / 32, eps, silu, (bf16_t*)y, ldy, coef, mr);
  else
    hipLaunchKernelGGL(gn_fwd_small_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, gamma, beta, film,
                       film_ld ? film_ld : 2L * C, (int)HW, C, C / 32, eps, silu, (float*)y, ldy, coef, mr);
  prof_end(st);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_small_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                                           const float* __restrict__ coef, const float* __restrict__ mr, int HW,
                                                           int C, int cpg, int silu, const T* __restrict__ addend, long lda,
                                                           const T* __restrict__ addend2, long lda2, T* __restrict__ dx,
                                                           long lddx) {
  constexpr int EPV = TypeInfo<T>::EPV;
  __shared__ double red[4][2];
  __shared__ float sa[GN_SMALL_MAX_CPG], sb[GN_SMALL_MAX_CPG];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int vpg = cpg / EPV, nvec = HW * vpg;
  if (tid < cpg) {
    sa[tid] = coef[((long)b * C + g * cpg + tid) * 2];
    sb[tid] = coef[((long)b * C + g * cpg + tid) * 2 + 1];
  }
  const float mean = mr[((long)b * 32 + g) * 2], rstd = mr[((long)b * 32 + g) * 2 + 1];
  __syncthreads();
  const T* xb = x + (long)b * HW * ldx + (long)g * cpg;
  const T* db = dy + (long)b * HW * lddy + (long)g * cpg;
  float t1 = 0.f, t2 = 0.f;
  for (int i = tid; i < nvec; i += 256) {
    const int px = i / vpg, v = i - px * vpg;
    float fx[EPV], fd[EPV];
    unpack16<T>(*(const uint4*)(xb + (long)px * ldx + v * EPV), fx);
    unpack16<T>(*(const uint4*)(db + (long)px * lddy + v * EPV), fd);
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const float a = sa[v * EPV + e], z = a * fx[e] + sb[v * EPV + e];
      const float dz = silu ? fd[e] * silu_grad_T<T>(z) : fd[e];
      const float adz = a * dz;
      t1 += adz;
      t2 += adz * (fx[e] - mean) * rstd;
    }
  }
  double d1 = t1, d2 = t2;
  block_sum2_d(d1, d2, red);
  const float invN = 1.f / ((float)HW * (float)cpg);
  const float k1 = rstd * ((float)d2 * invN), k0 = (float)d1 * invN - mean * k1;
  const T* ab = addend ? addend + (long)b * HW * lda + (long)g * cpg : nullptr;
  const T* ab2 = addend2 ? addend2 + (long)b * HW * lda2 + (long)g * cpg : nullptr;
  T* ob = dx + (long)b * HW * lddx + (long)g * cpg;
  for (int i = tid; i < nvec; i += 256) {
    const int px = i / vpg, v = i - px * vpg;
    float fx[EPV], fd[EPV], fa[EPV], fa2[EPV], out[EPV];
    unpack16<T>(*(const uint4*)(xb + (long)px * ldx + v * EPV), fx);
    unpack16<T>(*(const uint4*)(db + (long)px * lddy + v * EPV), fd);
    if (ab) unpack16<T>(*(const uint4*)(ab + (long)px * lda + v * EPV), fa);
    if (ab2) unpack16<T>(*(const uint4*)(ab2 + (long)px * lda2 + v * EPV), fa2);
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const float a = sa[v * EPV + e], z = a * fx[e] + sb[v * EPV + e];
      const float dz = silu ? fd[e] * silu_grad_T<T>(z) : fd[e];
      float r = a * dz - (k0 + k1 * fx[e]);
      if (ab) r += fa[e];
      if (ab2) r += fa2[e];
      out[e] = r;
    }
    *(uint4*)(ob + (long)px * lddx + v * EPV) = pack16<T>(out);
  }
}

int gn_bwd_small(hipStream_t st, DType dt, const void* x, long ldx, const void* dy, long lddy, const float* coef,
                 const float* mr, int B, long HW, int C, int silu, const void* addend, long lda, void* dx, long lddx,
                 const void* addend2, long lda2) {
  KDIP_REQUIRE(gn_small_eligible(dt, HW, C), "groupnorm(small) backward: unsupported shape HW=%ld C=%d", HW, C);
  dim3 grid(32, B);
  prof_begin(st, PC_GN_BWD_APPLY, 0, (3.0 + (addend ? 1 : 0) + (addend2 ? 1 : 0)) * B * HW * C * (dt == DT_BF16 ? 2.0 : 4.0), "gn_small", B, HW, C, (addend ? 1 : 0) + (addend2 ? 1 : 0));
  if (dt == DT_BF16)
    hipLaunchKernelGGL(gn_bwd_small_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, coef,
                       mr, (int)HW, C, C / 32, silu, (const bf16_t*)addend, lda, (const bf16_t*)addend2, lda2, (bf16_t*)dx, lddx);
  else
    hipLaunchKernelGGL(gn_bwd_small_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, (const float*)dy, lddy, coef, mr,
                       (int)HW, C, C / 32, silu, (const float*)addend, lda, (const float*)addend2, lda2, (float*)dx, lddx);
  prof_end(st);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

}  // namespace kdip
