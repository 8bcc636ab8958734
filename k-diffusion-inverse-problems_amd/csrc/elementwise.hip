// HBM-bound glue kernels of the UNet path (NHWC): 2x2 average pool / nearest x2 upsample
// (Downsample/Upsample with use_conv=False inside up/down ResBlocks,
// guided_diffusion/unet.py:101-108,133-137,190-197), channel concat/split copies
// (th.cat at :662), residual adds, row softmax fwd/bwd (:352), SiLU, sinusoidal timestep
// embedding (guided_diffusion/nn.py:103-121) and the NCHW<->NHWC boundary converts.
// All are 16-byte-per-lane, channel-contiguous streaming kernels.
#include "common.h"
#include "kernels.h"

namespace kdip {

static inline int ew_grid(long n, int per = 256) {
  long g = (n + per - 1) / per;
  if (g < 1) g = 1;
  if (g > 16384) g = 16384;
  return (int)g;
}

// ------------------------------------------------------------------ pool / upsample ----
template <typename T>
__global__ void avgpool2_kernel(const T* __restrict__ x, long ldx, int B, int H, int W, int VP, T* __restrict__ y,
                                long ldy, float scale) {
  constexpr int EPV = TypeInfo<T>::EPV;
  const int Ho = H / 2, Wo = W / 2;
  long nvec = (long)B * Ho * Wo * VP;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
    int vi = (int)(v % VP);
    long op = v / VP;
    int ox = (int)(op % Wo);
    long t = op / Wo;
    int oy = (int)(t % Ho), b = (int)(t / Ho);
    const T* p = x + (((long)b * H + 2 * oy) * W + 2 * ox) * ldx + (long)vi * EPV;
    float a[EPV], c[EPV], d[EPV], e[EPV], o[EPV];
    unpack16<T>(*(const uint4*)p, a);
    unpack16<T>(*(const uint4*)(p + ldx), c);
    unpack16<T>(*(const uint4*)(p + (long)W * ldx), d);
    unpack16<T>(*(const uint4*)(p + (long)W * ldx + ldx), e);
#pragma unroll
    for (int i = 0; i < EPV; ++i) o[i] = (a[i] + c[i] + d[i] + e[i]) * scale;
    *(uint4*)(y + op * ldy + (long)vi * EPV) = pack16<T>(o);
  }
}

int avgpool2(hipStream_t st, DType dt, const void* x, long ldx, int B, int H, int W, int C, void* y, long ldy,
             float scale) {
  int VP = C / (dt == DT_BF16 ? 8 : 4);
  long nvec = (long)B * (H / 2) * (W / 2) * VP;
  if (dt == DT_BF16)
    hipLaunchKernelGGL(avgpool2_kernel<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, st, (const bf16_t*)x, ldx, B, H, W, VP,
                       (bf16_t*)y, ldy, scale);
  else
    hipLaunchKernelGGL(avgpool2_kernel<float>, dim3(ew_grid(nvec)), dim3(256), 0, st, (const float*)x, ldx, B, H, W, VP,
                       (float*)y, ldy, scale);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// ------------------------------------------------------------------------- softmax ----
// One wavefront per row (cols <= 4096); fp32 math as in `th.softmax(weight.float())`.
template <typename T>
__global__ void softmax_rows_kernel(const float* __restrict__ s, long rows, int cols, T* __restrict__ p) {
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = s + row * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, sr[c]);
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) sum += __expf(sr[c] - m);
  sum = wave_sum(sum);
  float inv = 1.f / sum;
  for (int c = lane; c < cols; c += 64) p[row * cols + c] = from_f32<T>(__expf(sr[c] - m) * inv);
}

int softmax_rows(hipStream_t st, DType dt, const float* s, long rows, int cols, void* p) {
  dim3 grid(cdiv(rows, 4));
  if (dt == DT_BF16) hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, grid, dim3(256), 0, st, s, rows, cols, (bf16_t*)p);
  else hipLaunchKernelGGL(softmax_rows_kernel<float>, grid, dim3(256), 0, st, s, rows, cols, (float*)p);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

template <typename T>
__global__ void softmax_bwd_rows_kernel(const T* __restrict__ p, const float* __restrict__ dp, long rows, int cols,
                                        T* __restrict__ ds) {
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const T* pr = p + row * cols;
  const float* dr = dp + row * cols;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += to_f32(pr[c]) * dr[c];
  dot = wave_sum(dot);
  for (int c = lane; c < cols; c += 64) ds[row * cols + c] = from_f32<T>(to_f32(pr[c]) * (dr[c] - dot));
}

int softmax_bwd_rows(hipStream_t st, DType dt, const void* p, const float* dp, long rows, int cols, void* ds) {
  dim3 grid(cdiv(rows, 4));
  if (dt == DT_BF16)
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)p, dp, rows, cols, (bf16_t*)ds);
  else
    hipLaunchKernelGGL(softmax_bwd_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)p, dp, rows, cols, (float*)ds);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// ----------------------------------------------------------------- small fp32 helpers ----
__global__ void silu_f32_kernel(const float* x, long n, float* y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float z = x[i];
    y[i] = z / (1.f + expf(-z));
  }
}
// (16-byte loads over the aligned body, scalar tail: the VJP's first launch waits for this pass over the cotangent)
__global__ void amax_bits_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
  unsigned m = 0;
  const long nv = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n / 4 : 0;
  const float4* xv = (const float4*)x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    const float4 v = xv[i];
    const unsigned b0 = __float_as_uint(v.x) & 0x7fffffffu, b1 = __float_as_uint(v.y) & 0x7fffffffu;
    const unsigned b2 = __float_as_uint(v.z) & 0x7fffffffu, b3 = __float_as_uint(v.w) & 0x7fffffffu;
    if (b0 < 0x7f800000u && b0 > m) m = b0;       // finite values only
    if (b1 < 0x7f800000u && b1 > m) m = b1;
    if (b2 < 0x7f800000u && b2 > m) m = b2;
    if (b3 < 0x7f800000u && b3 > m) m = b3;
  }
  for (long i = nv * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned b = __float_as_uint(x[i]) & 0x7fffffffu;
    if (b < 0x7f800000u && b > m) m = b;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// the same over a SAMPLE of the tensor (every `stride`-th 16-byte vector): *out (pre-zeroed by the caller) = bits of the sampled max |x|.
// A power-of-two scale only needs the magnitude to within a few binades, and reading 1 / stride of a gradient tensor costs a launch
// (~4 us), not a pass over it.
__global__ void amax_bits_sampled_kernel(const float4* __restrict__ x, long nvec, int stride, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * stride; i < nvec; i += (long)gridDim.x * blockDim.x * stride) {
    const float4 v = x[i];
    const unsigned b0 = __float_as_uint(v.x) & 0x7fffffffu, b1 = __float_as_uint(v.y) & 0x7fffffffu;
    const unsigned b2 = __float_as_uint(v.z) & 0x7fffffffu, b3 = __float_as_uint(v.w) & 0x7fffffffu;
    if (b0 < 0x7f800000u && b0 > m) m = b0;
    if (b1 < 0x7f800000u && b1 > m) m = b1;
    if (b2 < 0x7f800000u && b2 > m) m = b2;
    if (b3 < 0x7f800000u && b3 > m) m = b3;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
int amax_bits_sampled(hipStream_t st, const float* x, long n, unsigned* out_zeroed) {
  const long nvec = n / 4;
  int stride = (int)(nvec / (64L * 1024));              // ~64 K sampled vectors at most
  if (stride < 1) stride = 1;
  // an ODD stride: the NHWC pixel pitches here are powers of two (x 3 at most) vectors, and an even stride that divides the pitch would
  // sample the same few channels of every k-th pixel (stride 512 at ld 256: channels 0..3 only); an odd one walks through all channel vectors
  if (stride > 1) stride |= 1;
  if (stride > 1 && stride % 3 == 0) stride += 2;
  long g = (nvec / stride + 255) / 256; if (g > 256) g = 256; if (g < 1) g = 1;
  hipLaunchKernelGGL(amax_bits_sampled_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float4*)x, nvec, stride, out_zeroed);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}
int amax_bits(hipStream_t st, const float* x, long n, unsigned* out) {
  KDIP_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(unsigned), st));
  long g = (n + 256 * 16 - 1) / (256 * 16); if (g > 1024) g = 1024; if (g < 1) g = 1;
  hipLaunchKernelGGL(amax_bits_kernel, dim3((unsigned)g), dim3(256), 0, st, x, n, out);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

int silu_f32(hipStream_t st, const float* x, long n, float* y) {
  hipLaunchKernelGGL(silu_f32_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, n, y);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

__global__ void timestep_embedding_kernel(const float* t, int B, int dim, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int half = dim / 2;
  if (i >= B * half) return;
  int b = i / half, j = i % half;
  float freq = expf(-logf(10000.f) * (float)j / (float)half);
  float arg = t[b] * freq;
  out[(long)b * dim + j] = cosf(arg);
  out[(long)b * dim + half + j] = sinf(arg);
}
int timestep_embedding(hipStream_t st, const float* t, int B, int dim, float* out) {
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv((long)B * dim / 2, 256)), dim3(256), 0, st, t, B, dim, out);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

template <typename T>
__global__ void f32_to_T_kernel(const float* x, long n, T* y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = from_f32<T>(x[i]);
}
int f32_to_T(hipStream_t st, DType dt, const float* x, long n, void* y) {
  if (dt == DT_BF16) hipLaunchKernelGGL(f32_to_T_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, st, x, n, (bf16_t*)y);
  else hipLaunchKernelGGL(f32_to_T_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, st, x, n, (float*)y);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}
template <typename T>
__global__ void T_to_f32_kernel(const T* x, long n, float* y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = to_f32(x[i]);
}
int T_to_f32(hipStream_t st, DType dt, const void* x, long n, float* y) {
  if (dt == DT_BF16) hipLaunchKernelGGL(T_to_f32_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)x, n, y);
  else hipLaunchKernelGGL(T_to_f32_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, st, (const float*)x, n, y);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// NCHW fp32 -> NHWC T with channel padding (pixel-major threads; C is tiny: 3 or 6): the padded pixel
// row is assembled in registers and written with 16-byte stores.
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int B, int C, long HW, float scale, T* __restrict__ y,
                                    long ld, int Cpad) {
  constexpr int EPV = TypeInfo<T>::EPV;
  long n = (long)B * HW;
  if (Cpad % EPV == 0 && C <= EPV && (ld * (long)sizeof(T)) % 16 == 0) {
    // few real channels (the 3-channel image, 6-channel cotangent in bf16): one thread per 16-byte vector of the padded row, so that
    // consecutive lanes store consecutive 16 bytes (one thread per pixel wrote 4 vectors 64 bytes apart per instruction)
    const int nv = Cpad / EPV;
    const long nvec = n * nv;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < nvec; j += (long)gridDim.x * blockDim.x) {
      const long i = j / nv; const int v = (int)(j - i * nv);
      uint4 o = make_uint4(0, 0, 0, 0);
      if (v == 0) {
        const long b = i / HW, p = i % HW;
        float f[EPV];
#pragma unroll
        for (int c = 0; c < EPV; ++c) f[c] = c < C ? x[(b * C + c) * HW + p] * scale : 0.f;
        o = pack16<T>(f);
      }
      *(uint4*)(y + i * ld + (long)v * EPV) = o;
    }
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long b = i / HW, p = i % HW;
    T* o = y + i * ld;
    if (Cpad % EPV == 0 && C <= 2 * EPV && (ld * (long)sizeof(T)) % 16 == 0) {
      float f[2 * EPV];
#pragma unroll
      for (int c = 0; c < 2 * EPV; ++c) f[c] = c < C ? x[(b * C + c) * HW + p] * scale : 0.f;
      *(uint4*)o = pack16<T>(f);
      const int nv = Cpad / EPV;
      if (nv > 1) *(uint4*)(o + EPV) = pack16<T>(f + EPV);
      const uint4 z = make_uint4(0, 0, 0, 0);
      for (int v = 2; v < nv; ++v) *(uint4*)(o + (long)v * EPV) = z;
    } else {
      for (int c = 0; c < Cpad; ++c) o[c] = from_f32<T>(c < C ? x[(b * C + c) * HW + p] * scale : 0.f);
    }
  }
}
int nchw_to_nhwc(hipStream_t st, DType dt, const float* x, int B, int C, int H, int W, float scale, void* y, long ld,
                 int Cpad) {
  long HW = (long)H * W;
  if (dt == DT_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(ew_grid(B * HW)), dim3(256), 0, st, x, B, C, HW, scale, (bf16_t*)y, ld, Cpad);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(ew_grid(B * HW)), dim3(256), 0, st, x, B, C, HW, scale, (float*)y, ld, Cpad);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, long ld, int B, int C, long HW, float* __restrict__ y) {
  long n = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long b = i / HW, p = i % HW;
    if (sizeof(T) == 4 && C <= 8 && ld % 4 == 0) {       // fp32 rows padded to 32 channels, 3 / 6 used: two 16-byte loads instead of C scalar ones
      const float4 lo = *(const float4*)((const float*)x + i * ld);
      const float4 hi = C > 4 ? *(const float4*)((const float*)x + i * ld + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int c = 0; c < 8; ++c) if (c < C) y[(b * C + c) * HW + p] = v[c];
    } else {
      for (int c = 0; c < C; ++c) y[(b * C + c) * HW + p] = to_f32(x[i * ld + c]);
    }
  }
}
int nhwc_to_nchw_f32(hipStream_t st, const float* x, long ld, int B, int C, int H, int W, float* y) {
  long HW = (long)H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_grid(B * HW)), dim3(256), 0, st, x, ld, B, C, HW, y);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}
int nhwc_T_to_nchw_f32(hipStream_t st, DType dt, const void* x, long ld, int B, int C, int H, int W, float* y) {
  long HW = (long)H * W;
  if (dt == DT_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(ew_grid(B * HW)), dim3(256), 0, st, (const bf16_t*)x, ld, B, C, HW, y);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_grid(B * HW)), dim3(256), 0, st, (const float*)x, ld, B, C, HW, y);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}


// ---- 3x3 convs with a handful of input or output channels as 1x1 convs (fp32-storage modes; round 6) --------------------------------
// The image conv (3 -> C), the output head (C -> 6) and their input-gradients are 3x3 convs whose K or N the implicit-GEMM kernel pads to
// 32: 10x / 5x the MFMA work of the useful products, 2.5 % of the MFMA instructions of a guided call.  Both are 1x1 convs in disguise:
//   few INPUT channels (guided_diffusion/unet.py:484 forward, :614-618 backward): fold the taps into K.  im2col3_nchw writes
//       X1[b, y, x, t*C + c] = scale * x[b, c, y + t/3 - 1, x + t%3 - 1]   (zero outside the image, zero in the padding channels)
//     straight from the NCHW fp32 tensor the C ABI hands over (it replaces nchw_to_nhwc), and the conv is a 1x1 with K = 9 C (27 -> 32, 54 -> 64).
//   few OUTPUT channels (:614-618 forward, :484 backward): fold the taps into N.  A 1x1 conv with N = 9 Co (54 -> 64, 27 -> 32) leaves the per-tap
//     partial products P[b, y, x, t*Co + co]; tap_gather_nchw adds the nine shifted taps and the bias and writes the NCHW fp32 result
//       out[b, co, y, x] = bias[co] + sum_t P[b, y + t/3 - 1, x + t%3 - 1, t*Co + co]
//     (it replaces nhwc_to_nchw_f32).
__global__ void im2col3_nchw_kernel(const float* __restrict__ x, int B, int C, int H, int W, float scale, float* __restrict__ y, long ld) {
  const int nv = (int)(ld / 4);                       // 16-byte vectors per pixel row (ld = padded 9 C)
  const long HW = (long)H * W, nvec = (long)B * HW * nv;
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < nvec; j += (long)gridDim.x * blockDim.x) {
    const long i = j / nv; const int v = (int)(j - i * nv);
    const long b = i / HW; const int p = (int)(i - b * HW), py = p / W, px = p - py * W;
    float f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = v * 4 + e, t = k / C, c = k - t * C;
      const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
      f[e] = (t < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[(b * C + c) * HW + (long)yy * W + xx] * scale : 0.f;
    }
    *(float4*)(y + i * ld + (long)v * 4) = make_float4(f[0], f[1], f[2], f[3]);
  }
}
int im2col3_nchw(hipStream_t st, const float* x, int B, int C, int H, int W, float scale, float* y, long ld) {
  KDIP_REQUIRE(ld % 4 == 0 && ld >= 9 * C && ((uintptr_t)y % 16) == 0, "im2col3: bad row stride %ld for %d channels", ld, C);
  hipLaunchKernelGGL(im2col3_nchw_kernel, dim3(ew_grid((long)B * H * W * (ld / 4))), dim3(256), 0, st, x, B, C, H, W, scale, y, ld);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}
template <int CO>
__global__ void tap_gather_nchw_kernel(const float* __restrict__ P, long ld, int B, int H, int W, const float* __restrict__ bias, float* __restrict__ out) {
  const long HW = (long)H * W, n = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW; const int p = (int)(i - b * HW), py = p / W, px = p - py * W;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = bias ? bias[c] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {                      // fixed tap order: the result is run-to-run reproducible
      const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const float* q = P + (b * HW + (long)yy * W + xx) * ld + t * CO;
        if (CO % 2 == 0) {                             // t * CO floats is 8-byte aligned
#pragma unroll
          for (int c = 0; c < CO; c += 2) { const float2 v = *(const float2*)(q + c); acc[c] += v.x; acc[c + 1] += v.y; }
        } else {
#pragma unroll
          for (int c = 0; c < CO; ++c) acc[c] += q[c];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) out[(b * CO + c) * HW + p] = acc[c];
  }
}
int tap_gather_nchw(hipStream_t st, const float* P, long ld, int B, int Co, int H, int W, const float* bias, float* out) {
  KDIP_REQUIRE(ld >= 9 * Co && ld % 2 == 0 && ((uintptr_t)P % 8) == 0, "tap_gather: bad row stride %ld for %d channels", ld, Co);
  const dim3 g(ew_grid((long)B * H * W)), bl(256);
  if (Co == 3) hipLaunchKernelGGL(tap_gather_nchw_kernel<3>, g, bl, 0, st, P, ld, B, H, W, bias, out);
  else if (Co == 6) hipLaunchKernelGGL(tap_gather_nchw_kernel<6>, g, bl, 0, st, P, ld, B, H, W, bias, out);
  else return set_error(KDIP_ERR_UNSUPPORTED, "tap_gather: %d output channels (3 or 6)", Co);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// ------------------------------------------------------------------ LPIPS (VGG) helpers ----
// The perceptual metric of the caller harness (sample_condition_openai.py:46,161: lpips.LPIPS(net='vgg')) runs its 13 VGG convs
// through conv_forward; these kernels are the glue on fp32 NCHW planes: ReLU (+ 2x2 max pool) and the per-layer distance.
__global__ void relu_maxpool_planes_kernel(const float* __restrict__ x, long planes, int H, int W, int pool, float* __restrict__ y) {
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  const long n = planes * Ho * Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const long t = i / Wo;
    const int oy = (int)(t % Ho);
    const long pl = t / Ho;
    const float* p = x + pl * H * W;
    float v;
    if (pool) v = fmaxf(fmaxf(p[(2 * oy) * W + 2 * ox], p[(2 * oy) * W + 2 * ox + 1]), fmaxf(p[(2 * oy + 1) * W + 2 * ox], p[(2 * oy + 1) * W + 2 * ox + 1]));
    else v = p[oy * W + ox];
    y[i] = fmaxf(v, 0.f);
  }
}
int relu_maxpool_planes(hipStream_t st, const float* x, long planes, int H, int W, int pool, float* y) {
  KDIP_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), "maxpool2: odd plane size %dx%d", H, W);
  const long n = planes * (pool ? H / 2 : H) * (pool ? W / 2 : W);
  hipLaunchKernelGGL(relu_maxpool_planes_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, planes, H, W, pool, y);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// out[b] += mean over pixels of sum_c w[c] * (f0[c] / (|f0| + eps) - f1[c] / (|f1| + eps))^2, |f| = sqrt(sum_c f[c]^2) per pixel
// (lpips.normalize_tensor + lin layer + spatial_average); f0, f1: [B, C, HW] fp32.  One thread per pixel, coalesced over pixels.
__global__ void lpips_layer_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ w, int C, long HW,
                                   float* __restrict__ out) {
  const int b = blockIdx.y;
  const float* a = f0 + (long)b * C * HW;
  const float* c = f1 + (long)b * C * HW;
  float acc = 0.f;
  for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < HW; px += (long)gridDim.x * blockDim.x) {
    float n0 = 0.f, n1 = 0.f;
    for (int ch = 0; ch < C; ++ch) { const float u = a[ch * HW + px], v = c[ch * HW + px]; n0 += u * u; n1 += v * v; }
    const float r0 = 1.f / (sqrtf(n0) + 1e-10f), r1 = 1.f / (sqrtf(n1) + 1e-10f);
    float d = 0.f;
    for (int ch = 0; ch < C; ++ch) { const float t = a[ch * HW + px] * r0 - c[ch * HW + px] * r1; d += w[ch] * t * t; }
    acc += d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(out + b, acc / (float)HW);
}
int lpips_layer(hipStream_t st, const float* f0, const float* f1, const float* w, int B, int C, long HW, float* out) {
  long g = (HW + 255) / 256; if (g > 1024) g = 1024;
  hipLaunchKernelGGL(lpips_layer_kernel, dim3((unsigned)g, B), dim3(256), 0, st, f0, f1, w, C, HW, out);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}


// out[b] (+)= mean_i ((pred - target)^2 * exp(-logvar) + logvar) over the `per` elements of sample b: the two halves (pixel space,
// transform space) of the DWT-Var / DCT-Var objective, OpenAIDenoiserV2.loss (k_diffusion/external.py:145-159)
__global__ void gauss_nll_mean_kernel(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ logvar, long per,
                                      int accumulate, float* __restrict__ out) {
  const int b = blockIdx.y;
  const float *p = pred + (long)b * per, *t = target + (long)b * per, *lv = logvar + (long)b * per;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
    const float e = p[i] - t[i], l = lv[i];
    acc += e * e * __expf(-l) + l;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(out + b, acc / (float)per);
  (void)accumulate;
}
int gauss_nll_mean(hipStream_t st, const float* pred, const float* target, const float* logvar, int B, long per, int accumulate, float* out) {
  if (!accumulate) KDIP_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float) * B, st));
  long g = (per + 255) / 256; if (g > 256) g = 256;
  hipLaunchKernelGGL(gauss_nll_mean_kernel, dim3((unsigned)g, B), dim3(256), 0, st, pred, target, logvar, per, accumulate, out);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

}  // namespace kdip
