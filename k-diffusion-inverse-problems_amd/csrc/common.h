// Shared device/host helpers for libkdip_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "../../include/kdip.h"

namespace kdip {

typedef unsigned short bf16_t;   // raw bfloat16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// DT_F32X3: fp32 STORAGE with split-bf16 conv arithmetic (operands split into bf16 hi + lo, hi*hi + hi*lo + lo*hi on the bf16 MFMA,
// fp32 accumulate: ~2^-17 relative operand error instead of bf16's 2^-9).  Only conv_forward / pack_conv_weight /
// packed_weight_bytes take it; every other kernel of that mode runs its DT_F32 instantiation.
// DT_F32H3: the same with an fp16 HEAD as well (three fp16 MFMAs per product, 11 + 11-bit operands: two A planes instead of three, no weight-head
// re-encode) -- faster and more accurate inside the fp16 window, but an operand beyond +-65504 after scaling saturates the product instead of
// degrading it; handles of this mode carry the DT_F32X3 weights as well and callers redo a flagged call with them (unet.h: x3_alt).
enum DType { DT_F32 = 0, DT_BF16 = 1, DT_F32X3 = 2, DT_F32H3 = 3 };
inline bool is_x3(DType dt) { return dt == DT_F32X3 || dt == DT_F32H3; }
inline DType storage_dtype(DType dt) { return dt == DT_BF16 ? DT_BF16 : DT_F32; }
struct f32x3_t { float v; };       // kernel tag type of DT_F32X3: an fp32 element
struct f32h3_t { float v; };       // kernel tag type of DT_F32H3

// status codes: the global KDIP_* enum of include/kdip.h (included above)

extern thread_local std::string g_last_error;
int set_error(int code, const char* fmt, ...);

#define KDIP_HIP_CHECK(expr)                                                        \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess)                                                           \
      return kdip::set_error(KDIP_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                             hipGetErrorString(_e));                                \
  } while (0)

#define KDIP_LAUNCH_CHECK() KDIP_HIP_CHECK(hipGetLastError())

#define KDIP_REQUIRE(cond, ...)                                             \
  do {                                                                      \
    if (!(cond)) return kdip::set_error(KDIP_ERR_ARG, __VA_ARGS__);   \
  } while (0)

// ---- bf16 conversion (round-to-nearest-even), host+device ----
__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)v) << 16;
  return c.f;
}
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float> {
  static constexpr int EPV = 4;   // elements per 16-byte vector
  static constexpr DType dt = DT_F32;
};
template <> struct TypeInfo<bf16_t> {
  static constexpr int EPV = 8;
  static constexpr DType dt = DT_BF16;
};

__device__ inline float to_f32(float v) { return v; }
__device__ inline float to_f32(bf16_t v) { return bf16_to_f32(v); }
__device__ inline float to_f32(f32x3_t v) { return v.v; }
__device__ inline float to_f32(f32h3_t v) { return v.v; }
template <typename T> __device__ inline T from_f32(float v);
template <> __device__ inline float from_f32<float>(float v) { return v; }
template <> __device__ inline bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }
template <> __device__ inline f32x3_t from_f32<f32x3_t>(float v) { return f32x3_t{v}; }
template <> __device__ inline f32h3_t from_f32<f32h3_t>(float v) { return f32h3_t{v}; }

// 16-byte vector <-> float[EPV]
template <typename T> __device__ inline void unpack16(const uint4& v, float* out);
template <> __device__ inline void unpack16<float>(const uint4& v, float* out) {
  out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y);
  out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
}
template <> __device__ inline void unpack16<bf16_t>(const uint4& v, float* out) {
  out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xffff0000u);
  out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xffff0000u);
  out[4] = __uint_as_float(v.z << 16); out[5] = __uint_as_float(v.z & 0xffff0000u);
  out[6] = __uint_as_float(v.w << 16); out[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T> __device__ inline uint4 pack16(const float* in);
template <> __device__ inline uint4 pack16<float>(const float* in) {
  return make_uint4(__float_as_uint(in[0]), __float_as_uint(in[1]), __float_as_uint(in[2]), __float_as_uint(in[3]));
}
// two fp32 -> packed bf16x2 (round-to-nearest-even): one v_cvt_pk_bf16_f32 on gfx950
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
  typedef float f32x2_hw __attribute__((ext_vector_type(2)));
  f32x2_hw v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
}
template <> __device__ inline uint4 pack16<bf16_t>(const float* in) {
  return make_uint4(pack_bf16x2(in[0], in[1]), pack_bf16x2(in[2], in[3]), pack_bf16x2(in[4], in[5]), pack_bf16x2(in[6], in[7]));
}

__device__ inline float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ inline float silu_grad_f(float z) {
  float s = 1.f / (1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}

// v_rcp_f32 form (1 ulp) for the bf16 pipeline; the f32 parity mode keeps the IEEE division above
__device__ inline float silu_grad_fast(float z) {
  float s = __builtin_amdgcn_rcpf(1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}

// storage-type dispatch: bf16 tensors take the v_rcp_f32 forms (the IEEE division is ~10 extra VALU ops per element,
// enough to make the GroupNorm streaming kernels VALU- instead of HBM-bound); f32 (parity mode) keeps IEEE division
__device__ inline float silu_fast(float z) { return z * __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
template <typename T> __device__ inline float silu_T(float z) { return sizeof(T) == 2 ? silu_fast(z) : silu_f(z); }
template <typename T> __device__ inline float silu_grad_T(float z) { return sizeof(T) == 2 ? silu_grad_fast(z) : silu_grad_f(z); }

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// compute units a stream may use: the device's count, or what kdip_stream_create_cu_mask registered for a CU-masked stream
// (persistent kernels size their grids by it)
void stream_register_cus(hipStream_t st, int ncus);      // ncus <= 0: forget
int stream_cus(hipStream_t st, int device_cus);

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- opt-in per-launch profiler (HIP events on the launch stream; used by bench.py's roofline leg) ----
enum ProfClass { PC_CONV3_128x128 = 0, PC_CONV3_128x64, PC_CONV3_128x32, PC_CONV1_128x128, PC_CONV1_128x64,
                 PC_CONV1_128x32, PC_GN_STATS, PC_GN_APPLY, PC_GN_BWD_STATS, PC_GN_BWD_APPLY,
                 // operator / transform / sampler kernels on fp32 [B,3,S,S] planes (HBM-bound: algorithmic bytes = every tensor read or written once)
                 PC_OP_BLUR, PC_OP_FFT, PC_OP_OTF, PC_OP_DWT, PC_OP_GATHER, PC_OP_RESIZE, PC_OP_POINTWISE, PC_COUNT };
constexpr int PC_NUM_CONV = PC_GN_STATS;      // classes [0, PC_NUM_CONV) are MFMA convs (flops); the rest are HBM streaming passes (bytes)
extern bool g_prof_on;
void prof_begin(hipStream_t st, int cls, double flops, double bytes, const char* tag = nullptr, long d0 = 0, long d1 = 0, long d2 = 0, long d3 = 0);
void prof_end(hipStream_t st);
// brackets every launch of the enclosing scope (HIP events on `st`; nothing happens unless the profiler is on)
struct ProfScope {
  hipStream_t st;
  ProfScope(hipStream_t s, int cls, double bytes, const char* tag, long d0 = 0, long d1 = 0, long d2 = 0, long d3 = 0) : st(s) { prof_begin(s, cls, 0, bytes, tag, d0, d1, d2, d3); }
  ~ProfScope() { prof_end(st); }
};

}  // namespace kdip
