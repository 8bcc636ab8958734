// Strided, batched MFMA GEMM  C[b] = alpha * A[b] * B[b]  with arbitrary element strides.
//
// Used where both operands are activations (no pre-packing possible): the attention
// products of QKVAttentionLegacy (guided_diffusion/unet.py:339-356: QK^T, PV) and their
// backward (dP, dV, dQ, dK), and the dense DCT basis products (condition/utils.py:91-103).
// < 2 % of path FLOPs, so the design goal is generality (any transpose via strides) on the
// matrix cores, not peak: 64x64 tile, 4 waves (2x2), K-chunk 32 staged through padded LDS
// ([row][k] images, 16-byte conflict-free fragment reads), fp32 accumulate.
#include "common.h"
#include "kernels.h"

namespace kdip {

template <typename T> struct Mma2;
template <> struct Mma2<bf16_t> {
  static constexpr int KSTEP = 16;
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma2<float> {
  static constexpr int KSTEP = 8;
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], c, 0, 0, 0);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void bgemm_kernel(BGemm g) {
  constexpr int BM = 64, BN = 64, BK = 32;
  constexpr int ROWB = BK * (int)sizeof(T) + 16;
  constexpr int KSTEP = Mma2<T>::KSTEP;
  __shared__ __attribute__((aligned(16))) unsigned char As[BM * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bz = blockIdx.z, b1 = bz / g.nb2, b2 = bz % g.nb2;
  const T* A = (const T*)g.A + b1 * g.sab1 + b2 * g.sab2;
  const T* Bp = (const T*)g.Bm + b1 * g.sbb1 + b2 * g.sbb2;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  // staging index maps: walk the unit-stride dimension with consecutive threads
  const bool a_kfast = (g.sak == 1), b_kfast = (g.sbk == 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int k0 = 0; k0 < g.K; k0 += BK) {
#pragma unroll
    for (int i = 0; i < (BM * BK) / 256; ++i) {
      int e = tid + i * 256;
      int r, k;
      if (a_kfast) { r = e / BK; k = e % BK; } else { k = e / BM; r = e % BM; }
      T v = A[(long)(m0 + r) * g.sam + (long)(k0 + k) * g.sak];
      *(T*)(As + r * ROWB + k * (int)sizeof(T)) = v;
      if (b_kfast) { r = e / BK; k = e % BK; } else { k = e / BN; r = e % BN; }
      T w = Bp[(long)(k0 + k) * g.sbk + (long)(n0 + r) * g.sbn];
      *(T*)(Bs + r * ROWB + k * (int)sizeof(T)) = w;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < BK / KSTEP; ++ks) {
      uint4 a = *(const uint4*)(As + (wm * 32 + (lane & 31)) * ROWB + ks * 32 + (lane >> 5) * 16);
      uint4 b = *(const uint4*)(Bs + (wn * 32 + (lane & 31)) * ROWB + ks * 32 + (lane >> 5) * 16);
      Mma2<T>::run(a, b, acc);
    }
    __syncthreads();
  }
  const int n = n0 + wn * 32 + (lane & 31);
  char* Cb = (char*)g.C;
  const long cb = b1 * g.scb1 + b2 * g.scb2;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    long off = cb + (long)m * g.scm + (long)n * g.scn;
    float v = acc[r] * g.alpha;
    if (g.c_f32) ((float*)Cb)[off] = v;
    else ((T*)Cb)[off] = from_f32<T>(v);
  }
}

int bgemm(hipStream_t st, DType dt, const BGemm& g) {
  KDIP_REQUIRE(g.M % 64 == 0 && g.N % 64 == 0 && g.K % 32 == 0, "bgemm: M=%d N=%d K=%d must be multiples of 64/64/32",
               g.M, g.N, g.K);
  dim3 grid(g.N / 64, g.M / 64, g.nb1 * g.nb2);
  if (dt == DT_BF16) hipLaunchKernelGGL(bgemm_kernel<bf16_t>, grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL(bgemm_kernel<float>, grid, dim3(256), 0, st, g);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

}  // namespace kdip
