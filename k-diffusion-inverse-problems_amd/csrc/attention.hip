// Fused multi-head self-attention of the UNet's AttentionBlock, bf16 storage, head width 64 (QKVAttentionLegacy:
// guided_diffusion/unet.py:297-307 AttentionBlock._forward, :339-356 QKVAttentionLegacy.forward -- weight =
// softmax((q*s)^T (k*s)) in fp32 with s = ch^-1/4, a = weight v) and its VJP.  No T x T matrix ever reaches HBM: scores live in
// MFMA accumulators, the softmax is the online (running max / running sum) form, the backward pass recomputes the probabilities
// from the saved per-row log-sum-exp.
//
// Orientation.  Every score tile is computed TRANSPOSED with respect to the tensor that is being accumulated, so that a lane
// owns ONE column of the tile (v_mfma_f32_32x32x16_bf16: lane (n, h) holds rows 8g + 4h + j of column n in register 4g + j):
//   forward / dQ:  S^T[key][query] = K Q^T   -> per-query quantities (running max, sum, D, log-sum-exp) are per-lane scalars and
//                  the accumulated O^T[d][query] / dQ^T[d][query] tiles share the lane's query column: rescaling needs no
//                  cross-lane traffic, row reductions need one exchange with lane n + 32;
//   dK / dV:       S[query][key] = Q K^T     -> dK^T[d][key], dV^T[d][key] share the lane's key column.
// A finished score tile becomes the B operand of the second MFMA (K index = the tile's row index) with one v_cvt_pk_bf16 pair and
// two v_permlane32_swap per 16 rows (rows_to_b below) -- no LDS round trip.  The A operands of the second MFMAs need the K index
// contiguous per lane, i.e. V^T, K^T, Q^T, dO^T: small per-head transposes ([64][T] per (image, head)) written by
// head_transpose_kernel just before (B*T*C elements each; the score matrices they replace are T/64 times larger).
//
// One wave owns 32 tokens (queries, or keys in the dK/dV kernel).  Two families of kernels: the direct ones stream the other side
// in tiles straight from global memory, wave by wave, no barriers (kept as the KDIP_ATTN_LDS=0 reference); the shipped ones stage
// the tiles through LDS for the 4 waves of a block (further down: 2 x faster at T = 1024).
#include "kernels.h"

namespace kdip {
namespace {

__device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float other_half(float v) { return __shfl_xor(v, 32, 64); }   // lane n + 32: same column, the other rows
__device__ __forceinline__ float ex2(float v) { return __builtin_amdgcn_exp2f(v); }

// rows 16s .. 16s+15 of an accumulator tile -> bf16 B operand of k-step s (K index = tile row, natural order): lane (n, h) ends
// up with rows 16s + 8h .. + 7 of column n.  Also the 16-byte store vector of those rows for column n.
__device__ __forceinline__ uint4 rows_to_b(const f32x16& t, int s) {
  uint32_t a0 = pack_bf16x2(t[8 * s + 0], t[8 * s + 1]), a1 = pack_bf16x2(t[8 * s + 2], t[8 * s + 3]);
  uint32_t b0 = pack_bf16x2(t[8 * s + 4], t[8 * s + 5]), b1 = pack_bf16x2(t[8 * s + 6], t[8 * s + 7]);
  auto r = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
  a0 = r[0]; b0 = r[1];
  r = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
  a1 = r[0]; b1 = r[1];
  return make_uint4(a0, a1, b0, b1);
}

__device__ __forceinline__ float dot8(const uint4& a, const uint4& b) {
  float s = 0.f;
  const uint32_t* x = (const uint32_t*)&a; const uint32_t* y = (const uint32_t*)&b;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += __uint_as_float(x[i] << 16) * __uint_as_float(y[i] << 16) + __uint_as_float(x[i] & 0xffff0000u) * __uint_as_float(y[i] & 0xffff0000u);
  return s;
}

struct AttnP {
  const bf16_t* qkv; long ld;          // [B][T][ld]: head h owns channels 192 h + (q: 0..63, k: 64..127, v: 128..191)
  const bf16_t* vt;                    // V^T [B][heads][64][T]
  bf16_t* o; long ldo;                 // attention output [B][T][ldo], head h at channels 64 h ..
  float* lse;                          // [B][heads][T]: log2-domain log-sum-exp of the scaled scores
  int T, heads; float c;               // c = log2(e) / sqrt(64)
};

__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  if (q0 >= p.T) return;
  const int head = blockIdx.y, b = blockIdx.z;
  const bf16_t* base = p.qkv + (long)b * p.T * p.ld + head * 192 + 8 * h;
  uint4 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *(const uint4*)(base + (long)(q0 + n) * p.ld + 16 * s);
  const bf16_t* kb = base + 64;
  const bf16_t* vtb = p.vt + ((long)(b * p.heads + head) * 64) * p.T + 8 * h;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  // software pipeline over key tiles: the K fragments of tile i+1 are requested right after the score MFMAs of tile i have
  // consumed the current ones, the V^T fragments of tile i at its top -- both round trips hide under the softmax arithmetic
  uint4 kf[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s][0] = *(const uint4*)(kb + (long)n * p.ld + 16 * s);
    kf[s][1] = *(const uint4*)(kb + (long)(32 + n) * p.ld + 16 * s);
  }
  for (int k0 = 0; k0 < p.T; k0 += 64) {
    uint4 vf[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      vf[ks][0] = *(const uint4*)(vtb + (long)n * p.T + k0 + 16 * ks);
      vf[ks][1] = *(const uint4*)(vtb + (long)(32 + n) * p.T + k0 + 16 * ks);
    }
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      s0 = mma(kf[s][0], qf[s], s0);
      s1 = mma(kf[s][1], qf[s], s1);
    }
    {
      const int kn = k0 + 64 < p.T ? k0 + 64 : k0;          // (last tile: a redundant reload keeps the loop branch-free)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kf[s][0] = *(const uint4*)(kb + (long)(kn + n) * p.ld + 16 * s);
        kf[s][1] = *(const uint4*)(kb + (long)(kn + 32 + n) * p.ld + 16 * s);
      }
    }
    float mx = s0[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) { mx = fmaxf(mx, s0[r]); mx = fmaxf(mx, s1[r]); }
    mx = fmaxf(mx, other_half(mx));
    const float m_new = fmaxf(m_run, mx * p.c);
    const float corr = ex2(m_run - m_new);
    m_run = m_new;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = ex2(s0[r] * p.c - m_new); s1[r] = ex2(s1[r] * p.c - m_new);
      ls += s0[r] + s1[r];
      o0[r] *= corr; o1[r] *= corr;
    }
    l_run = l_run * corr + ls;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 pf = rows_to_b(ks < 2 ? s0 : s1, ks & 1);          // keys k0 + 16 ks + 8h .. + 7 of this lane's query
      o0 = mma(vf[ks][0], pf, o0);
      o1 = mma(vf[ks][1], pf, o1);
    }
  }
  const float l_tot = l_run + other_half(l_run);
  const float inv = 1.f / l_tot;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] *= inv; o1[r] *= inv; }
  bf16_t* orow = p.o + ((long)b * p.T + q0 + n) * p.ldo + head * 64 + 8 * h;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    *(uint4*)(orow + 16 * s) = rows_to_b(o0, s);
    *(uint4*)(orow + 32 + 16 * s) = rows_to_b(o1, s);
  }
  if (h == 0) p.lse[((long)b * p.heads + head) * p.T + q0 + n] = m_run + __log2f(l_tot);
}

struct AttnBP {
  const bf16_t* qkv; long ld;
  const bf16_t* dO; long lddo;         // cotangent of the attention output [B][T][lddo]
  const bf16_t* o; long ldo;           // saved attention output
  const float* lse; float* D;          // [B][heads][T]; D = rowsum(dO * O), written by the dQ kernel
  const bf16_t *kt, *qt, *dot;         // K^T, Q^T, dO^T [B][heads][64][T]
  bf16_t* dqkv; long ldg;              // [B][T][ldg], layout of qkv
  int T, heads; float c, alpha;        // alpha = 1 / sqrt(64)
};

// dQ^T[d][q] = alpha * sum_k K^T[d][k] dS^T[k][q],  dS^T = P^T o (dP^T - D_q),  dP^T[k][q] = sum_d V[k][d] dO[q][d]
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  if (q0 >= p.T) return;
  const int head = blockIdx.y, b = blockIdx.z;
  const bf16_t* base = p.qkv + (long)b * p.T * p.ld + head * 192 + 8 * h;
  const bf16_t* dorow = p.dO + ((long)b * p.T + q0 + n) * p.lddo + head * 64 + 8 * h;
  const bf16_t* orow = p.o + ((long)b * p.T + q0 + n) * p.ldo + head * 64 + 8 * h;
  uint4 qf[4], dof[4];
  float dpart = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = *(const uint4*)(base + (long)(q0 + n) * p.ld + 16 * s);
    dof[s] = *(const uint4*)(dorow + 16 * s);
    dpart += dot8(dof[s], *(const uint4*)(orow + 16 * s));
  }
  const float Dq = dpart + other_half(dpart);
  const long row = ((long)b * p.heads + head) * p.T + q0 + n;
  if (h == 0) p.D[row] = Dq;
  const float Lq = p.lse[row];
  const bf16_t* kb = base + 64;
  const bf16_t* vb = base + 128;
  const bf16_t* ktb = p.kt + ((long)(b * p.heads + head) * 64) * p.T + 8 * h;
  f32x16 g0, g1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
  uint4 ka[4], va[4];                                   // K / V fragments of the current key tile, requested one tile ahead
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    ka[s] = *(const uint4*)(kb + (long)n * p.ld + 16 * s);
    va[s] = *(const uint4*)(vb + (long)n * p.ld + 16 * s);
  }
  for (int k0 = 0; k0 < p.T; k0 += 32) {
    uint4 ta[2], tc[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      ta[ks] = *(const uint4*)(ktb + (long)n * p.T + k0 + 16 * ks);
      tc[ks] = *(const uint4*)(ktb + (long)(32 + n) * p.T + k0 + 16 * ks);
    }
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      st = mma(ka[s], qf[s], st);
      dp = mma(va[s], dof[s], dp);
    }
    {
      const int kn = k0 + 32 < p.T ? k0 + 32 : k0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        ka[s] = *(const uint4*)(kb + (long)(kn + n) * p.ld + 16 * s);
        va[s] = *(const uint4*)(vb + (long)(kn + n) * p.ld + 16 * s);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = ex2(st[r] * p.c - Lq) * (dp[r] - Dq);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 dsf = rows_to_b(st, ks);
      g0 = mma(ta[ks], dsf, g0);
      g1 = mma(tc[ks], dsf, g1);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] *= p.alpha; g1[r] *= p.alpha; }
  bf16_t* grow = p.dqkv + ((long)b * p.T + q0 + n) * p.ldg + head * 192 + 8 * h;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    *(uint4*)(grow + 16 * s) = rows_to_b(g0, s);
    *(uint4*)(grow + 32 + 16 * s) = rows_to_b(g1, s);
  }
}

// dV^T[d][k] = sum_q dO^T[d][q] P[q][k],  dK^T[d][k] = alpha * sum_q Q^T[d][q] dS[q][k]   (a lane owns one key column)
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
  const int key0 = (blockIdx.x * 4 + wave) * 32;
  if (key0 >= p.T) return;
  const int head = blockIdx.y, b = blockIdx.z;
  const bf16_t* base = p.qkv + (long)b * p.T * p.ld + head * 192 + 8 * h;
  uint4 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = *(const uint4*)(base + 64 + (long)(key0 + n) * p.ld + 16 * s);
    vf[s] = *(const uint4*)(base + 128 + (long)(key0 + n) * p.ld + 16 * s);
  }
  const bf16_t* dob = p.dO + (long)b * p.T * p.lddo + head * 64 + 8 * h;
  const long hrow = ((long)b * p.heads + head) * p.T;
  const bf16_t* qtb = p.qt + hrow * 64 + 8 * h;
  const bf16_t* dtb = p.dot + hrow * 64 + 8 * h;
  f32x16 k0a, k1a, v0a, v1a;
#pragma unroll
  for (int r = 0; r < 16; ++r) { k0a[r] = 0.f; k1a[r] = 0.f; v0a[r] = 0.f; v1a[r] = 0.f; }
  uint4 qa[4], da[4];                                   // Q / dO fragments of the current query tile, requested one tile ahead
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qa[s] = *(const uint4*)(base + (long)n * p.ld + 16 * s);
    da[s] = *(const uint4*)(dob + (long)n * p.lddo + 16 * s);
  }
  for (int q0 = 0; q0 < p.T; q0 += 32) {
    uint4 dta[2], dtc[2], qta[2], qtc[2];
    float4 Lr[4], Dr[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      dta[ks] = *(const uint4*)(dtb + (long)n * p.T + q0 + 16 * ks);
      dtc[ks] = *(const uint4*)(dtb + (long)(32 + n) * p.T + q0 + 16 * ks);
      qta[ks] = *(const uint4*)(qtb + (long)n * p.T + q0 + 16 * ks);
      qtc[ks] = *(const uint4*)(qtb + (long)(32 + n) * p.T + q0 + 16 * ks);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {          // rows (queries) q0 + 8g + 4h + j of this lane
      Lr[g] = *(const float4*)(p.lse + hrow + q0 + 8 * g + 4 * h);
      Dr[g] = *(const float4*)(p.D + hrow + q0 + 8 * g + 4 * h);
    }
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      st = mma(qa[s], kf[s], st);
      dp = mma(da[s], vf[s], dp);
    }
    {
      const int qn = q0 + 32 < p.T ? q0 + 32 : q0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qa[s] = *(const uint4*)(base + (long)(qn + n) * p.ld + 16 * s);
        da[s] = *(const uint4*)(dob + (long)(qn + n) * p.lddo + 16 * s);
      }
    }
    f32x16 ds;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float lr[4] = {Lr[g].x, Lr[g].y, Lr[g].z, Lr[g].w}, dr[4] = {Dr[g].x, Dr[g].y, Dr[g].z, Dr[g].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pr = ex2(st[4 * g + j] * p.c - lr[j]);
        st[4 * g + j] = pr;
        ds[4 * g + j] = pr * (dp[4 * g + j] - dr[j]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 pB = rows_to_b(st, ks), dB = rows_to_b(ds, ks);
      v0a = mma(dta[ks], pB, v0a); v1a = mma(dtc[ks], pB, v1a);
      k0a = mma(qta[ks], dB, k0a); k1a = mma(qtc[ks], dB, k1a);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { k0a[r] *= p.alpha; k1a[r] *= p.alpha; }
  bf16_t* grow = p.dqkv + ((long)b * p.T + key0 + n) * p.ldg + head * 192 + 8 * h;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    *(uint4*)(grow + 64 + 16 * s) = rows_to_b(k0a, s);
    *(uint4*)(grow + 96 + 16 * s) = rows_to_b(k1a, s);
    *(uint4*)(grow + 128 + 16 * s) = rows_to_b(v0a, s);
    *(uint4*)(grow + 160 + 16 * s) = rows_to_b(v1a, s);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-staged variants (KDIP_ATTN_LDS, default): the tiles of the streamed side are copied global -> LDS by the whole block with
// row-contiguous 16-byte loads (8 lanes per 128-byte row; the direct variants above fetch every fragment with one row per lane
// pair, 32 segments per wave instruction, and every one of the 4 waves fetches the same data) and all 4 waves read their MFMA
// fragments from there.  Rows are padded by 16 bytes (144- / 80-byte pitch: 16 consecutive rows fall on 16 distinct 4-bank
// groups, conflict-free ds_read_b128).  Double-buffered: tile i+1 is in flight in registers while tile i is consumed, one
// barrier per tile.  Waves whose 32 tokens lie beyond T (T = 64: two of four) only help with the copies.
#ifndef KDIP_ATTN_LDS
#define KDIP_ATTN_LDS 1
#endif
constexpr int AT_P64 = 144, AT_P32 = 80;              // LDS row pitch (bytes) of 64- / 32-element bf16 rows

// [64][64] / [32][64] bf16 tile, global row stride ld (elements): registers <- global, LDS <- registers (plain structs, not
// arrays: hipcc otherwise "promotes" the staging arrays of the three-tile kernel to 24 KB of LDS)
struct Stage2 { uint4 a, b; };
__device__ __forceinline__ Stage2 tile64_load(const bf16_t* src, long ld, int tid) {
  Stage2 r;
  r.a = *(const uint4*)(src + (long)(tid >> 3) * ld + (tid & 7) * 8);
  r.b = *(const uint4*)(src + (long)(32 + (tid >> 3)) * ld + (tid & 7) * 8);
  return r;
}
__device__ __forceinline__ void tile64_store(unsigned char* lds, const Stage2& r, int tid) {
  *(uint4*)(lds + (tid >> 3) * AT_P64 + (tid & 7) * 16) = r.a;
  *(uint4*)(lds + (32 + (tid >> 3)) * AT_P64 + (tid & 7) * 16) = r.b;
}
__device__ __forceinline__ uint4 tile32x64_load(const bf16_t* src, long ld, int tid) { return *(const uint4*)(src + (long)(tid >> 3) * ld + (tid & 7) * 8); }
__device__ __forceinline__ void tile32x64_store(unsigned char* lds, uint4 r, int tid) { *(uint4*)(lds + (tid >> 3) * AT_P64 + (tid & 7) * 16) = r; }
// [64][32] bf16 tile (64 rows of 64 bytes): one 16-byte vector per thread
__device__ __forceinline__ uint4 tile32_load(const bf16_t* src, long ld, int tid) { return *(const uint4*)(src + (long)(tid >> 2) * ld + (tid & 3) * 8); }
__device__ __forceinline__ void tile32_store(unsigned char* lds, uint4 r, int tid) { *(uint4*)(lds + (tid >> 2) * AT_P32 + (tid & 3) * 16) = r; }
// MFMA operand fragment of lane (n, h): row n of a 32-row block, elements 16 s + 8 h .. + 7
__device__ __forceinline__ uint4 frag64(const unsigned char* lds, int row, int s, int h) { return *(const uint4*)(lds + row * AT_P64 + (16 * s + 8 * h) * 2); }
__device__ __forceinline__ uint4 frag32(const unsigned char* lds, int row, int s, int h) { return *(const uint4*)(lds + row * AT_P32 + (16 * s + 8 * h) * 2); }

__global__ __launch_bounds__(256) void attn_fwd_lds_kernel(AttnP p) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[2][2][64 * AT_P64];          // [buffer][K | V^T][rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const bool active = q0 < p.T;
  const int qrow = active ? q0 + n : n;
  const int head = blockIdx.y, b = blockIdx.z;
  const bf16_t* base = p.qkv + (long)b * p.T * p.ld + head * 192;
  uint4 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *(const uint4*)(base + (long)qrow * p.ld + 8 * h + 16 * s);
  const bf16_t* kb = base + 64;
  const bf16_t* vtb = p.vt + ((long)(b * p.heads + head) * 64) * p.T;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  Stage2 kr = tile64_load(kb, p.ld, tid), vr = tile64_load(vtb, p.T, tid);
  tile64_store(sm[0][0], kr, tid);
  tile64_store(sm[0][1], vr, tid);
  __syncthreads();
  const int ntile = p.T >> 6;
  for (int it = 0; it < ntile; ++it) {
    const unsigned char* kl = sm[it & 1][0];
    const unsigned char* vl = sm[it & 1][1];
    const bool more = it + 1 < ntile;
    if (more) {
      kr = tile64_load(kb + (long)(it + 1) * 64 * p.ld, p.ld, tid);
      vr = tile64_load(vtb + (it + 1) * 64, p.T, tid);
    }
    if (active) {
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        s0 = mma(frag64(kl, n, s, h), qf[s], s0);
        s1 = mma(frag64(kl, 32 + n, s, h), qf[s], s1);
      }
      float mx = s0[0];
#pragma unroll
      for (int r = 0; r < 16; ++r) { mx = fmaxf(mx, s0[r]); mx = fmaxf(mx, s1[r]); }
      mx = fmaxf(mx, other_half(mx));
      const float m_new = fmaxf(m_run, mx * p.c);
      const float corr = ex2(m_run - m_new);
      m_run = m_new;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = ex2(s0[r] * p.c - m_new); s1[r] = ex2(s1[r] * p.c - m_new);
        ls += s0[r] + s1[r];
        o0[r] *= corr; o1[r] *= corr;
      }
      l_run = l_run * corr + ls;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 pf = rows_to_b(ks < 2 ? s0 : s1, ks & 1);
        o0 = mma(frag64(vl, n, ks, h), pf, o0);
        o1 = mma(frag64(vl, 32 + n, ks, h), pf, o1);
      }
    }
    if (more) {
      tile64_store(sm[(it + 1) & 1][0], kr, tid);
      tile64_store(sm[(it + 1) & 1][1], vr, tid);
    }
    __syncthreads();
  }
  if (!active) return;
  const float l_tot = l_run + other_half(l_run);
  const float inv = 1.f / l_tot;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] *= inv; o1[r] *= inv; }
  bf16_t* orow = p.o + ((long)b * p.T + q0 + n) * p.ldo + head * 64 + 8 * h;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    *(uint4*)(orow + 16 * s) = rows_to_b(o0, s);
    *(uint4*)(orow + 32 + 16 * s) = rows_to_b(o1, s);
  }
  if (h == 0) p.lse[((long)b * p.heads + head) * p.T + q0 + n] = m_run + __log2f(l_tot);
}

__global__ __launch_bounds__(256) void attn_bwd_dq_lds_kernel(AttnBP p) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[2][3][64 * AT_P64];          // [buffer][K | V | K^T][rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const bool active = q0 < p.T;
  const int qrow = active ? q0 + n : n;
  const int head = blockIdx.y, b = blockIdx.z;
  const bf16_t* base = p.qkv + (long)b * p.T * p.ld + head * 192;
  const bf16_t* dorow = p.dO + ((long)b * p.T + qrow) * p.lddo + head * 64 + 8 * h;
  const bf16_t* orow = p.o + ((long)b * p.T + qrow) * p.ldo + head * 64 + 8 * h;
  uint4 qf[4], dof[4];
  float dpart = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = *(const uint4*)(base + (long)qrow * p.ld + 8 * h + 16 * s);
    dof[s] = *(const uint4*)(dorow + 16 * s);
    dpart += dot8(dof[s], *(const uint4*)(orow + 16 * s));
  }
  const float Dq = dpart + other_half(dpart);
  const long row = ((long)b * p.heads + head) * p.T + qrow;
  if (active && h == 0) p.D[row] = Dq;
  const float Lq = p.lse[row];
  const bf16_t* kb = base + 64;
  const bf16_t* vb = base + 128;
  const bf16_t* ktb = p.kt + ((long)(b * p.heads + head) * 64) * p.T;
  f32x16 g0, g1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
  Stage2 kr = tile64_load(kb, p.ld, tid), vr = tile64_load(vb, p.ld, tid), tr = tile64_load(ktb, p.T, tid);
  tile64_store(sm[0][0], kr, tid); tile64_store(sm[0][1], vr, tid); tile64_store(sm[0][2], tr, tid);
  __syncthreads();
  const int ntile = p.T >> 6;
  for (int it = 0; it < ntile; ++it) {
    const unsigned char* kl = sm[it & 1][0];
    const unsigned char* vl = sm[it & 1][1];
    const unsigned char* tl = sm[it & 1][2];
    const bool more = it + 1 < ntile;
    if (more) {
      kr = tile64_load(kb + (long)(it + 1) * 64 * p.ld, p.ld, tid);
      vr = tile64_load(vb + (long)(it + 1) * 64 * p.ld, p.ld, tid);
      tr = tile64_load(ktb + (it + 1) * 64, p.T, tid);
    }
    if (active) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {                 // two 32-key halves of the staged tile
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          st = mma(frag64(kl, 32 * sub + n, s, h), qf[s], st);
          dp = mma(frag64(vl, 32 * sub + n, s, h), dof[s], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = ex2(st[r] * p.c - Lq) * (dp[r] - Dq);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const uint4 dsf = rows_to_b(st, ks);
          g0 = mma(frag64(tl, n, 2 * sub + ks, h), dsf, g0);
          g1 = mma(frag64(tl, 32 + n, 2 * sub + ks, h), dsf, g1);
        }
      }
    }
    if (more) {
      tile64_store(sm[(it + 1) & 1][0], kr, tid); tile64_store(sm[(it + 1) & 1][1], vr, tid); tile64_store(sm[(it + 1) & 1][2], tr, tid);
    }
    __syncthreads();
  }
  if (!active) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] *= p.alpha; g1[r] *= p.alpha; }
  bf16_t* grow = p.dqkv + ((long)b * p.T + q0 + n) * p.ldg + head * 192 + 8 * h;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    *(uint4*)(grow + 16 * s) = rows_to_b(g0, s);
    *(uint4*)(grow + 32 + 16 * s) = rows_to_b(g1, s);
  }
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_lds_kernel(AttnBP p) {
  // per stage (32 queries): Q and dO rows [32][64] (144-byte pitch), Q^T and dO^T [64][32] (80-byte pitch)
  __shared__ __attribute__((aligned(16))) unsigned char sm[2][2 * 32 * AT_P64 + 2 * 64 * AT_P32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
  const int key0 = (blockIdx.x * 4 + wave) * 32;
  const bool active = key0 < p.T;
  const int krow = active ? key0 + n : n;
  const int head = blockIdx.y, b = blockIdx.z;
  const bf16_t* base = p.qkv + (long)b * p.T * p.ld + head * 192;
  uint4 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = *(const uint4*)(base + 64 + (long)krow * p.ld + 8 * h + 16 * s);
    vf[s] = *(const uint4*)(base + 128 + (long)krow * p.ld + 8 * h + 16 * s);
  }
  const bf16_t* dob = p.dO + (long)b * p.T * p.lddo + head * 64;
  const long hrow = ((long)b * p.heads + head) * p.T;
  const bf16_t* qtb = p.qt + hrow * 64;
  const bf16_t* dtb = p.dot + hrow * 64;
  constexpr int OQ = 0, OD = 32 * AT_P64, OQT = 2 * 32 * AT_P64, ODT = OQT + 64 * AT_P32;
  f32x16 k0a, k1a, v0a, v1a;
#pragma unroll
  for (int r = 0; r < 16; ++r) { k0a[r] = 0.f; k1a[r] = 0.f; v0a[r] = 0.f; v1a[r] = 0.f; }
  uint4 qr, dr, qtr, dtr;
  auto load_stage = [&](int q0) {
    qr = tile32x64_load(base + (long)q0 * p.ld, p.ld, tid);
    dr = tile32x64_load(dob + (long)q0 * p.lddo, p.lddo, tid);
    qtr = tile32_load(qtb + q0, p.T, tid);
    dtr = tile32_load(dtb + q0, p.T, tid);
  };
  auto store_stage = [&](unsigned char* s_) {
    tile32x64_store(s_ + OQ, qr, tid); tile32x64_store(s_ + OD, dr, tid);
    tile32_store(s_ + OQT, qtr, tid); tile32_store(s_ + ODT, dtr, tid);
  };
  load_stage(0);
  store_stage(sm[0]);
  __syncthreads();
  const int ntile = p.T >> 5;
  for (int it = 0; it < ntile; ++it) {
    const unsigned char* sl = sm[it & 1];
    const int q0 = it * 32;
    const bool more = it + 1 < ntile;
    if (more) load_stage(q0 + 32);
    if (active) {
      float4 Lr[4], Dr[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {          // rows (queries) q0 + 8g + 4h + j of this lane
        Lr[g] = *(const float4*)(p.lse + hrow + q0 + 8 * g + 4 * h);
        Dr[g] = *(const float4*)(p.D + hrow + q0 + 8 * g + 4 * h);
      }
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        st = mma(frag64(sl + OQ, n, s, h), kf[s], st);
        dp = mma(frag64(sl + OD, n, s, h), vf[s], dp);
      }
      f32x16 ds;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float lr[4] = {Lr[g].x, Lr[g].y, Lr[g].z, Lr[g].w}, dd[4] = {Dr[g].x, Dr[g].y, Dr[g].z, Dr[g].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pr = ex2(st[4 * g + j] * p.c - lr[j]);
          st[4 * g + j] = pr;
          ds[4 * g + j] = pr * (dp[4 * g + j] - dd[j]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint4 pB = rows_to_b(st, ks), dB = rows_to_b(ds, ks);
        v0a = mma(frag32(sl + ODT, n, ks, h), pB, v0a); v1a = mma(frag32(sl + ODT, 32 + n, ks, h), pB, v1a);
        k0a = mma(frag32(sl + OQT, n, ks, h), dB, k0a); k1a = mma(frag32(sl + OQT, 32 + n, ks, h), dB, k1a);
      }
    }
    if (more) store_stage(sm[(it + 1) & 1]);
    __syncthreads();
  }
  if (!active) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) { k0a[r] *= p.alpha; k1a[r] *= p.alpha; }
  bf16_t* grow = p.dqkv + ((long)b * p.T + key0 + n) * p.ldg + head * 192 + 8 * h;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    *(uint4*)(grow + 64 + 16 * s) = rows_to_b(k0a, s);
    *(uint4*)(grow + 96 + 16 * s) = rows_to_b(k1a, s);
    *(uint4*)(grow + 128 + 16 * s) = rows_to_b(v0a, s);
    *(uint4*)(grow + 160 + 16 * s) = rows_to_b(v1a, s);
  }
}

// out[b][head][d][t] = in[b][t][head * hstride + off + d], d < 64; one block = 64 tokens of one (image, head)
__global__ __launch_bounds__(256) void head_transpose_kernel(const bf16_t* __restrict__ in, long ld, int hstride, int off, int T, int heads,
                                                            bf16_t* __restrict__ out) {
  __shared__ bf16_t tile[64][72];
  const int t0 = blockIdx.x * 64, head = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const bf16_t* src = in + ((long)b * T + t0) * ld + head * hstride + off;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = tid + i * 256, t = v >> 3, c = (v & 7) * 8;
    *(uint4*)&tile[t][c] = *(const uint4*)(src + (long)t * ld + c);
  }
  __syncthreads();
  bf16_t* dst = out + ((long)(b * heads + head) * 64) * T + t0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = tid + i * 256, d = v >> 3, t = (v & 7) * 8;
    bf16_t w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = tile[t + e][d];
    *(uint4*)(dst + (long)d * T + t) = *(const uint4*)w;
  }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, long n, bf16_t* __restrict__ y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (bf16_t)(pack_bf16x2(x[i], 0.f) & 0xffffu);
}

}  // namespace

int f32_to_bf16(hipStream_t st, const float* x, long n, void* y) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st, x, n, (bf16_t*)y);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

bool attn_fused_eligible(DType dt, int T, int head_channels, long ld_qkv) {
  return dt == DT_BF16 && head_channels == 64 && T >= 64 && T % 64 == 0 && ld_qkv % 8 == 0;
}

int head_transpose(hipStream_t st, const void* in, long ld, int hstride, int off, int B, int T, int heads, void* out) {
  hipLaunchKernelGGL(head_transpose_kernel, dim3(T / 64, heads, B), dim3(256), 0, st, (const bf16_t*)in, ld, hstride, off, T, heads, (bf16_t*)out);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

int attn_fused_forward(hipStream_t st, const void* qkv, long ld, int B, int T, int heads, void* vt_ws, void* o, long ldo, float* lse) {
  KDIP_REQUIRE(T % 64 == 0 && ld % 8 == 0 && ldo % 8 == 0 && (uintptr_t)qkv % 16 == 0 && (uintptr_t)o % 16 == 0, "fused attention: T must be a multiple of 64, rows 16-byte aligned");
  int rc = head_transpose(st, qkv, ld, 192, 128, B, T, heads, vt_ws);
  if (rc) return rc;
  AttnP p{(const bf16_t*)qkv, ld, (const bf16_t*)vt_ws, (bf16_t*)o, ldo, lse, T, heads, 1.4426950408889634f / 8.f};
  if (KDIP_ATTN_LDS) hipLaunchKernelGGL(attn_fwd_lds_kernel, dim3((T + 127) / 128, heads, B), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(attn_fwd_kernel, dim3((T + 127) / 128, heads, B), dim3(256), 0, st, p);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// ws: 3 * B*T*heads*64 bf16 (K^T, Q^T, dO^T); D: B*heads*T floats
int attn_fused_backward(hipStream_t st, const void* qkv, long ld, const void* dO, long lddo, const void* o, long ldo, const float* lse,
                        int B, int T, int heads, void* ws, float* D, void* dqkv, long ldg) {
  KDIP_REQUIRE(T % 64 == 0 && ld % 8 == 0 && lddo % 8 == 0 && ldg % 8 == 0, "fused attention backward: T must be a multiple of 64, rows 16-byte aligned");
  const size_t per = (size_t)B * T * heads * 64;
  bf16_t* kt = (bf16_t*)ws; bf16_t* qt = kt + per; bf16_t* dot = qt + per;
  int rc = head_transpose(st, qkv, ld, 192, 64, B, T, heads, kt);
  if (!rc) rc = head_transpose(st, qkv, ld, 192, 0, B, T, heads, qt);
  if (!rc) rc = head_transpose(st, dO, lddo, 64, 0, B, T, heads, dot);
  if (rc) return rc;
  AttnBP p{(const bf16_t*)qkv, ld, (const bf16_t*)dO, lddo, (const bf16_t*)o, ldo, lse, D, kt, qt, dot, (bf16_t*)dqkv, ldg, T, heads,
           1.4426950408889634f / 8.f, 0.125f};
  const dim3 grid((T + 127) / 128, heads, B);
  if (KDIP_ATTN_LDS) {
    hipLaunchKernelGGL(attn_bwd_dq_lds_kernel, grid, dim3(256), 0, st, p);
    hipLaunchKernelGGL(attn_bwd_dkv_lds_kernel, grid, dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, st, p);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, st, p);
  }
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

}  // namespace kdip
