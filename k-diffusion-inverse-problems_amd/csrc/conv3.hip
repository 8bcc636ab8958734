// bf16 implicit-GEMM 3x3 convolution for the large feature maps of the UNet (W >= 32, Cout % 128 == 0) on CDNA4 MFMA,
// second generation.  Replaces nn.Conv2d(3x3, pad 1) of guided_diffusion/unet.py:182-222 and its input-gradient, WITH the
// GroupNorm + FiLM + SiLU that precedes every such conv in a ResBlock (unet.py:183-184,207-208,249-253) applied while the
// input patch is staged (forward), and with the GroupNorm backward `dx = a*dz - (k0 + k1*x)` applied while the dgrad conv's
// input patch is staged (VJP): the activated tensor / the GroupNorm input-gradient never exist in HBM.
//
// GEMM view per block: D[cout 128][pixel 256] += W[cout][k] * X[k][pixel], k = (tap, cin).
//   * orientation: cout is the MFMA ROW index, pixels the COLUMN index, so after v_mfma_f32_32x32x16_bf16 a lane owns, for
//     ONE pixel (lane & 31), channel quads 8q + 4(lane >> 5) + {0..3}: the epilogue packs them to bf16 in registers, pairs
//     quads with v_permlane32_swap into 16-byte vectors and stores straight to NHWC -- no LDS transpose, no barrier.
//   * block = 4 waves, tile = 8 x 32 pixels x 128 cout, wave tile = 4 pixel rows (4 MFMA column tiles) x 64 cout;
//     2 blocks per CU (<= 256 VGPRs, 70 KB LDS each).
//   * X (activations): a 10 x 34 halo patch x 32 channels per chunk, double-buffered in LDS with an 80-byte pixel pitch: an
//     MFMA column tile is one 32-pixel patch row, so every 16-lane group of ds_read_b128 hits 16 distinct 16-byte slots
//     (conflict free; the first-generation kernel's 16-wide patches collided on 2 of 16 slots).  Staging goes through
//     registers because the GroupNorm transform is applied on the way (packed back to bf16, halo / padding pixels stay zero).
//   * W (weights): the host-packed fragment order of conv.hip ([tap][k-step][n-tile][lane][16 B]) is streamed stage by stage
//     (one tap of one 32-channel chunk = 8 KiB) into a 2-slot LDS ring with global_load_lds_dwordx4 (no VGPR round trip);
//     the two waves that share a cout half read each fragment from LDS instead of each pulling it through the CU's L1
//     (the first-generation kernel needed the full 64 B/clk of the vector cache for the B fragments alone).
//   * one barrier per stage (16 MFMAs per wave); the weight DMA of stage s+1 and the patch loads of chunk c+1 are in flight
//     under the MFMAs of stage s.
//   * epilogue: bias, residual (optionally read through a fused nearest x2 upsample), bf16 store, and the GroupNorm
//     forward sums (mode 1) or backward sums (mode 2) of the OUTPUT accumulated per lane and combined once per block.
#include <atomic>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "conv3p.h"

namespace kdip {

namespace {

constexpr int C3_TH = 8, C3_TW = 32, C3_BM = C3_TH * C3_TW, C3_BN = 128;
constexpr int C3_PW = C3_TW + 2, C3_PH = C3_TH + 2, C3_NPIX = C3_PW * C3_PH;   // 34 x 10 = 340 halo pixels
constexpr int C3_PIXB = 80;                               // LDS pixel pitch: 64 B of channels + 16 B pad
constexpr int C3_ABUF = C3_NPIX * C3_PIXB;                // 27200 B per patch buffer
constexpr int C3_BSLOT = 8192;                            // one weight stage: 2 k-steps x 4 n-tiles x 1 KiB
constexpr int C3_NSLOT = 3;                               // weight ring: stage s+2 is in flight while stage s is consumed
constexpr int C3_MAXV = (C3_NPIX * 4 + 255) / 256;        // staged 16-byte vectors per thread (6)
constexpr int C3_LDS = 2 * C3_ABUF + C3_NSLOT * C3_BSLOT; // 78976 B (+ 64 B dummy slot): two blocks per CU
// The per-channel coefficients of the staging transform ride in the 16-byte PADS of the patch pixels (never touched by the
// staging writes or the fragment reads): table slot j = pad of pixel j % 336 of patch buffer j / 336 -> 672 slots.
// TF 1: slot = 2 channels x (a, b); TF 2: slot = 1 channel x (a, b, k0, k1).
constexpr int C3_LDS_TOTAL = C3_LDS + 64 + 512;           // + dummy staging slot + statistics exchange [4 waves][2][16] floats
#ifndef C3_CARRY
#define C3_CARRY 1             // TF == 0: a stage's k-step-1 MFMAs are issued after the NEXT stage's barrier (explicit software pipeline)
#endif
#ifndef C3_KLOOP_PRIO
#define C3_KLOOP_PRIO 2        // wave priority inside the K loop (epilogue / tile bookkeeping run at 0): the two blocks of a CU share each
#endif                         // SIMD's VALU issue port, and an older block's epilogue VALU stream otherwise starves the younger block's MFMAs
#ifndef C3_NT_STORE
#define C3_NT_STORE 0          // non-temporal epilogue stores: measured 41.1 -> 43.4 ms per step (the 16-byte pieces of a line no longer merge in L2)
#endif
#ifndef C3_NT_AUX
#define C3_NT_AUX 0            // non-temporal loads of the epilogue's residual / GroupNorm-input rows: measured 41.1 -> 42.1 ms per step
#endif
#ifndef C3_STAGGER
#define C3_STAGGER 0            // experiment: start delay of block j = (j / 8 % 16) * C3_STAGGER * 64 cycles (de-synchronises the chip-wide epilogue bursts)
#endif
#ifndef C3_ANTIPHASE
#define C3_ANTIPHASE 0          // experiment: the second-resident block of a CU (LDS base != 0) starts C3_ANTIPHASE x 8128 cycles late, so that
#endif                          // one block's epilogue (VALU / memory) runs beside the other block's K loop (MFMA)
#ifndef C3_BLOCKS_PER_CU
#define C3_BLOCKS_PER_CU 2
#endif
constexpr int C3_TABPIX = 336, C3_TABSLOTS = 2 * C3_TABPIX;
// TF 2 keeps one float4 (a, b, k0, k1) per channel and uses 320 pixels per buffer, so that a 32-channel chunk never straddles the
// two buffers; within a chunk channel 8g + e sits in slot 4e + g: the four lane groups g = tid & 3 that read coefficient e of their
// channel together hit four consecutive pads (80 bytes apart = banks 0-3, 20-23, 40-43, 60-63: conflict-free; the channel-major
// order put groups 0 / 2 and 1 / 3 on the same banks -- 39 % of the LDS cycles of the TF 2 launches were conflict cycles)
constexpr int C3_TABPIX2 = 320;
constexpr int c3_max_cin(int tf) { return tf == 1 ? 2 * C3_TABSLOTS : (tf == 2 ? 2 * C3_TABPIX2 : (1 << 20)); }
__device__ __forceinline__ int c3_tab_off(int slot) { return (slot / C3_TABPIX) * C3_ABUF + (slot % C3_TABPIX) * C3_PIXB + 64; }
__device__ __forceinline__ int c3_tab_off2(int slot) { return (slot / C3_TABPIX2) * C3_ABUF + (slot % C3_TABPIX2) * C3_PIXB + 64; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef C3_ABL_NOATOM
#define C3_ABL_NOATOM 0        // timing ablations (tools/): statistics atomics off / staging transform math off / no epilogue
#endif
#ifndef C3_ABL_NODMA
#define C3_ABL_NODMA 0         // (K-loop ablations, TF 0 only, results are garbage: no weight DMA / no patch staging / no stage barrier /
#endif                         //  no fragment reads)
#ifndef C3_ABL_NOSTG
#define C3_ABL_NOSTG 0
#endif
#ifndef C3_ABL_NOBAR
#define C3_ABL_NOBAR 0
#endif
#ifndef C3_ABL_NOLDSR
#define C3_ABL_NOLDSR 0
#endif
#ifndef C3_ABL_EPI_NOSILU
#define C3_ABL_EPI_NOSILU 0    // backward-statistics epilogue without the silu' transcendentals (timing ablation)
#endif
#ifndef C3_ABL_NOTF
#define C3_ABL_NOTF 0
#endif
#ifndef C3_ABL_NOEPI
#define C3_ABL_NOEPI 0
#endif
#ifndef C3_SCHED_GROUPS
#define C3_SCHED_GROUPS 1      // pin the fragment-read / MFMA issue order of the plain (TF 0) kernels
#endif
#ifndef C3_AUX_PREFETCH
#define C3_AUX_PREFETCH 0      // measured: touching the epilogue's lines from the K loop does not pay (+5 %: the epilogue is not latency bound)
#endif
#ifndef C3_TIMING
#define C3_TIMING 0            // diagnostic build: per-block phase stamps (100 MHz) into the buffer set by conv3_debug_timing
#endif
#ifndef C3_STRICT_LGKM
#define C3_STRICT_LGKM 0       // 1: drain the LDS queue in front of every stage barrier (validation builds)
#endif


unsigned long long* g_c3_dbg = nullptr;
#if C3_TIMING
#define C3_STAMP(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(long)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define C3_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ f32x16 mfma_bf16(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void c3_store(void* p, uint4 o) {
  if (C3_NT_STORE) { u32x4 v = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(v, (u32x4*)p); }
  else *(uint4*)p = o;
}
__device__ __forceinline__ uint4 c3_aux_load(const void* p) {
  if (C3_NT_AUX) { u32x4 v = __builtin_nontemporal_load((const u32x4*)p); return make_uint4(v[0], v[1], v[2], v[3]); }
  return *(const uint4*)p;
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// TF: staging transform (0 none, 1 GroupNorm forward apply, 2 GroupNorm backward apply); STM: statistics mode of the output
//
// PERSISTENT blocks: (2 per CU) x (CUs) workgroups walk the tile list; XCD k (= blockIdx % 8) owns a contiguous tile range
// (neighbouring halos and all n-blocks of an m-tile share one L2).  The staging pipeline runs ACROSS tile boundaries: during a
// tile's last chunk the vectors being staged and the weight DMAs of the last two stages already belong to the block's next
// tile, so a tile starts without the load -> transform -> LDS -> barrier prologue (3.6 - 10 us of a 30 - 45 us tile).
template <int TF, int STM, bool RES>
__global__ __launch_bounds__(256, 2) void conv3_kernel(Conv3Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = p.mtiles * p.nblkN;
  const int xq = ntiles >> 3, xr = ntiles & 7, xcd = blockIdx.x & 7;
  const int xstart = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq, xend = xstart + xq + (xcd < xr ? 1 : 0);
  const int nper = gridDim.x >> 3;                       // blocks per XCD share
  int cur = xstart + (blockIdx.x >> 3);
  if (cur >= xend) return;
  if (C3_ANTIPHASE) {
    if (__builtin_amdgcn_s_getreg(0x3806) != 0) { for (int i = 0; i < C3_ANTIPHASE; ++i) __builtin_amdgcn_s_sleep(127); }      // HW_REG_LDS_ALLOC.LDS_BASE
  }
  if (C3_STAGGER) { for (int i = (int)((blockIdx.x >> 3) & 15); i > 0; --i) __builtin_amdgcn_s_sleep(C3_STAGGER); }
  C3_STAMP(0);
#if C3_TIMING
  if (p.dbg && threadIdx.x == 0 && STM == 0)              // slot 7 (unused without statistics): shader-clock counter at block start
    p.dbg[(long)blockIdx.x * 8 + 7] = __builtin_readcyclecounter();
#endif
  const int tpi = p.tilesX * p.tilesY;
  const int Hs = p.in_ups ? p.H >> 1 : p.H, Ws = p.in_ups ? p.W >> 1 : p.W;
  const int nchunks = p.Cin >> 5;
  const long kStride = (long)p.ntilesN * 64, tapStride = (long)(p.Cin >> 4) * kStride;     // packed weights, in 16-byte units

  // ---- staging descriptors of the tile whose chunks are being STAGED (the current tile, or -- during its last chunk -- the
  // next one): vector v = tid + 256 i -> (halo pixel v >> 2, 16-byte channel group tid & 3).  Byte offsets are relative to the
  // image (32 bits); padding / out-of-patch lanes read offset 0 and are zeroed when written, so every wave issues exactly the
  // same number of loads (the K loop counts its outstanding loads by hand, see below).
  unsigned goff[C3_MAXV];
  unsigned zmask = 0;                  // bit i: vector i is zero fill (conv padding); bit 8 + i: vector i is outside the patch
  const bf16_t* ximg = p.x;
  const bf16_t* x2img = p.x;
  int tab_img = -1;                    // image whose transform coefficients sit in the LDS table
  auto set_staging = [&](int bid) {
    const int mtile = bid / p.nblkN;
    const int im = mtile / tpi, trem = mtile - im * tpi;
    const int ty = trem / p.tilesX;
    const int yy = ty * C3_TH, xx = (trem - ty * p.tilesX) * C3_TW;
    ximg = p.x + (long)im * Hs * Ws * p.ldx;
    if (TF == 2) x2img = p.x2 + (long)im * Hs * Ws * p.ldx;          // same geometry as x (checked on the host)
    zmask = 0;
#pragma unroll
    for (int i = 0; i < C3_MAXV; ++i) {
      const int v = tid + i * 256, pix = v >> 2;
      const int hy = pix / C3_PW, hx = pix - hy * C3_PW;
      const int gy = yy + hy - 1, gx = xx + hx - 1;
      goff[i] = (unsigned)((tid & 3) * 16);
      if (pix >= C3_NPIX) zmask |= 0x101u << i;
      else if (gy < 0 || gy >= p.H || gx < 0 || gx >= p.W) zmask |= 1u << i;
      else {
        const int spix = p.in_ups ? (gy >> 1) * Ws + (gx >> 1) : gy * Ws + gx;
        goff[i] = (unsigned)((spix * (int)p.ldx + (tid & 3) * 8) * 2);
      }
    }
    return im;
  };
  // per-channel coefficients of the staging transform -> pads of the patch pixels (see c3_tab_off); block-uniform call
  auto load_table = [&](int im) {
    if (TF == 1) {
      if (p.fold_stats) {      // coefficients from the statistics (no gn_coef launch in front of this conv); also left in HBM for the VJP
        const int cpg = p.Cin >> 5;
        for (int j = tid; j < p.Cin / 2; j += 256) {
          float4 k; float m0, r0, m1, r1;
          c3_fold_coef_fwd(p, im, 2 * j, k.x, k.y, m0, r0);
          c3_fold_coef_fwd(p, im, 2 * j + 1, k.z, k.w, m1, r1);
          *(float4*)(smem + c3_tab_off(j)) = k;
          ((float4*)(p.fold_coef_out + (long)im * p.Cin * 2))[j] = k;
          if ((2 * j) % cpg == 0) *(float2*)(p.fold_mr_out + ((long)im * 32 + (2 * j) / cpg) * 2) = make_float2(m0, r0);
          if ((2 * j + 1) % cpg == 0) *(float2*)(p.fold_mr_out + ((long)im * 32 + (2 * j + 1) / cpg) * 2) = make_float2(m1, r1);
        }
      } else {
        const float4* src = (const float4*)(p.tf_coef + (long)im * p.Cin * 2);
        for (int j = tid; j < p.Cin / 2; j += 256) *(float4*)(smem + c3_tab_off(j)) = src[j];
      }
    } else if (TF == 2) {
      const float4* src = (const float4*)(p.tf_coef + (long)im * p.Cin * 4);
      for (int j = tid; j < p.Cin; j += 256)
        *(float4*)(smem + c3_tab_off2((j & ~31) + (j & 7) * 4 + ((j >> 3) & 3))) = p.fold_stats ? c3_fold_coef_bwd(p, im, j) : src[j];
    }
    tab_img = im;
  };

  // ---- staging of the NEXT chunk under the MFMAs of the current one.  Vector i is requested at tap i (i = 0..5) and
  // transformed + written to the other patch buffer at tap i+3: three stage times (~2k cycles) cover the HBM latency.  The
  // loads are inline asm so that hipcc neither waits for them itself (next to LDS-DMA traffic it would drain the whole queue
  // with vmcnt(0)) nor touches their destination registers before the counted wait at the top of the consuming stage.
  // (TF 2 stages two tensors: it keeps two vectors per tensor in flight (written 2 stages after the request) instead of three,
  // the third set did not fit the register file next to the transform's temporaries)
  constexpr int DIST = TF == 2 ? 2 : 3;
  u32x4 sa[3], sb[3];                                // (unused sets are dead code)
  auto vec_load_asm = [&](int cbytes, int i) {      // chunk byte offset (chunk * 64), vector i -> register set i % DIST
    const unsigned off = goff[i] + (unsigned)cbytes;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sa[i % DIST]) : "v"(off), "s"(ximg) : "memory");
    if (TF == 2) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sb[i % DIST]) : "v"(off), "s"(x2img) : "memory");
  };
  auto transform = [&](int c, int i, uint4 o, uint4 o2) -> uint4 {
    if (TF == 1) {
      const int slot = (c * 32 + (tid & 3) * 8) >> 1;                       // 4 consecutive slots: (a, b) of 2 channels each
      const unsigned char* tab = smem + c3_tab_off(slot);
      float f[8];
      unpack16<bf16_t>(o, f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 k = *(const float4*)(tab + j * C3_PIXB);
        const float z0 = k.x * f[2 * j] + k.y, z1 = k.z * f[2 * j + 1] + k.w;
        f[2 * j] = C3_ABL_NOTF ? z0 : silu_fast(z0);
        f[2 * j + 1] = C3_ABL_NOTF ? z1 : silu_fast(z1);
      }
      o = pack16<bf16_t>(f);
    } else if (TF == 2) {
      const unsigned char* tab = smem + c3_tab_off2(c * 32 + (tid & 3));     // channel 8 (tid & 3) + e of the chunk: slot 4e + (tid & 3)
      float fd[8], fx[8];
      unpack16<bf16_t>(o, fd);
      unpack16<bf16_t>(o2, fx);
#pragma unroll
      for (int e = 0; e < 8; ++e) {          // the staged tensor already holds dz = dy * silu'(z) (written by the producing epilogue)
        const float4 k = *(const float4*)(tab + e * 4 * C3_PIXB);
        fd[e] = k.x * fd[e] - (k.z + k.w * fx[e]);
      }
      o = pack16<bf16_t>(fd);
    }
    if ((zmask >> i) & 1) o = make_uint4(0, 0, 0, 0);      // conv zero padding applies to the TRANSFORMED tensor
    return o;
  };
  auto vec_store = [&](int buf, int i, uint4 o) {    // branch free: lanes outside the patch (vector 5, tid >= 80) write a dummy slot
    const int pix = (tid + i * 256) >> 2;
    int addr = buf * C3_ABUF + pix * C3_PIXB + (tid & 3) * 16;
    if (i * 256 + 255 >= C3_NPIX * 4) addr = ((zmask >> (8 + i)) & 1) ? C3_LDS + (tid & 3) * 16 : addr;
    *(uint4*)(smem + addr) = o;
  };

  // ---- weight stream: stage (chunk c, tap t) = k-steps 2c, 2c+1 of tap t, n-tiles 4 nb .. 4 nb + 3, ring slot t % 3
  // (9 taps per chunk: the slot of stage 9c + t is t % 3, also across tile boundaries).  The DMAs are inline asm as well: next
  // to a builtin LDS-DMA hipcc drains the whole memory queue (vmcnt(0)) in front of the next ds_read of the same LDS object,
  // i.e. once per stage.  M0 (the LDS destination base) is saved / restored around the two transfers; address = wave-uniform
  // stage base (SGPR pair) + this thread's constant 32-bit byte offset.
  const unsigned w_voff = (unsigned)(tid * 16);
  const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + 2 * C3_ABUF + wave * 1024;
  auto dma_b = [&](int nb_, int c, int tap, int slot) {
    const uint4* g0 = p.wp + (long)nb_ * 256 + tap * tapStride + (long)(2 * c) * kStride;
    const uint4* g1 = g0 + kStride;
    const unsigned l0 = lds_b + slot * C3_BSLOT, l1 = l0 + 4096;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(w_voff), "s"(g0), "s"(g1), "s"(l0), "s"(l1) : "memory");
  };

  const int a_lane = (wm * 4 * C3_PW + (lane & 31)) * C3_PIXB + (lane >> 5) * 16;   // + (mt + ty) * PW * PIXB + tx * PIXB + ks * 32
  const int b_lane = 2 * C3_ABUF + (wn * 2 * 64 + lane) * 16;                       // + slot * BSLOT + (ks * 4 + nt) * 1024
  constexpr int NV = TF == 2 ? 2 : 1;                  // staging loads per vector
  // The epilogue reads a second tensor (residual / GroupNorm input of the produced gradient) whose HBM latency would sit exposed
  // behind the last MFMA: every chunk, taps 6 and 7 (no staging loads there) touch the two 128-byte lines of this thread's tile
  // pixel, so the epilogue's loads are served by L2 / Infinity Cache.  Results are discarded; the loads are counted like the others.
  constexpr bool AUXPF = C3_AUX_PREFETCH && (RES || STM == 2);
  unsigned pf0 = 0, pf1 = 0;
  const int cpg = p.Cout >> 5;

  // ---- prologue of the block's FIRST tile: weight stages 0 and 1 in flight; coefficient table -> pads; the whole first patch
  // (compiler-managed loads) into patch buffer 0
  {
    const int nb0 = cur % p.nblkN;
    dma_b(nb0, 0, 0, 0);
    dma_b(nb0, nchunks > 1 || true ? 0 : 0, 1, 1);
    const int im = set_staging(cur);
    if (TF) { load_table(im); __syncthreads(); }
    uint4 o[C3_MAXV], o2[C3_MAXV];
#pragma unroll
    for (int i = 0; i < C3_MAXV; ++i) {
      o[i] = *(const uint4*)((const char*)ximg + goff[i]);
      o2[i] = make_uint4(0, 0, 0, 0);
      if (TF == 2) o2[i] = *(const uint4*)((const char*)x2img + goff[i]);
    }
#pragma unroll
    for (int i = 0; i < C3_MAXV; ++i) vec_store(0, i, transform(0, i, o[i], o2[i]));
  }
  int pb = 0;                                            // patch buffer that holds chunk 0 of the current tile
  bool first = true;
  uint4 pa[4];                                           // k-step-0 patch fragments of the next stage (carried across stage barriers)
  // CARRY: the k-step-1 fragments of a stage are consumed after the NEXT stage's barrier, so that the only reads that must wait
  // for a barrier (the weight fragments of k-step 0) have 8 MFMAs of older work to hide their LDS round trip under
  constexpr bool CARRY = C3_CARRY && TF == 0;
  uint4 a1c[4], b1c[2];

  for (;;) {
    // ---- current tile
    const int mtile = cur / p.nblkN, nb = cur - mtile * p.nblkN;
    const int img = mtile / tpi, trem = mtile - img * tpi;
    const int ty0 = trem / p.tilesX;
    const int y0 = ty0 * C3_TH, x0 = (trem - ty0 * p.tilesX) * C3_TW;
    const int nxt = cur + nper;
    const bool has_next = nxt < xend;
    const int nb_n = has_next ? nxt % p.nblkN : nb;
    if (first) C3_STAMP(1);

    // accumulators start at the bias (lane (pixel, h) owns channels nt*32 + 8q + 4h + j in registers 4q + j): no bias pass later.
    // (Lane-derived values of the tile prologue / epilogue come from an opaque copy of the thread index: otherwise hipcc hoists
    // every such address out of the tile loop and keeps it live across the K loop -- hundreds of bytes of spills.)
    int tv = threadIdx.x;
    asm volatile("" : "+v"(tv));
    const bf16_t* aux_img = p.y;      // image base of the tensor the epilogue reads (wave-uniform) + this thread's pixel offset
    unsigned aux_off = 0;
    if (AUXPF) {
      const int py = y0 + ((tv >> 5) & 7), px = x0 + (tv & 31);
      if (RES) {
        const int Hr = p.res_ups ? p.H >> 1 : p.H, Wr = p.res_ups ? p.W >> 1 : p.W;
        aux_img = p.res + (long)img * Hr * Wr * p.ldr + nb * C3_BN;
        aux_off = (unsigned)(((p.res_ups ? (py >> 1) * Wr + (px >> 1) : py * Wr + px) * (int)p.ldr) * 2);
      } else {
        aux_img = p.st_x + (long)img * p.H * p.W * p.st_ldx + nb * C3_BN;
        aux_off = (unsigned)(((py * p.W + px) * (int)p.st_ldx) * 2);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { sa[j] = (u32x4){0, 0, 0, 0}; sb[j] = (u32x4){0, 0, 0, 0}; }
    pf0 = 0; pf1 = 0;     // new definitions: nothing of the staging
                                                                                                   // registers lives across the epilogue
    f32x16 acc[4][2];
    {
    const int h = (tv >> 5) & 1;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) bq = *(const float4*)(p.bias + nb * C3_BN + wn * 64 + nt * 32 + 8 * q + 4 * h);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          acc[mt][nt][4 * q + 0] = bq.x; acc[mt][nt][4 * q + 1] = bq.y; acc[mt][nt][4 * q + 2] = bq.z; acc[mt][nt][4 * q + 3] = bq.w;
        }
      }
    }

    if (C3_KLOOP_PRIO) __builtin_amdgcn_s_setprio(C3_KLOOP_PRIO);
    // Per stage (chunk c, tap t), every wave issues, in this order: 2 weight DMAs (stage s+2), then NV staging loads if t < 6.
    // At the top of stage s the weights of stage s (issued at s-2) and the staging vector written in this stage (issued at
    // s-3) must have landed; still allowed in flight: the staging loads of s-2 and everything of s-1
    //   -> vmcnt(2 + nv(t-1) + nv(t-2)), nv(t) = NV for t < 6 else 0 (taps wrap within the 9-tap chunk); with DIST = 2 the vector
    //   written in stage s was requested at s-2 AFTER that stage's DMAs, so only stage s-1 may be in flight: vmcnt(2 + nv(t-1)).
    // The first stage of a continued tile follows an epilogue with compiler-managed loads / stores in the queue: it drains
    // everything (vmcnt(0)).  The raw s_barrier that follows the wait (a) publishes every wave's landed DMA of stage s and the
    // patch writes of earlier stages, (b) guarantees all waves are done reading ring slot (s-1) % 3 before it is refilled.
#define C3_NVT(t) (((t) + 9) % 9 < 6 ? NV : ((AUXPF && ((t) + 9) % 9 < 8) ? 1 : 0))
    if (!first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile boundary: the epilogue's stores / loads leave the queue
    for (int c = 0; c < nchunks; ++c) {
      const bool last = c + 1 == nchunks;
      if (last && has_next) {                            // from here on the staging loads belong to the next tile
        const int im = set_staging(nxt);
        if (TF && im != tab_img) {                       // (rare: the block's next tile lies in another image)
          asm volatile("; C3_RARE_BEGIN\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();                  // every wave is past its last transform of the current image
          load_table(im);
          asm volatile("; C3_RARE_END" ::: "memory");
        }
      }
      const unsigned char* ab = smem + ((pb + c) & 1) * C3_ABUF + a_lane;
      const int nbuf = (pb + c + 1) & 1;                 // patch buffer being staged
      const int cn = last ? (has_next ? 0 : c) : c + 1;  // chunk index (within its tile) being staged (no next tile: redundant re-loads keep the counts uniform)
      const int wc = last ? (has_next ? 0 : c) : c + 1, wnb = last ? nb_n : nb;     // chunk / n-block of the weight stages that wrap
      auto stage = [&](auto tapc) {
        constexpr int tap = decltype(tapc)::value;
        // lgkmcnt(0) (patch writes of taps 3..8 visible to the other waves) is only needed before a chunk's first stage;
        // draining the LDS queue in front of every barrier exposed the latency of the fragment reads hipcc hoists there
        constexpr bool LG = tap == 0 || C3_STRICT_LGKM;
        constexpr int NW = 2 + C3_NVT(tap - 1) + (DIST == 3 ? C3_NVT(tap - 2) : 0);
        if (TF == 2) {
          if (LG) asm volatile("s_waitcnt vmcnt(%6) lgkmcnt(0)" : "+v"(sa[0]), "+v"(sa[1]), "+v"(sb[0]), "+v"(sb[1]), "+v"(pf0), "+v"(pf1) : "n"(NW) : "memory");
          else asm volatile("s_waitcnt vmcnt(%6)" : "+v"(sa[0]), "+v"(sa[1]), "+v"(sb[0]), "+v"(sb[1]), "+v"(pf0), "+v"(pf1) : "n"(NW) : "memory");
        } else {
          if (LG) asm volatile("s_waitcnt vmcnt(%5) lgkmcnt(0)" : "+v"(sa[0]), "+v"(sa[1]), "+v"(sa[2]), "+v"(pf0), "+v"(pf1) : "n"(NW) : "memory");
          else asm volatile("s_waitcnt vmcnt(%5)" : "+v"(sa[0]), "+v"(sa[1]), "+v"(sa[2]), "+v"(pf0), "+v"(pf1) : "n"(NW) : "memory");
        }
        if (!C3_ABL_NOBAR) __builtin_amdgcn_s_barrier();
        if (!C3_ABL_NODMA) {
          constexpr int t2 = tap + 2 >= 9 ? tap + 2 - 9 : tap + 2;
          if (tap + 2 >= 9) dma_b(wnb, wc, t2, t2 % 3); else dma_b(nb, c, t2, t2 % 3);
        }
        if (!C3_ABL_NOSTG && tap >= DIST && tap < DIST + C3_MAXV) {   // (no next tile: the other buffer is dead, the redundant write is harmless and keeps the stage branch free)
          constexpr int i = tap >= DIST ? tap - DIST : 0;
          const uint4 o = __builtin_bit_cast(uint4, sa[i % DIST]);
          const uint4 o2 = TF == 2 ? __builtin_bit_cast(uint4, sb[i % DIST]) : make_uint4(0, 0, 0, 0);
          vec_store(nbuf, i, transform(cn, i, o, o2));
        }
        if (!C3_ABL_NOSTG && tap < C3_MAXV) vec_load_asm(cn * 64, tap);
        if (AUXPF && (tap == 6 || tap == 7)) {
          // mode 2: only the tile's LAST chunk touches the real lines (a few us ahead of the epilogue: still in L2 when it reads
          // them); the other chunks issue the same instruction on an L2-hot dummy line so that the counted waits stay uniform
          const bool real = C3_AUX_PREFETCH != 2 || last;
          const bf16_t* ab_ = real ? aux_img : (const bf16_t*)p.wp;
          const unsigned ao_ = real ? aux_off : 0u;
          if (tap == 6) asm volatile("global_load_dword %0, %1, %2" : "=v"(pf0) : "v"(ao_), "s"(ab_) : "memory");
          else asm volatile("global_load_dword %0, %1, %2 offset:128" : "=v"(pf1) : "v"(ao_), "s"(ab_) : "memory");
        }
        // explicit software pipeline of the fragment reads: the k-step-0 patch fragments of stage s+1 are requested under the
        // k-step-1 MFMAs of stage s (same chunk: the patch buffer is stable) and carried across the barrier in pa[]; only the
        // weight fragments wait for the barrier (their DMA is published by it)
        const unsigned char* bb = smem + b_lane + (tap % 3) * C3_BSLOT;
        constexpr int toff = ((tap / 3) * C3_PW + (tap % 3)) * C3_PIXB;
        constexpr int toffn = (((tap + 1) / 3) * C3_PW + ((tap + 1) % 3)) * C3_PIXB;
        if (tap == 0) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) pa[mt] = *(const uint4*)(ab + mt * C3_PW * C3_PIXB + toff);
        }
        if (CARRY) {
          // taps 1..8 issue the previous stage's k-step-1 MFMAs first; nothing is carried across a chunk boundary (tap 8 runs its own
          // k-step 1), where the tile / chunk bookkeeping needs the registers
          if (C3_ABL_NOLDSR) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_bf16(pa[nt], pa[mt], acc[mt][nt]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_bf16(pa[nt + 2], pa[mt], acc[mt][nt]);
            return;
          }
          uint4 b0[2];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) b0[nt] = *(const uint4*)(bb + nt * 1024);
          if (tap > 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_bf16(b1c[nt], a1c[mt], acc[mt][nt]);
          }
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) a1c[mt] = *(const uint4*)(ab + mt * C3_PW * C3_PIXB + toff + 32);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) b1c[nt] = *(const uint4*)(bb + (4 + nt) * 1024);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_bf16(b0[nt], pa[mt], acc[mt][nt]);
          if (tap < 8) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) pa[mt] = *(const uint4*)(ab + mt * C3_PW * C3_PIXB + toffn);
          } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_bf16(b1c[nt], a1c[mt], acc[mt][nt]);
          }
          if (C3_SCHED_GROUPS) {
            if (tap == 0) {
              __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            } else {
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, tap < 8 ? 8 : 16, 0);
              if (tap < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
          }
        } else {
        uint4 b0[2], b1[2], a1[4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b0[nt] = *(const uint4*)(bb + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a1[mt] = *(const uint4*)(ab + mt * C3_PW * C3_PIXB + toff + 32);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b1[nt] = *(const uint4*)(bb + (4 + nt) * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_bf16(b0[nt], pa[mt], acc[mt][nt]);
        if (tap < 8) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) pa[mt] = *(const uint4*)(ab + mt * C3_PW * C3_PIXB + toffn);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_bf16(b1[nt], a1[mt], acc[mt][nt]);
        if (C3_SCHED_GROUPS && TF == 0) {
        // pin the issue order of this region: all fragment reads of the stage first (their LDS latency then hides under the
        // MFMAs carried over from the previous stage and the k-step-0 MFMAs), the next stage's patch fragments under k-step 1
        __builtin_amdgcn_sched_group_barrier(0x100, tap == 0 ? 12 : 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        if (tap < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        }
      };
      stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{}); stage(std::integral_constant<int, 2>{});
      stage(std::integral_constant<int, 3>{}); stage(std::integral_constant<int, 4>{}); stage(std::integral_constant<int, 5>{});
      stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{}); stage(std::integral_constant<int, 8>{});
    }
#undef C3_NVT
    if (C3_KLOOP_PRIO) __builtin_amdgcn_s_setprio(0);
    pb = (pb + nchunks) & 1;
    if (first) C3_STAMP(2);

    if (C3_ABL_NOEPI) {     // timing ablation: keep the accumulators live, skip the epilogue
      float t = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) t += acc[mt][nt][r];
      if (t == 12345.678f) p.y[0] = (bf16_t)1;
    } else {
    // ---- epilogue (LDS only for the statistics combine, in its own region).  Addresses = wave-uniform 64-bit row bases + one
    // 32-bit per-lane byte offset per tensor (saddr + voffset loads / stores: no 64-bit vector arithmetic).  A lane owns, for its
    // pixel, channel quads 8q + 4h + {0..3} of each n-tile; residual / GroupNorm-input rows are fetched up front as 16-byte
    // vectors in the STORE layout (lane h: channels 16k + 8h .. +7) -- 32 contiguous bytes per pixel per instruction, all 16
    // loads in flight at once -- and brought to the accumulator layout with the inverse of the store's v_permlane32_swap pairing.
    int te = threadIdx.x;
    asm volatile("" : "+v"(te));
    const int h = (te >> 5) & 1, pl = te & 31;
    const int nbase = nb * C3_BN + wn * 64;                 // first channel of this wave
    const int row0 = y0 + wm * 4;                           // first pixel row of this wave
    char* const yb = (char*)(p.y + ((long)img * p.H + row0) * p.W * p.ldy + nbase);
    const unsigned rsy = (unsigned)(p.W * p.ldy * 2), lane_y = (unsigned)(((x0 + pl) * p.ldy + 8 * h) * 2);
    uint4 aux[4][2][2];                                     // [mt][nt][k]: residual (RES) or GroupNorm input (STM 2) vectors
    uint4 wst[STM == 2 ? 4 : 1][2][2];                      // STM 2: the stored (packed bf16) output vectors, store layout
    if (RES) {
      const int Hr = p.res_ups ? p.H >> 1 : p.H, Wr = p.res_ups ? p.W >> 1 : p.W;
      const char* const rb = (const char*)(p.res + (long)img * Hr * Wr * p.ldr + nbase);
      const unsigned rsr = (unsigned)(Wr * p.ldr * 2), lane_r = (unsigned)((((p.res_ups ? (x0 + pl) >> 1 : x0 + pl)) * p.ldr + 8 * h) * 2);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const char* r = rb + (unsigned)(p.res_ups ? (row0 + mt) >> 1 : row0 + mt) * rsr + lane_r;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int k = 0; k < 2; ++k) aux[mt][nt][k] = c3_aux_load(r + (nt * 32 + 16 * k) * 2);
      }
    } else if (STM == 2) {
      // GroupNorm-input rows in the STORE layout; they are consumed in the second sweep below, so their latency hides under
      // the pack / store sweep
      const char* const xb = (const char*)(p.st_x + ((long)img * p.H + row0) * p.W * p.st_ldx + nbase);
      const unsigned rsx = (unsigned)(p.W * p.st_ldx * 2), lane_x = (unsigned)(((x0 + pl) * p.st_ldx + 8 * h) * 2);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int k = 0; k < 2; ++k) aux[mt][nt][k] = c3_aux_load(xb + mt * rsx + lane_x + (nt * 32 + 16 * k) * 2);
    }
    float ss[16];                                           // [0..7]: sum 1 per channel quad, [8..15]: sum 2
#pragma unroll
    for (int i = 0; i < 16; ++i) ss[i] = 0.f;
    // ---- sweep 1: (residual,) pack, forward statistics, store.  Pixel row outermost: the four 16-byte vectors of a pixel (this
    // wave's 64 channels = one 128-byte run) are stored back to back so that L2 merges them into whole lines before they leave
    // for HBM (channel-vector-major order wrote 1.14 - 1.27 x the tensor: WRITE_SIZE of the in-network launches).
    f32x2 t1v[2][2][2], t2v[2][2][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) { (&t1v[0][0][0])[i] = (f32x2){0.f, 0.f}; (&t2v[0][0][0])[i] = (f32x2){0.f, 0.f}; }
    auto emit = [&](int mt, int nt, int k) {
          const unsigned coff = (unsigned)((nt * 32 + 16 * k) * 2);
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[mt][nt][8 * k + e];
          if (RES) {          // residual vector -> accumulator layout: (u0, u1) = quad 2k, (u2, u3) = quad 2k+1
            const uint4 a4 = aux[mt][nt][k];
            auto r = __builtin_amdgcn_permlane32_swap(a4.x, a4.z, false, false);
            const uint32_t u0 = r[0], u2 = r[1];
            r = __builtin_amdgcn_permlane32_swap(a4.y, a4.w, false, false);
            const uint32_t u1 = r[0], u3 = r[1];
            v[0] += bf_lo(u0); v[1] += bf_hi(u0); v[2] += bf_lo(u1); v[3] += bf_hi(u1);
            v[4] += bf_lo(u2); v[5] += bf_hi(u2); v[6] += bf_lo(u3); v[7] += bf_hi(u3);
          }
          uint32_t w0x = pack_bf16x2(v[0], v[1]), w0y = pack_bf16x2(v[2], v[3]);
          uint32_t w1x = pack_bf16x2(v[4], v[5]), w1y = pack_bf16x2(v[6], v[7]);
          if (STM == 1) {     // statistics of the fp32 values (before the bf16 rounding of the store), two lanes of packed fp32 math
            t1v[nt][k][0] += (f32x2){v[0], v[1]}; t1v[nt][k][0] += (f32x2){v[2], v[3]};
            t2v[nt][k][0] += (f32x2){v[0], v[1]} * (f32x2){v[0], v[1]}; t2v[nt][k][0] += (f32x2){v[2], v[3]} * (f32x2){v[2], v[3]};
            t1v[nt][k][1] += (f32x2){v[4], v[5]}; t1v[nt][k][1] += (f32x2){v[6], v[7]};
            t2v[nt][k][1] += (f32x2){v[4], v[5]} * (f32x2){v[4], v[5]}; t2v[nt][k][1] += (f32x2){v[6], v[7]} * (f32x2){v[6], v[7]};
          }
          // lanes l and l + 32 hold the same pixel: after the swaps lanes < 32 own channels 16k .. 16k+7 and lanes >= 32
          // own 16k+8 .. 16k+15 of their n-tile -> one 16-byte store each
          {
            auto r = __builtin_amdgcn_permlane32_swap(w0x, w1x, false, false);
            w0x = r[0]; w1x = r[1];
            r = __builtin_amdgcn_permlane32_swap(w0y, w1y, false, false);
            w0y = r[0]; w1y = r[1];
          }
          const uint4 o = make_uint4(w0x, w0y, w1x, w1y);
          if (STM == 2) wst[STM == 2 ? mt : 0][nt][k] = o;        // turned into dz and stored by sweep 2
          else c3_store(yb + mt * rsy + coff + lane_y, o);
    };
    // Order of the 16 (pixel row, channel vector) items.  Row-major is what the memory system wants (above); without a residual
    // it costs 11 - 20 spilled registers in this region and measures 3 - 5 % slower, so those variants keep vector-major order.
    constexpr bool ROW_MAJOR = RES || STM == 2;
    if (ROW_MAJOR) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int k = 0; k < 2; ++k) emit(mt, nt, k);
    } else {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) emit(mt, nt, k);
    }
    if (STM == 1) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k = 0; k < 2; ++k) {     // quad index nt*4 + 2k (+1) = channels nt*32 + 16k + 4h (+8) .. +3
          ss[nt * 4 + 2 * k] = t1v[nt][k][0][0] + t1v[nt][k][0][1]; ss[8 + nt * 4 + 2 * k] = t2v[nt][k][0][0] + t2v[nt][k][0][1];
          ss[nt * 4 + 2 * k + 1] = t1v[nt][k][1][0] + t1v[nt][k][1][1]; ss[8 + nt * 4 + 2 * k + 1] = t2v[nt][k][1][0] + t2v[nt][k][1][1];
        }
    }
    if (first) C3_STAMP(4);
    // ---- sweep 2 (backward statistics), in the STORE layout: this lane's vector (nt, k) = channels nt*32 + 16k + 8h .. +7 of
    // its pixel, from the stored (rounded) dy and the GroupNorm input; the accumulators are dead, so the silu' chains have room.
    // Quad index nt*4 + 2k (+1) = channels nt*32 + 16k + 8h (+4) .. +3.
    if (STM == 2) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int cs = nbase + nt * 32 + 16 * k + 8 * h;
          const float4* cc = (const float4*)(p.st_coef + ((long)img * p.Cout + cs) * 2);
          const float4 ka[4] = {cc[0], cc[1], cc[2], cc[3]};           // (a, b) of the 8 channels
          float t1[2] = {0.f, 0.f}, t2[2] = {0.f, 0.f};
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const uint4 wd = wst[STM == 2 ? mt : 0][nt][k], wx = aux[mt][nt][k];
            const float dy[8] = {bf_lo(wd.x), bf_hi(wd.x), bf_lo(wd.y), bf_hi(wd.y), bf_lo(wd.z), bf_hi(wd.z), bf_lo(wd.w), bf_hi(wd.w)};
            const float xg[8] = {bf_lo(wx.x), bf_hi(wx.x), bf_lo(wx.y), bf_hi(wx.y), bf_lo(wx.z), bf_hi(wx.z), bf_lo(wx.w), bf_hi(wx.w)};
            float dzv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float4 kk = ka[e >> 1];
              const float a = (e & 1) ? kk.z : kk.x, b = (e & 1) ? kk.w : kk.y;
              const float z = a * xg[e] + b;
              dzv[e] = dy[e] * (C3_ABL_EPI_NOSILU ? z : silu_grad_fast(z));
              const float adz = a * dzv[e];
              t1[e >> 2] += adz;
              t2[e >> 2] += adz * xg[e];
            }
            // the tensor this conv leaves in HBM is dz = dy * silu'(z): its consumers (the GroupNorm-backward apply, fused into the
            // next dgrad conv's staging or run as a pass) then need no transcendental at all
            wst[STM == 2 ? mt : 0][nt][k] = pack16<bf16_t>(dzv);          // (in place: stored below, a pixel's 128 bytes together)
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float2 m = *(const float2*)(p.st_mr + ((long)img * 32 + (cs + 4 * q) / cpg) * 2);
            ss[nt * 4 + 2 * k + q] = t1[q];
            ss[8 + nt * 4 + 2 * k + q] = (t2[q] - m.x * t1[q]) * m.y;      // sum a*dz*xhat over this lane's values
          }
        }
      }
    }
    if (STM == 2) {
      // stores of the whole tile, pixel row by pixel row: the four 16-byte vectors of a pixel (channels 0..63 of this wave = one
      // 128-byte run) back to back, so L2 merges them into whole lines.  Issued from inside the (nt, k) loop above, the pieces of
      // a line were written a long stretch of VALU work apart and reached HBM as partial lines (WRITE_SIZE = 2 x the tensor).
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int k = 0; k < 2; ++k)
            c3_store(yb + mt * rsy + (unsigned)((nt * 32 + 16 * k) * 2) + lane_y, wst[STM == 2 ? mt : 0][nt][k]);
    }
    if (first) C3_STAMP(5);
    if (STM) {
      // 16 partial sums per lane, 32 pixel lanes per half-wave: butterfly reduce-scatter (8 + 4 + 2 + 1 exchanges, then one
      // plain exchange) leaves value j = bits (4,3,2,1) of the lane index, summed over the half-wave, in every lane
#pragma unroll
      for (int st_ = 0; st_ < 4; ++st_) {
        const int off = 16 >> st_, n = 8 >> st_;            // partner distance, values kept
        const bool up = (pl & off) != 0;
#pragma unroll
        for (int j = 0; j < n; ++j) {
          // (opaque copies: hipcc otherwise folds `up ? ss[j] : ss[j+n]` into a lane-indexed access of the register array,
          // i.e. a 16-way compare / select chain per access -- 1600 instructions, 6 us per tile)
          float lo = ss[j], hi = ss[j + n];
          asm volatile("" : "+v"(lo), "+v"(hi));
          const float send = up ? lo : hi;
          const float keep = up ? hi : lo;
          ss[j] = keep + __shfl_xor(send, off, 64);
        }
      }
      ss[0] += __shfl_xor(ss[0], 1, 64);
      float* sred = (float*)(smem + C3_LDS + 64);           // [wave][h][16], own LDS region (the patch buffers hold the next tile)
      if ((pl & 1) == 0) sred[(wave * 2 + h) * 16 + (pl >> 1)] = ss[0];
      if (first) C3_STAMP(7);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // LDS only: a __syncthreads() here would also wait for the output stores
      __builtin_amdgcn_s_barrier();
      if (tid < 64) {
        // tid -> (wn', h', j): j < 8: sum 1 of quad j = nt * 4 + q, j >= 8: sum 2
        const int wn2 = tid >> 5, h2 = (tid >> 4) & 1, j = tid & 15;
        const float a = sred[((0 * 2 + wn2) * 2 + h2) * 16 + j] + sred[((1 * 2 + wn2) * 2 + h2) * 16 + j];
        const int i = j & 7;          // quad (nt = i >> 2, 2k + q = i & 3): accumulator layout (mode 1) or store layout (mode 2)
        const int ch = nb * C3_BN + wn2 * 64 + (i >> 2) * 32 + (STM == 2 ? ((i >> 1) & 1) * 16 + 8 * h2 + (i & 1) * 4 : (i & 3) * 8 + 4 * h2);
#if C3_ABL_NOATOM
        if (a == 12345.678f) p.st_sums[0] = a;
#else
        atomicAdd(p.st_sums + ((long)img * 32 + ch / cpg) * 2 + (j >> 3), (double)a);
#endif
      }
      // (sred is rewritten by the next tile's epilogue only after a full K loop of stage barriers)
    }
    }
    if (first) C3_STAMP(3);
    first = false;
    if (!has_next) break;
    cur = nxt;
  }
  // drain the redundant tail loads / DMAs before the wave ends
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  C3_STAMP(6);
#if C3_TIMING
  if (p.dbg && threadIdx.x == 0 && STM == 0) p.dbg[(long)blockIdx.x * 8 + 5] = __builtin_readcyclecounter();      // (slot 5: shader clock at block end)
#endif
}

template <int TF, int STM, bool RES>
int launch3(const Conv3Params& p, hipStream_t st) {
  auto kern = conv3_kernel<TF, STM, RES>;
  static std::atomic<unsigned long long> granted{0};     // dynamic-LDS cap raised once per (instantiation, device)
  int dev = 0;
  KDIP_HIP_CHECK(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(granted.load(std::memory_order_acquire) & bit)) {
    KDIP_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS_TOTAL));
    granted.fetch_or(bit, std::memory_order_release);
  }
  // persistent launch: two resident blocks per CU (VGPR / LDS budget of the kernel), a multiple of 8 (one share per XCD)
  static std::atomic<int> num_cu{0};
  if (!num_cu.load()) {
    int n = 0;
    KDIP_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    num_cu.store(n > 0 ? n : 256);
  }
  const long ntiles8 = ((long)p.mtiles * p.nblkN + 7) / 8 * 8;
  long grid = (long)C3_BLOCKS_PER_CU * stream_cus(st, num_cu.load()) / 8 * 8;
  if (grid > ntiles8) grid = ntiles8;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C3_LDS_TOTAL, st, p);
  return KDIP_OK;
}

}  // namespace

int conv3_debug_timing(void* buf) {
  if (!C3_TIMING) return set_error(KDIP_ERR_UNSUPPORTED, "conv3 timing: library not built with -DC3_TIMING=1");
  g_c3_dbg = (unsigned long long*)buf;
  return KDIP_OK;
}

int conv3_tf_max_cin(int tf) { return c3_max_cin(tf); }

bool conv3_eligible(DType dt, int ntaps, int H, int W, int Cin_pad, int Cout, long ldx, long ldy) {
  return dt == DT_BF16 && ntaps == 9 && H % C3_TH == 0 && W % C3_TW == 0 && Cin_pad % 32 == 0 && Cout % C3_BN == 0 &&
         ldx % 8 == 0 && ldy % 8 == 0;
}

int conv3_forward(hipStream_t st, const void* x, long ldx, int B, int H, int W, int Cin, const void* wp, const float* bias, int Cout,
                  void* y, long ldy, const void* res, long ldr, const Conv3Fuse* fu, int cin_real) {
  KDIP_REQUIRE(conv3_eligible(DT_BF16, 9, H, W, Cin, Cout, ldx, ldy), "conv3: shape not eligible (H=%d W=%d Cin=%d Cout=%d)", H, W, Cin, Cout);
  KDIP_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)wp % 16) == 0 && ((uintptr_t)y % 16) == 0, "conv3: pointers must be 16-byte aligned");
  KDIP_REQUIRE(!res || (ldr % 8 == 0 && (uintptr_t)res % 16 == 0), "conv3: residual must be 16-byte aligned");
  Conv3Params p{};
  p.x = (const bf16_t*)x; p.ldx = ldx; p.wp = (const uint4*)wp; p.bias = bias; p.res = (const bf16_t*)res; p.ldr = ldr;
  p.y = (bf16_t*)y; p.ldy = ldy; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.ntilesN = Cout / 32; p.tilesX = W / C3_TW; p.tilesY = H / C3_TH; p.mtiles = B * p.tilesX * p.tilesY; p.nblkN = Cout / C3_BN;
  int tf = 0, stm = 0;
  if (fu) {
    p.in_ups = fu->in_ups; p.res_ups = fu->res_ups;
    tf = fu->tf; p.tf_coef = fu->tf_coef; p.tf_silu = fu->tf_silu; p.x2 = (const bf16_t*)fu->x2; p.ldx2 = fu->ldx2;
    stm = fu->st_mode; p.st_silu = fu->st_silu; p.st_sums = fu->st_sums; p.st_x = (const bf16_t*)fu->st_x; p.st_ldx = fu->st_ldx;
    p.st_coef = fu->st_coef; p.st_mr = fu->st_mr;
  }
  if (fu) {
    p.fold_stats = fu->fold_stats; p.fold_stats2 = fu->fold_stats2; p.fold_C1 = fu->fold_C1; p.fold_gamma = fu->fold_gamma; p.fold_beta = fu->fold_beta;
    p.fold_film = fu->fold_film; p.fold_film_ld = fu->fold_film_ld ? fu->fold_film_ld : 2L * Cin; p.fold_HW = fu->fold_HW; p.fold_eps = fu->fold_eps;
    p.fold_coef_out = fu->fold_coef_out; p.fold_mr_out = fu->fold_mr_out; p.fold_coef = fu->fold_coef; p.fold_mr = fu->fold_mr;
  }
  KDIP_REQUIRE(!(p.in_ups || p.res_ups) || (H % 2 == 0 && W % 2 == 0), "conv3: fused x2 upsample needs even H, W");
p.dbg = C3_TIMING ? g_c3_dbg : nullptr;
  KDIP_REQUIRE(tf >= 0 && tf <= 2 && stm >= 0 && stm <= 2, "conv3: bad fusion modes");
  // every GroupNorm in front of / behind a 3x3 conv of the UNet is followed by SiLU: the activation is compiled in (a
  // run-time switch doubled the epilogue's register footprint: 170 spilled VGPRs)
  KDIP_REQUIRE((tf == 0 || p.tf_silu) && (stm != 2 || p.st_silu), "conv3: the fused GroupNorm transforms include SiLU");
  KDIP_REQUIRE(tf == 0 || p.fold_stats || (p.tf_coef && ((uintptr_t)p.tf_coef % 16) == 0), "conv3: transform coefficients must be given and 16-byte aligned");
  KDIP_REQUIRE(!p.fold_stats || (tf == 1 && p.fold_gamma && p.fold_beta && p.fold_coef_out && p.fold_mr_out && p.fold_HW > 0 && (uintptr_t)p.fold_coef_out % 16 == 0 &&
                                 (!p.fold_stats2 || (p.fold_C1 > 0 && p.fold_C1 < Cin))) ||
                   (tf == 2 && p.fold_coef && p.fold_mr && p.fold_HW > 0), "conv3: incomplete GroupNorm-coefficient fold");
  KDIP_REQUIRE(tf != 2 || (p.x2 && p.ldx2 == ldx && (uintptr_t)p.x2 % 16 == 0 && !p.in_ups), "conv3: GroupNorm-backward staging needs the GroupNorm input in the layout of x");
  KDIP_REQUIRE(Cin <= c3_max_cin(tf), "conv3: too many input channels (%d) for the staging-transform table", Cin);
  KDIP_REQUIRE((long)H * W * ldx * 2 < (1L << 31), "conv3: image too large for 32-bit staging offsets");
  KDIP_REQUIRE(stm == 0 || ((Cout >> 5) % 4 == 0 && p.st_sums), "conv3: fused statistics need Cout / 32 to be a multiple of 4");
  KDIP_REQUIRE(stm != 2 || (p.st_x && p.st_ldx % 8 == 0 && (uintptr_t)p.st_x % 16 == 0 && (uintptr_t)p.st_coef % 16 == 0 && (uintptr_t)p.st_mr % 8 == 0),
               "conv3: backward statistics need the GroupNorm input / coefficients (aligned)");
  // (all launch preconditions are checked before prof_begin: an early return must not leave an unmatched event pair)
  // instantiated combinations: forward convs (tf 0 / 1, statistics 0 / 1, with / without residual) and dgrad convs
  // (tf 0 / 2, statistics 0 / 2, never a residual)
  KDIP_REQUIRE(!(res && (tf == 2 || stm == 2)), "conv3: residual together with GroupNorm-backward fusion is not instantiated");
  KDIP_REQUIRE(!(tf == 1 && stm == 2) && !(tf == 2 && stm == 1), "conv3: fusion mode combination is not instantiated");
  if (g_prof_on) {
    const double px = (double)B * H * W;
    const int cr = cin_real > 0 ? cin_real : Cin;
    // tag = conv3[_gnf|_gnb][_s1|_s2][_res]; algorithmic bytes = every tensor the launch must read / write once
    static const char* tags[3][3][2] = {{{"conv3", "conv3_res"}, {"conv3_s1", "conv3_s1_res"}, {"conv3_s2", "conv3_s2_res"}},
                                        {{"conv3_gnf", "conv3_gnf_res"}, {"conv3_gnf_s1", "conv3_gnf_s1_res"}, {"conv3_gnf_s2", "conv3_gnf_s2_res"}},
                                        {{"conv3_gnb", "conv3_gnb_res"}, {"conv3_gnb_s1", "conv3_gnb_s1_res"}, {"conv3_gnb_s2", "conv3_gnb_s2_res"}}};
    const double in_px = p.in_ups ? px / 4 : px, res_px = p.res_ups ? px / 4 : px;
    const double bytes = in_px * cr * 2.0 * (tf == 2 ? 2 : 1) + 9.0 * cr * Cout * 2.0 + px * Cout * 2.0 + (res ? res_px * Cout * 2.0 : 0.0) +
                         (stm == 2 ? px * Cout * 2.0 : 0.0);
    prof_begin(st, PC_CONV3_128x128, 2.0 * px * cr * Cout * 9, bytes, tags[tf][stm][res ? 1 : 0], B, H, cr, Cout);
  }
  int rc;
#define C3_GO(T, S, R) rc = launch3<T, S, R>(p, st)
  if (tf == 0 && stm == 0) { if (res) C3_GO(0, 0, true); else C3_GO(0, 0, false); }
  else if (tf == 0 && stm == 1) { if (res) C3_GO(0, 1, true); else C3_GO(0, 1, false); }
  else if (tf == 1 && stm == 0) { if (res) C3_GO(1, 0, true); else C3_GO(1, 0, false); }
  else if (tf == 1 && stm == 1) { if (res) C3_GO(1, 1, true); else C3_GO(1, 1, false); }
  else if (tf == 0 && stm == 2) C3_GO(0, 2, false);
  else if (tf == 2 && stm == 0) C3_GO(2, 0, false);
  else C3_GO(2, 2, false);
#undef C3_GO
  prof_end(st);
  if (rc) return rc;
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

}  // namespace kdip
