// bf16 implicit-GEMM 3x3 convolution for the large feature maps of the UNet, third generation: the conv3.hip data path (cout-row
// MFMA orientation with a register epilogue, conflict-free 32-wide halo patches, weights streamed into an LDS ring by
// global_load_lds, GroupNorm forward / backward apply fused into the patch staging, GroupNorm statistics fused into the
// epilogue) on a PING-PONG schedule.  Replaces nn.Conv2d(3x3, pad 1) of guided_diffusion/unet.py:182-222 and its
// input-gradient, with the GroupNorm + FiLM + SiLU in front of it (unet.py:183-184,207-208,249-253) / the GroupNorm backward
// behind its dgrad applied while the input patch is staged.
//
// Why a third kernel: conv3 runs two independent 4-wave blocks per CU.  The two waves that share a SIMD execute the same
// stage structure in no particular phase relation, each stage of 16 MFMAs has its own weight DMAs, staging, waits and barrier in
// front of the MFMAs, and the measured result is that the second block adds < 10 %: both waves of a SIMD
// spend their non-MFMA issue time at moments the other one cannot cover.  Here ONE 8-wave block per CU owns a 16 x 32-pixel x
// 128-channel tile, and its two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run the same phase sequence ONE
// BARRIER APART: a phase is
//     load part : 12 ds_read_b128 (patch + weight fragments of this stage), 1 weight DMA (stage + 2), patch staging
//     s_barrier
//     MFMA part : 16 x v_mfma_f32_32x32x16_bf16 at raised priority, nothing else
//     s_barrier
// so while group 0 is in its MFMA part, group 1 is in its load part and vice versa: each SIMD's matrix pipe always has exactly
// one wave issuing back-to-back MFMAs, and all LDS / VMEM / VALU issue of the other wave sits beside them.  The block shares
// one weight ring (half the weight DMAs per MFMA of conv3) and one 18 x 34 halo patch (10 % fewer staged pixels per output).
//
//   * GEMM view per block: D[cout 128][pixel 512] += W[cout][k] * X[k][pixel], k = (tap, cin); wave tile = 4 pixel rows
//     (4 MFMA column tiles) x 64 cout = acc[4][2], waves = 4 (pixel rows) x 2 (cout halves), group = wave >> 2.
//   * stage = (tap, 32-channel chunk) = 2 k-steps = 16 MFMAs per wave; weights of a stage = 8 KiB = one 1 KiB DMA per wave,
//     issued two phases ahead into a 4-slot ring; counted s_waitcnt vmcnt(N) one phase ahead of the first read, never 0
//     inside a tile.
//   * patch of the next chunk: vector i (16 bytes of one halo pixel) is requested at the end of phase i's load part and
//     transformed + written to the other patch buffer in phase i + 2.
//   * the groups re-synchronise for the epilogue (one extra barrier each per tile) so that both store at the same time.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "conv3p.h"

namespace kdip {

namespace {

constexpr int C4_TH = 16, C4_TW = 32, C4_BN = 128, C4_NTHR = 512;
constexpr int C4_PW = C4_TW + 2, C4_PH = C4_TH + 2, C4_NPIX = C4_PW * C4_PH;   // 34 x 18 = 612 halo pixels
constexpr int C4_PIXB = 80;                               // LDS pixel pitch: 64 B of channels + 16 B pad (conflict-free ds_read_b128 of 32-pixel rows)
constexpr int C4_ABUF = C4_NPIX * C4_PIXB;                // 48960 B per patch buffer
constexpr int C4_BSLOT = 8192;                            // one weight stage: 2 k-steps x 4 n-tiles x 1 KiB
#ifndef C4_DDIST
#define C4_DDIST 3             // weight DMA distance in phases: stage s + DDIST is requested in phase s
#endif
constexpr int C4_NSLOT = C4_DDIST + 2;                    // weight ring: the slot refilled in phase s was last read in phase s - 2
constexpr int C4_MAXV = (C4_NPIX * 4 + C4_NTHR - 1) / C4_NTHR;   // staged 16-byte vectors per thread (5)
constexpr int C4_LDS = 2 * C4_ABUF + C4_NSLOT * C4_BSLOT; // 130688 B: one block per CU
constexpr int C4_MAXCOUT = 1024;                          // bias table in LDS (the accumulator init must not queue behind the previous tile's stores)
#ifndef C4_TIMING
#define C4_TIMING 0            // diagnostic build: s_memtime stamps of the first tile's phases (4 per phase) -> buffer set by conv4_debug_timing
#endif
constexpr int C4_NSTAMP = 176;
constexpr int C4_LDS_TOTAL = C4_LDS + 64 + 1024 + 4 * C4_MAXCOUT + (C4_TIMING ? 8 * C4_NSTAMP * 8 : 0);   // + dummy staging slot + statistics exchange [8 waves][2][16] floats + bias [Cout]
// per-channel coefficients of the staging transform ride in the 16-byte pads of the patch pixels (as in conv3): slot j = pad of
// pixel j % 608 of buffer j / 608.  TF 1: slot = 2 channels x (a, b); TF 2: slot = 1 channel x (a, b, k0, k1), 608 = 19 x 32 so
// a 32-channel chunk never straddles the buffers; within a chunk channel 8g + e sits in slot 4e + g (conflict-free, conv3.hip).
constexpr int C4_TABPIX = 608;
constexpr int c4_max_cin(int tf) { return tf == 1 ? 4 * C4_TABPIX : (tf == 2 ? 2 * C4_TABPIX : (1 << 20)); }
__device__ __forceinline__ int c4_tab_off(int slot) { return (slot / C4_TABPIX) * C4_ABUF + (slot % C4_TABPIX) * C4_PIXB + 64; }

#ifndef C4_PRIO
#define C4_PRIO 1              // s_setprio around the MFMA part
#endif
#ifndef C4_STAGGER
#define C4_STAGGER 1           // 0: both wave groups in lock step (A/B build)
#endif
#ifndef C4_ABL_NOWAIT
#define C4_ABL_NOWAIT 0        // timing ablations (results are garbage): no counted vmcnt waits / no weight DMA / no patch staging / no fragment
#endif                         // reads / no epilogue / no MFMAs
#ifndef C4_ABL_NODMA
#define C4_ABL_NODMA 0
#endif
#ifndef C4_ABL_NOSTG
#define C4_ABL_NOSTG 0
#endif
#ifndef C4_ABL_NOLDSR
#define C4_ABL_NOLDSR 0
#endif
#ifndef C4_ABL_NOEPI
#define C4_ABL_NOEPI 0
#endif
#ifndef C4_ABL_NOATOM
#define C4_ABL_NOATOM 0        // epilogue ablations: no statistics atomics / no silu' transcendentals / no second input tensor / no output stores
#endif
#ifndef C4_ABL_NOSILU
#define C4_ABL_NOSILU 0
#endif
#ifndef C4_ABL_NOAUX
#define C4_ABL_NOAUX 0
#endif
#ifndef C4_ABL_NOSTORE
#define C4_ABL_NOSTORE 0
#endif
#ifndef C4_ABL_NOMFMA
#define C4_ABL_NOMFMA 0
#endif
#ifndef C4_SLEEP
#define C4_SLEEP 0             // experiment: block j starts (j / 8 % 16) * C4_SLEEP * 64 cycles late (de-synchronises the chip-wide epilogue bursts)
#endif
#ifndef C4_TFINM
#define C4_TFINM 1             // staging transform + LDS write between the MFMAs instead of in the load part (TF 1 / 2)
#endif
#ifndef C4_TFM_VALU
#define C4_TFM_VALU 5
#endif
#ifndef C4_AUX_PREFETCH
#define C4_AUX_PREFETCH 0     // measured: 197 -> 209 us (STM 2), 212 -> 222 us (RES): the touches queue HBM-latency loads in front of the counted weight DMAs
#endif
#ifndef C4_ONEBAR
#define C4_ONEBAR 1            // one barrier per phase: group 0 runs (MFMA part, next load part) between barriers, group 1 (load part, MFMA part):
#endif                         // the MFMA clusters of a SIMD's two waves alternate without a barrier hand-off between them
#ifndef C4_CUNROLL
#define C4_CUNROLL 2           // chunks per iteration of the chunk loop
#endif
#ifndef C4_MIN_TILES
#define C4_MIN_TILES 512       // launches with fewer than two 512-pixel tiles per CU stay on conv3's 256-pixel tiles (chip fill, first-tile prologue)
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma_bf16(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// TF: staging transform (0 none, 1 GroupNorm forward apply, 2 GroupNorm backward apply); STM: statistics mode of the output
template <int TF, int STM, bool RES>
__global__ __launch_bounds__(C4_NTHR, 2) void conv4_kernel(Conv3Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = C4_STAGGER ? wave >> 2 : 0;            // waves w and w + 4 sit on the same SIMD: one of each group per SIMD
  const int wm = wave >> 1, wn = wave & 1;
  // persistent blocks, one per CU: XCD k (= blockIdx % 8) walks a contiguous tile range (neighbouring halos and the n-blocks
  // of an m-tile share one L2)
  const int ntiles = p.mtiles * p.nblkN;
  const int xq = ntiles >> 3, xr = ntiles & 7, xcd = blockIdx.x & 7;
  const int xstart = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq, xend = xstart + xq + (xcd < xr ? 1 : 0);
  const int nper = gridDim.x >> 3;
  int cur = xstart + (blockIdx.x >> 3);
  if (cur >= xend) return;
  if (C4_SLEEP) { for (int i = (int)((blockIdx.x >> 3) & 15); i > 0; --i) __builtin_amdgcn_s_sleep(C4_SLEEP); }
  const int tpi = p.tilesX * p.tilesY;
  const int Hs = p.in_ups ? p.H >> 1 : p.H, Ws = p.in_ups ? p.W >> 1 : p.W;
  const int nchunks = p.Cin >> 5;
  const long kStride = (long)p.ntilesN * 64, tapStride = (long)(p.Cin >> 4) * kStride;     // packed weights, in 16-byte units

  // ---- staging descriptors of the tile whose chunks are being STAGED (the current tile or, during its last chunk, the next
  // one): vector v = tid + 512 i -> (halo pixel v >> 2, 16-byte channel group tid & 3); byte offsets relative to the image
  unsigned goff[C4_MAXV];
  unsigned zmask = 0;                  // bit i: vector i is zero fill (conv padding); bit 8 + i: vector i is outside the patch
  const bf16_t* ximg = p.x;
  const bf16_t* x2img = p.x;
  int tab_img = -1;
  // tile-invariant part: halo coordinates of this thread's vectors (hy << 8 | hx), computed once; vectors outside the patch
  // (pixel >= 612, only possible for the last vector) are flagged in zmask bits 8.. for the whole kernel
  unsigned hyx[C4_MAXV];
  unsigned zout = 0;
#pragma unroll
  for (int i = 0; i < C4_MAXV; ++i) {
    const int pix = (tid + i * C4_NTHR) >> 2;
    const int hy = pix / C4_PW, hx = pix - hy * C4_PW;
    hyx[i] = (unsigned)(hy << 8 | hx);
    if (pix >= C4_NPIX) zout |= 0x101u << i;
  }
  auto fdiv = [](unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; };      // n / d with magic = ceil(2^32 / d) (host), exact for n * d < 2^32
  auto set_staging = [&](int bid) {                       // branch free, no divisions: it runs inside a load part
    const int mtile = (int)fdiv((unsigned)bid, p.mg_nblk);
    const int im = (int)fdiv((unsigned)mtile, p.mg_tpi), trem = mtile - im * tpi;
    const int ty = (int)fdiv((unsigned)trem, p.mg_tx);
    const int yy = ty * C4_TH - 1, xx = (trem - ty * p.tilesX) * C4_TW - 1;
    ximg = p.x + (long)im * Hs * Ws * p.ldx;
    if (TF == 2) x2img = p.x2 + (long)im * Hs * Ws * p.ldx;
    zmask = zout;
#pragma unroll
    for (int i = 0; i < C4_MAXV; ++i) {
      const int gy = yy + (int)(hyx[i] >> 8), gx = xx + (int)(hyx[i] & 255);
      const bool inimg = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && !((zout >> i) & 1);
      const int spix = p.in_ups ? (gy >> 1) * Ws + (gx >> 1) : gy * Ws + gx;
      goff[i] = inimg ? (unsigned)((spix * (int)p.ldx + (tid & 3) * 8) * 2) : (unsigned)((tid & 3) * 16);
      zmask |= (inimg ? 0u : 1u) << i;
    }
    return im;
  };
  auto load_table = [&](int im) {
    if (TF == 1) {
      if (p.fold_stats) {      // coefficients from the statistics (Conv3Fuse::fold_*, as conv3.hip)
        const int cpg = p.Cin >> 5;
        for (int j = tid; j < p.Cin / 2; j += C4_NTHR) {
          float4 k; float m0, r0, m1, r1;
          c3_fold_coef_fwd(p, im, 2 * j, k.x, k.y, m0, r0);
          c3_fold_coef_fwd(p, im, 2 * j + 1, k.z, k.w, m1, r1);
          *(float4*)(smem + c4_tab_off(j)) = k;
          ((float4*)(p.fold_coef_out + (long)im * p.Cin * 2))[j] = k;
          if ((2 * j) % cpg == 0) *(float2*)(p.fold_mr_out + ((long)im * 32 + (2 * j) / cpg) * 2) = make_float2(m0, r0);
          if ((2 * j + 1) % cpg == 0) *(float2*)(p.fold_mr_out + ((long)im * 32 + (2 * j + 1) / cpg) * 2) = make_float2(m1, r1);
        }
      } else {
        const float4* src = (const float4*)(p.tf_coef + (long)im * p.Cin * 2);
        for (int j = tid; j < p.Cin / 2; j += C4_NTHR) *(float4*)(smem + c4_tab_off(j)) = src[j];
      }
    } else if (TF == 2) {
      const float4* src = (const float4*)(p.tf_coef + (long)im * p.Cin * 4);
      for (int j = tid; j < p.Cin; j += C4_NTHR)
        *(float4*)(smem + c4_tab_off((j & ~31) + (j & 7) * 4 + ((j >> 3) & 3))) = p.fold_stats ? c3_fold_coef_bwd(p, im, j) : src[j];
    }
    tab_img = im;
  };

  // ---- staging loads are inline asm: hipcc must neither wait for them itself (next to LDS-DMA traffic it drains the whole
  // queue) nor touch their destination registers before the counted wait of the phase that consumes them
  constexpr int NV = TF == 2 ? 2 : 1;                     // staging loads per vector
  // The epilogue reads a second tensor (residual / GroupNorm input of the produced gradient): 64 KB per tile whose HBM latency --
  // of every CU at the same moment -- would sit exposed behind the last MFMA (measured: 28 - 36 us of a 205 - 215 us launch).  Taps
  // 5 .. 8 of every chunk (no staging loads there) touch the four 128-byte lines of this thread's tile pixels, so the epilogue's
  // loads are served by L2.  Results are discarded; the loads are counted like the others.
  constexpr bool AUXPF = C4_AUX_PREFETCH && (RES || STM == 2);
  unsigned pf = 0;
  u32x4 sa[2], sb[2];
  auto vec_load_asm = [&](int cbytes, int i) {          // chunk byte offset (chunk * 64), vector i -> register set i & 1
    const unsigned off = goff[i] + (unsigned)cbytes;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sa[i & 1]) : "v"(off), "s"(ximg) : "memory");
    if (TF == 2) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sb[i & 1]) : "v"(off), "s"(x2img) : "memory");
  };
  auto transform = [&](int c, int i, uint4 o, uint4 o2) -> uint4 {
    if (TF == 1) {
      const int slot = (c * 32 + (tid & 3) * 8) >> 1;                       // 4 consecutive slots: (a, b) of 2 channels each
      const unsigned char* tab = smem + c4_tab_off(slot);
      float f[8];
      unpack16<bf16_t>(o, f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 k = *(const float4*)(tab + j * C4_PIXB);
        f[2 * j] = silu_fast(k.x * f[2 * j] + k.y);
        f[2 * j + 1] = silu_fast(k.z * f[2 * j + 1] + k.w);
      }
      o = pack16<bf16_t>(f);
    } else if (TF == 2) {
      const unsigned char* tab = smem + c4_tab_off(c * 32 + (tid & 3));      // channel 8 (tid & 3) + e of the chunk: slot 4e + (tid & 3)
      float fd[8], fx[8];
      unpack16<bf16_t>(o, fd);
      unpack16<bf16_t>(o2, fx);
#pragma unroll
      for (int e = 0; e < 8; ++e) {          // the staged tensor already holds dz = dy * silu'(z) (written by the producing epilogue)
        const float4 k = *(const float4*)(tab + e * 4 * C4_PIXB);
        fd[e] = k.x * fd[e] - (k.z + k.w * fx[e]);
      }
      o = pack16<bf16_t>(fd);
    }
    if ((zmask >> i) & 1) o = make_uint4(0, 0, 0, 0);      // conv zero padding applies to the TRANSFORMED tensor
    return o;
  };
  auto vec_store = [&](int buf, int i, uint4 o) {    // branch free: lanes outside the patch (vector 4, pixel >= 612) write a dummy slot
    const int pix = (tid + i * C4_NTHR) >> 2;
    int addr = buf * C4_ABUF + pix * C4_PIXB + (tid & 3) * 16;
    if (i * C4_NTHR + C4_NTHR - 1 >= C4_NPIX * 4) addr = ((zmask >> (8 + i)) & 1) ? C4_LDS + (tid & 3) * 16 : addr;
    *(uint4*)(smem + addr) = o;
  };

  // ---- weight stream: stage (chunk c, tap t) = k-steps 2c, 2c+1 of tap t, n-tiles 4 nb .. 4 nb + 3 = 8 x 1 KiB; wave w
  // copies piece w (k-step w >> 2, n-tile w & 3) with ONE global_load_lds_dwordx4 (no VGPR round trip).  Inline asm: next to a
  // builtin LDS-DMA hipcc drains vmcnt(0) in front of the next ds_read of the same LDS object.
  const unsigned w_voff = (unsigned)(lane * 16);
  const unsigned lds_w = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + 2 * C4_ABUF + wave * 1024;
  auto dma_w = [&](int nb_, int c, int tap, int slot) {
    const uint4* g = p.wp + (long)nb_ * 256 + tap * tapStride + (long)(2 * c + (wave >> 2)) * kStride + (wave & 3) * 64;
    const unsigned l = lds_w + slot * C4_BSLOT;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(w_voff), "s"(g), "s"(l) : "memory");
  };

  auto ring = [](int x) {      // x % NSLOT for x < 4 NSLOT, wave-uniform (kept in SGPRs: it feeds M0)
    return __builtin_amdgcn_readfirstlane(C4_NSLOT == 4 ? (x & 3) : x - C4_NSLOT * ((x >= C4_NSLOT ? 1 : 0) + (x >= 2 * C4_NSLOT ? 1 : 0) + (x >= 3 * C4_NSLOT ? 1 : 0)));
  };
  const int a_lane = (wm * 4 * C4_PW + (lane & 31)) * C4_PIXB + (lane >> 5) * 16;   // + (mt + ty) * PW * PIXB + tx * PIXB + ks * 32
  const int b_lane = 2 * C4_ABUF + (wn * 2 * 64 + lane) * 16;                       // + slot * BSLOT + (ks * 4 + nt) * 1024
  const int cpg = p.Cout >> 5;
  float* const sbias = (float*)(smem + C4_LDS + 64 + 1024);
#if C4_TIMING
  unsigned long long* const tl = (unsigned long long*)(smem + C4_LDS + 64 + 1024 + 4 * C4_MAXCOUT) + wave * C4_NSTAMP;
  bool tfirst = true;
  int tcount = 0;
  if (lane == 0) { tl[170] = __builtin_readcyclecounter(); tl[171] = __builtin_amdgcn_s_memrealtime(); }
#define C4_STAMP(k) do { if (C4_TIMING >= 2 && tfirst && c * 9 + tap < 36) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) tl[(c * 9 + tap) * 4 + (k)] = t_; } } while (0)
#else
#define C4_STAMP(k) do { } while (0)
#endif

  // ---- prologue of the block's FIRST tile: weight stages 0 and 1, coefficient table, the whole first patch
  {
    const int nb0 = cur - (int)fdiv((unsigned)cur, p.mg_nblk) * p.nblkN;
#pragma unroll
    for (int t = 0; t < C4_DDIST; ++t) dma_w(nb0, 0, t, t);
    const int im = set_staging(cur);
    // every global load of the prologue is requested before the first wait (one HBM round trip instead of three)
    uint4 o[C4_MAXV], o2[C4_MAXV];
#pragma unroll
    for (int i = 0; i < C4_MAXV; ++i) {
      o[i] = *(const uint4*)((const char*)ximg + goff[i]);
      o2[i] = make_uint4(0, 0, 0, 0);
      if (TF == 2) o2[i] = *(const uint4*)((const char*)x2img + goff[i]);
    }
    if (TF) load_table(im);
    for (int j = tid; j < p.Cout; j += C4_NTHR) sbias[j] = p.bias ? p.bias[j] : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // coefficient table visible to every wave
#pragma unroll
    for (int i = 0; i < C4_MAXV; ++i) vec_store(0, i, transform(0, i, o[i], o2[i]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  int pb = 0;                                            // patch buffer that holds chunk 0 of the current tile
  int sidx = 0;                                          // ring slot of the current chunk's tap 0 (stage counter mod 4)

  for (;;) {
    // ---- current tile
    const int mtile = (int)fdiv((unsigned)cur, p.mg_nblk), nb = cur - mtile * p.nblkN;
    const int img = (int)fdiv((unsigned)mtile, p.mg_tpi), trem = mtile - img * tpi;
    const int ty0 = (int)fdiv((unsigned)trem, p.mg_tx);
    const int y0 = ty0 * C4_TH, x0 = (trem - ty0 * p.tilesX) * C4_TW;
    const int nxt = cur + nper;
    const bool has_next = nxt < xend;
    const int nb_n = has_next ? nxt - (int)fdiv((unsigned)nxt, p.mg_nblk) * p.nblkN : nb;

    // accumulators start at the bias (lane (pixel, h) owns channels nt*32 + 8q + 4h + j in registers 4q + j)
    int tv = threadIdx.x;
    asm volatile("" : "+v"(tv));        // (opaque copy: lane-derived values must not be hoisted out of the tile loop and kept live)
#pragma unroll
    for (int j = 0; j < 2; ++j) { sa[j] = (u32x4){0, 0, 0, 0}; sb[j] = (u32x4){0, 0, 0, 0}; }
    const bf16_t* aux_img = p.y;      // wave-uniform row base of the tensor the epilogue reads + this thread's pixel offset, row stride
    unsigned aux_off = 0, aux_rs = 0;
    if (AUXPF) {
      const int px = x0 + (tv & 31), r0 = y0 + wm * 4;
      if (RES) {
        const int Hr = p.res_ups ? p.H >> 1 : p.H, Wr = p.res_ups ? p.W >> 1 : p.W;
        aux_img = p.res + ((long)img * Hr + (p.res_ups ? r0 >> 1 : r0)) * Wr * p.ldr + nb * C4_BN + wn * 64;
        aux_off = (unsigned)((p.res_ups ? px >> 1 : px) * (int)p.ldr * 2);
        aux_rs = (unsigned)(Wr * (int)p.ldr * 2);      // (res_ups: rows r0 .. r0 + 3 are two source rows: lines mt >> 1)
      } else {
        aux_img = p.st_x + ((long)img * p.H + r0) * p.W * p.st_ldx + nb * C4_BN + wn * 64;
        aux_off = (unsigned)(px * (int)p.st_ldx * 2);
        aux_rs = (unsigned)(p.W * (int)p.st_ldx * 2);
      }
    }
    pf = 0;
    f32x16 acc[4][2];
    {
      const int h = (tv >> 5) & 1;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bq = *(const float4*)(sbias + nb * C4_BN + wn * 64 + nt * 32 + 8 * q + 4 * h);      // (LDS: no VMEM at a tile start)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            acc[mt][nt][4 * q + 0] = bq.x; acc[mt][nt][4 * q + 1] = bq.y; acc[mt][nt][4 * q + 2] = bq.z; acc[mt][nt][4 * q + 3] = bq.w;
          }
        }
    }
    // from here to the epilogue every VMEM operation is counted by hand
    if (!C4_ONEBAR && grp) __builtin_amdgcn_s_barrier();               // group 1 runs one barrier behind group 0 until the end of the K loop

    // Per phase (chunk c, tap t) every wave issues, in this order: 12 fragment reads, the weight DMA of stage s + 2, the counted
    // wait, the transform + LDS write of staging vector t - 2, the staging load(s) of vector t.  The wait must cover the DMA of
    // stage s + 1 (issued in phase s - 1; read in phase s + 1 after two more barriers) and vector t - 2 (issued at the end of
    // phase s - 2, i.e. older): operations issued after that DMA = staging loads of phase s - 1 + this phase's DMA
    //   -> vmcnt(1 + nv(t - 1)), nv(t) = NV for t < 5 else 0 (taps wrap within the 9-tap chunk).
    // Phase 0 of a tile's first chunk needs nothing new (stage 1 was confirmed in front of the previous epilogue / by the
    // prologue) and does not wait: the previous epilogue's output stores stay in flight under the first phase.
#define C4_NVT(t) (((t) + 9) % 9 < C4_MAXV ? NV : (AUXPF && ((t) + 9) % 9 < C4_MAXV + 4 ? 1 : 0))
#if C4_TIMING
    if (lane == 0 && tcount < 8) tl[145 + 3 * tcount] = __builtin_readcyclecounter();          // K loop start
#endif
    auto chunk = [&](int c) __attribute__((always_inline)) {
      const bool last = c + 1 == nchunks;
      if (last && has_next) {                            // from here on the staging loads belong to the next tile
        const int im = set_staging(nxt);
        if (TF && im != tab_img) {                       // (rare: the block's next tile lies in another image)
          asm volatile("; C4_RARE_BEGIN\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();                  // every wave is past its last transform of the current image
          load_table(im);
          asm volatile("; C4_RARE_END" ::: "memory");
        }
      }
      const unsigned char* ab = smem + ((pb + c) & 1) * C4_ABUF + a_lane;
      const int nbuf = (pb + c + 1) & 1;                 // patch buffer being staged
      const int cn = last ? (has_next ? 0 : c) : c + 1;  // chunk (within its tile) being staged (no next tile: redundant re-loads keep the counts uniform)
      const int wc = cn, wnb = last ? nb_n : nb;         // chunk / n-block of the weight stages that wrap into the next chunk
      auto phase = [&](auto tapc) {
        constexpr int tap = decltype(tapc)::value;
        // ---- load part
        C4_STAMP(0);
        const unsigned char* bb = smem + b_lane + ring(sidx + tap) * C4_BSLOT;
        constexpr int toff = ((tap / 3) * C4_PW + (tap % 3)) * C4_PIXB;
        uint4 xa[2][4], wb[2][2];
        if (C4_ABL_NOLDSR) {
#pragma unroll
          for (int i = 0; i < 8; ++i) (&xa[0][0])[i] = make_uint4(tid + i, tid, i, tap);
#pragma unroll
          for (int i = 0; i < 4; ++i) (&wb[0][0])[i] = make_uint4(tid, i, tap, tid);
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) wb[ks][nt] = *(const uint4*)(bb + (ks * 4 + nt) * 1024);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) xa[ks][mt] = *(const uint4*)(ab + mt * C4_PW * C4_PIXB + toff + ks * 32);
        }
        if (!C4_ABL_NODMA) {
          constexpr int t2 = tap + C4_DDIST >= 9 ? tap + C4_DDIST - 9 : tap + C4_DDIST;
          if (tap + C4_DDIST >= 9) dma_w(wnb, wc, t2, ring(sidx + tap + C4_DDIST)); else dma_w(nb, c, t2, ring(sidx + tap + C4_DDIST));
        }
        // counted wait: the DMA of stage s + 1 (requested in phase s + 1 - DDIST) and, in the phases that write one, staging vector
        // t - 2 (requested at the end of phase s - 2) must have landed; everything requested later may stay in flight
        constexpr bool WR = tap >= 2 && tap < 2 + C4_MAXV;
        constexpr int NW = C4_DDIST == 2 ? 1 + C4_NVT(tap - 1) : (WR ? 2 + C4_NVT(tap - 1) : 2 + C4_NVT(tap - 2) + C4_NVT(tap - 1));
        if (!C4_ABL_NOWAIT && (tap >= C4_DDIST - 1 || c > 0)) {
          if (TF == 2) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(sa[0]), "+v"(sa[1]), "+v"(sb[0]), "+v"(sb[1]), "+v"(pf) : "n"(NW) : "memory");
          else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(sa[0]), "+v"(sa[1]), "+v"(pf) : "n"(NW) : "memory");
        }
        // the staging transform (GroupNorm apply: ~60 VALU per vector for TF 1) would make the load part longer than the partner's
        // MFMA part: with TFM it is issued between this wave's own MFMAs instead (<= 4 VALU per MFMA gap), and the next staging
        // load (same register set) follows the MFMAs
        constexpr bool TFM = C4_TFINM && TF != 0;
        auto stage_write = [&]() {
          if (!C4_ABL_NOSTG && tap >= 2 && tap < 2 + C4_MAXV) {
            constexpr int i = tap >= 2 ? tap - 2 : 0;
            const uint4 o = __builtin_bit_cast(uint4, sa[i & 1]);
            const uint4 o2 = TF == 2 ? __builtin_bit_cast(uint4, sb[i & 1]) : make_uint4(0, 0, 0, 0);
            vec_store(nbuf, i, transform(cn, i, o, o2));
          }
        };
#define C4_AUX_TOUCH() do { if (AUXPF && tap >= C4_MAXV && tap < C4_MAXV + 4) { \
            constexpr int mt_ = tap >= C4_MAXV ? tap - C4_MAXV : 0; \
            const unsigned off_ = aux_off + (unsigned)((RES && p.res_ups) ? mt_ >> 1 : mt_) * aux_rs; \
            asm volatile("global_load_dword %0, %1, %2" : "=v"(pf) : "v"(off_), "s"(aux_img) : "memory"); } } while (0)
        if (!TFM) {
          stage_write();
          if (!C4_ABL_NOSTG && tap < C4_MAXV) vec_load_asm(cn * 64, tap);
          C4_AUX_TOUCH();
        }
        // ---- MFMA part, between two barriers
        C4_STAMP(1);
        __builtin_amdgcn_sched_barrier(0);
        if (!C4_ONEBAR || !grp) __builtin_amdgcn_s_barrier();       // (one barrier per phase: group 0 here, group 1 behind its MFMAs)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        C4_STAMP(2);
        if (C4_PRIO) __builtin_amdgcn_s_setprio(C4_PRIO);
        if (TFM) stage_write();
        if (C4_ABL_NOMFMA) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i & 3][i >> 2][tap] += __uint_as_float((&xa[0][0])[i].x ^ (&wb[0][0])[i & 3].y);
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_bf16(wb[ks][nt], xa[ks][mt], acc[mt][nt]);
        }
        if (TFM && WR) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x006, C4_TFM_VALU, 0);     // then up to C4_TFM_VALU VALU / SALU
          }
        }
        if (C4_PRIO) __builtin_amdgcn_s_setprio(0);
        if (TFM && !C4_ABL_NOSTG && tap < C4_MAXV) vec_load_asm(cn * 64, tap);
        if (TFM) C4_AUX_TOUCH();
        C4_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        if (!C4_ONEBAR || grp) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      phase(std::integral_constant<int, 0>{}); phase(std::integral_constant<int, 1>{}); phase(std::integral_constant<int, 2>{});
      phase(std::integral_constant<int, 3>{}); phase(std::integral_constant<int, 4>{}); phase(std::integral_constant<int, 5>{});
      phase(std::integral_constant<int, 6>{}); phase(std::integral_constant<int, 7>{}); phase(std::integral_constant<int, 8>{});
      sidx = ring(sidx + 9);                             // 9 stages per chunk
    };
#if C4_CUNROLL == 1
    for (int c = 0; c < nchunks; ++c) chunk(c);
#else
    // the chunk loop's taken back-edge stalls a wave for 1 - 2 k cycles (measured with the C4_TIMING stamps: instruction fetch
    // of the branch target while the partner wave streams MFMAs); C4_CUNROLL chunks per iteration run as straight-line code
    for (int c = 0; c < nchunks; c += C4_CUNROLL) {
      chunk(c);
#pragma unroll
      for (int u = 1; u < C4_CUNROLL; ++u)
        if (c + u < nchunks) chunk(c + u);
    }
#endif
#undef C4_NVT
#if C4_TIMING
    if (lane == 0 && tcount < 8) tl[146 + 3 * tcount] = __builtin_readcyclecounter();          // K loop done
    if (C4_TIMING >= 2 && tfirst && p.dbg) {
      tl[144] = __builtin_readcyclecounter();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int i = lane; i < C4_NSTAMP; i += 64) p.dbg[((long)blockIdx.x * 8 + wave) * C4_NSTAMP + i] = tl[i];
    }
    tfirst = false;
#endif
    if (!C4_ONEBAR && C4_STAGGER && !grp) __builtin_amdgcn_s_barrier();   // re-synchronise the groups: both run the epilogue at the same time
    pb = (pb + nchunks) & 1;
    // everything the K loop issued (the next tile's stage-1 weights were requested one phase ago) is confirmed here, so that
    // the next tile's first phase does not have to wait behind this epilogue's stores
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf) :: "memory");

    if (C4_ABL_NOEPI) {     // timing ablation: keep the accumulators live, skip the epilogue
      float t = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) t += acc[mt][nt][r];
      if (t == 12345.678f) p.y[0] = (bf16_t)1;
    } else
    // ---- epilogue (as conv3): a lane owns, for its pixel, channel quads 8q + 4h + {0..3} of each n-tile; bf16 pack in
    // registers, v_permlane32_swap pairs quads into 16-byte vectors, straight NHWC stores; residual / GroupNorm-input rows
    // are fetched up front in the STORE layout
    {
    int te = threadIdx.x;
    asm volatile("" : "+v"(te));
    const int h = (te >> 5) & 1, pl = te & 31;
    const int nbase = nb * C4_BN + wn * 64;                 // first channel of this wave
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
          for (int k = 0; k < 2; ++k) aux[mt][nt][k] = C4_ABL_NOAUX ? make_uint4(te, mt, nt, k) : *(const uint4*)(r + (nt * 32 + 16 * k) * 2);
      }
    } else if (STM == 2) {
      const char* const xb = (const char*)(p.st_x + ((long)img * p.H + row0) * p.W * p.st_ldx + nbase);
      const unsigned rsx = (unsigned)(p.W * p.st_ldx * 2), lane_x = (unsigned)(((x0 + pl) * p.st_ldx + 8 * h) * 2);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int k = 0; k < 2; ++k) aux[mt][nt][k] = C4_ABL_NOAUX ? make_uint4(te, mt, nt, k) : *(const uint4*)(xb + mt * rsx + lane_x + (nt * 32 + 16 * k) * 2);
    }

#if C4_TIMING
    if (tcount == 1 && lane == 0) tl[160 + 0] = __builtin_readcyclecounter();
#endif
    float ss[16];                                           // [0..7]: sum 1 per channel quad, [8..15]: sum 2
#pragma unroll
    for (int i = 0; i < 16; ++i) ss[i] = 0.f;
    // ---- sweep 1: (residual,) pack, forward statistics, store
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
      if (STM == 1) {     // statistics of the fp32 values (before the bf16 rounding of the store)
        t1v[nt][k][0] += (f32x2){v[0], v[1]}; t1v[nt][k][0] += (f32x2){v[2], v[3]};
        t2v[nt][k][0] += (f32x2){v[0], v[1]} * (f32x2){v[0], v[1]}; t2v[nt][k][0] += (f32x2){v[2], v[3]} * (f32x2){v[2], v[3]};
        t1v[nt][k][1] += (f32x2){v[4], v[5]}; t1v[nt][k][1] += (f32x2){v[6], v[7]};
        t2v[nt][k][1] += (f32x2){v[4], v[5]} * (f32x2){v[4], v[5]}; t2v[nt][k][1] += (f32x2){v[6], v[7]} * (f32x2){v[6], v[7]};
      }
      // lanes l and l + 32 hold the same pixel: after the swaps lanes < 32 own channels 16k .. 16k+7 and lanes >= 32 own
      // 16k+8 .. 16k+15 of their n-tile -> one 16-byte store each
      {
        auto r = __builtin_amdgcn_permlane32_swap(w0x, w1x, false, false);
        w0x = r[0]; w1x = r[1];
        r = __builtin_amdgcn_permlane32_swap(w0y, w1y, false, false);
        w0y = r[0]; w1y = r[1];
      }
      const uint4 o = make_uint4(w0x, w0y, w1x, w1y);
      if (STM == 2) wst[STM == 2 ? mt : 0][nt][k] = o;        // turned into dz and stored by sweep 2
      else if (!C4_ABL_NOSTORE || o.x == 0x12345678u) *(uint4*)(yb + mt * rsy + coff + lane_y) = o;
    };
    // pixel-row-major item order: the four 16-byte vectors of a pixel (this wave's 64 channels = one 128-byte run) are stored
    // back to back so that L2 merges them into whole lines
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k = 0; k < 2; ++k) emit(mt, nt, k);
    if (STM == 1) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k = 0; k < 2; ++k) {     // quad index nt*4 + 2k (+1) = channels nt*32 + 16k + 4h (+8) .. +3
          ss[nt * 4 + 2 * k] = t1v[nt][k][0][0] + t1v[nt][k][0][1]; ss[8 + nt * 4 + 2 * k] = t2v[nt][k][0][0] + t2v[nt][k][0][1];
          ss[nt * 4 + 2 * k + 1] = t1v[nt][k][1][0] + t1v[nt][k][1][1]; ss[8 + nt * 4 + 2 * k + 1] = t2v[nt][k][1][0] + t2v[nt][k][1][1];
        }
    }

#if C4_TIMING
    if (tcount == 1 && lane == 0) tl[160 + 1] = __builtin_readcyclecounter();
#endif
    // ---- sweep 2 (backward statistics), in the STORE layout: this lane's vector (nt, k) = channels nt*32 + 16k + 8h .. +7 of
    // its pixel, from the stored (rounded) dy and the GroupNorm input.  Quad index nt*4 + 2k (+1) = channels nt*32 + 16k + 8h (+4) .. +3
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
              dzv[e] = dy[e] * (C4_ABL_NOSILU ? z : silu_grad_fast(z));
              const float adz = a * dzv[e];
              t1[e >> 2] += adz;
              t2[e >> 2] += adz * xg[e];
            }
            // the tensor this conv leaves in HBM is dz = dy * silu'(z): its consumers need no transcendental at all
            wst[STM == 2 ? mt : 0][nt][k] = pack16<bf16_t>(dzv);
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float2 m = *(const float2*)(p.st_mr + ((long)img * 32 + (cs + 4 * q) / cpg) * 2);
            ss[nt * 4 + 2 * k + q] = t1[q];
            ss[8 + nt * 4 + 2 * k + q] = (t2[q] - m.x * t1[q]) * m.y;      // sum a*dz*xhat over this lane's values
          }
        }
      }

#if C4_TIMING
    if (tcount == 1 && lane == 0) tl[160 + 2] = __builtin_readcyclecounter();
#endif
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (!C4_ABL_NOSTORE || wst[STM == 2 ? mt : 0][nt][k].x == 0x12345678u) *(uint4*)(yb + mt * rsy + (unsigned)((nt * 32 + 16 * k) * 2) + lane_y) = wst[STM == 2 ? mt : 0][nt][k];
    }
#if C4_TIMING
    if (tcount == 1 && lane == 0) tl[160 + 3] = __builtin_readcyclecounter();
#endif

    if (STM) {
      // 16 partial sums per lane, 32 pixel lanes per half-wave: butterfly reduce-scatter (8 + 4 + 2 + 1 exchanges, then one
      // plain exchange) leaves value j = bits (4,3,2,1) of the lane index, summed over the half-wave, in every lane
#pragma unroll
      for (int st_ = 0; st_ < 4; ++st_) {
        const int off = 16 >> st_, n = 8 >> st_;            // partner distance, values kept
        const bool up = (pl & off) != 0;
#pragma unroll
        for (int j = 0; j < n; ++j) {
          float lo = ss[j], hi = ss[j + n];
          asm volatile("" : "+v"(lo), "+v"(hi));           // (opaque: otherwise a 16-way select chain per access, conv3.hip)
          const float send = up ? lo : hi;
          const float keep = up ? hi : lo;
          ss[j] = keep + __shfl_xor(send, off, 64);
        }
      }
      ss[0] += __shfl_xor(ss[0], 1, 64);
      float* sred = (float*)(smem + C4_LDS + 64);           // [wave][h][16], own LDS region (the patch buffers hold the next tile)
      if ((pl & 1) == 0) sred[(wave * 2 + h) * 16 + (pl >> 1)] = ss[0];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // LDS only: a __syncthreads() here would also wait for the output stores
      __builtin_amdgcn_s_barrier();
      if (tid < 64) {
        // tid -> (wn', h', j): j < 8: sum 1 of quad j = nt * 4 + q, j >= 8: sum 2; summed over the four pixel-row waves
        const int wn2 = tid >> 5, h2 = (tid >> 4) & 1, j = tid & 15;
        float a = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) a += sred[((w4 * 2 + wn2) * 2 + h2) * 16 + j];
        const int i = j & 7;          // quad (nt = i >> 2, 2k + q = i & 3): accumulator layout (mode 1) or store layout (mode 2)
        const int ch = nb * C4_BN + wn2 * 64 + (i >> 2) * 32 + (STM == 2 ? ((i >> 1) & 1) * 16 + 8 * h2 + (i & 1) * 4 : (i & 3) * 8 + 4 * h2);
        if (C4_ABL_NOATOM) { if (a == 12345.678f) p.st_sums[0] = a; }
        else atomicAdd(p.st_sums + ((long)img * 32 + ch / cpg) * 2 + (j >> 3), (double)a);
      }
      // (sred is rewritten by the next tile's epilogue only after a full K loop of barriers)
    }
    }
#if C4_TIMING
    if (lane == 0 && tcount < 8) tl[147 + 3 * tcount] = __builtin_readcyclecounter();          // epilogue done (stores issued)
    ++tcount;
#endif
    if (!has_next) break;
    cur = nxt;
  }
  // drain the redundant tail loads / DMAs before the wave ends
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#if C4_TIMING
  if (p.dbg) {
    if (lane == 0) { tl[172] = __builtin_readcyclecounter(); tl[173] = __builtin_amdgcn_s_memrealtime(); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 145 + lane; i < C4_NSTAMP; i += 64) p.dbg[((long)blockIdx.x * 8 + wave) * C4_NSTAMP + i] = tl[i];
  }
#endif
}

template <int TF, int STM, bool RES>
int launch4(const Conv3Params& p, hipStream_t st) {
  auto kern = conv4_kernel<TF, STM, RES>;
  static std::atomic<unsigned long long> granted{0};     // dynamic-LDS cap raised once per (instantiation, device)
  int dev = 0;
  KDIP_HIP_CHECK(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(granted.load(std::memory_order_acquire) & bit)) {
    KDIP_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C4_LDS_TOTAL));
    granted.fetch_or(bit, std::memory_order_release);
  }
  static std::atomic<int> num_cu{0};
  if (!num_cu.load()) {
    int n = 0;
    KDIP_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    num_cu.store(n > 0 ? n : 256);
  }
  const long ntiles8 = ((long)p.mtiles * p.nblkN + 7) / 8 * 8;
  long grid = (long)stream_cus(st, num_cu.load()) / 8 * 8;               // persistent launch: one resident block per CU, a multiple of 8 (one share per XCD)
  if (grid > ntiles8) grid = ntiles8;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C4_NTHR), C4_LDS_TOTAL, st, p);
  return KDIP_OK;
}

unsigned long long* g_c4_dbg = nullptr;
// 0: automatic choice, 3 / 4: force the second- / third-generation kernel where the shape allows (initial value: KDIP_CONV_GEN)
std::atomic<int> g_conv_gen{[] { const char* e = getenv("KDIP_CONV_GEN"); const int v = e ? atoi(e) : 0; return v == 3 || v == 4 ? v : 0; }()};

}  // namespace

void conv_debug_generation(int gen) { g_conv_gen.store(gen); }
int conv4_debug_timing(void* buf) {
  if (!C4_TIMING) return set_error(KDIP_ERR_UNSUPPORTED, "conv4 timing: library not built with -DC4_TIMING=1");
  g_c4_dbg = (unsigned long long*)buf;
  return KDIP_OK;
}
int conv4_tf_max_cin(int tf) { return c4_max_cin(tf); }

// tf / stm / res: fusion mode of the launch.  Generation 0 (automatic) keeps every launch on conv3: measured in the network
// (bench.py roofline leg, 8 images per launch) the third-generation kernel is level with conv3 at 256 -> 128 @ 256^2 (297 vs 293 us)
// and slower at 128 -> 128 with the GroupNorm staging (186 vs 161 us), and the whole step is unchanged (40.7 ms both ways);
// in the back-to-back micro-benchmark it wins 3 - 6 % on light epilogues and loses 0 - 5 % on the residual / GroupNorm-backward
// ones.  KDIP_CONV_GEN=4 / kdip_debug_conv_generation(4) selects it wherever the shape allows.
bool conv4_shape_ok(const Conv3Params& p, int tf, int stm, bool res) {
  if (g_conv_gen.load() == 3) return false;
  if (g_conv_gen.load() != 4) return false;
  (void)tf; (void)stm; (void)res;
  if (p.H % C4_TH != 0 || p.W % C4_TW != 0 || p.Cout % C4_BN != 0 || p.Cin % 32 != 0 || p.Cout > C4_MAXCOUT) return false;
  const long tiles = (long)p.B * (p.H / C4_TH) * (p.W / C4_TW) * (p.Cout / C4_BN);
  return g_conv_gen.load() == 4 || tiles >= C4_MIN_TILES;
}

int conv4_launch(const Conv3Params& p0, int tf, int stm, bool res, hipStream_t st) {
  Conv3Params p = p0;
  p.dbg = C4_TIMING ? g_c4_dbg : nullptr;
  p.tilesX = p.W / C4_TW; p.tilesY = p.H / C4_TH; p.mtiles = p.B * p.tilesX * p.tilesY; p.nblkN = p.Cout / C4_BN;
  auto magic = [](unsigned d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); };
  p.mg_nblk = magic((unsigned)p.nblkN); p.mg_tpi = magic((unsigned)(p.tilesX * p.tilesY)); p.mg_tx = magic((unsigned)p.tilesX);
  KDIP_REQUIRE((long)p.mtiles * p.nblkN * (long)(p.tilesX * p.tilesY > p.nblkN ? p.tilesX * p.tilesY : p.nblkN) < (1L << 32), "conv4: too many tiles for the 32-bit magic divisions");
  KDIP_REQUIRE(p.Cin <= c4_max_cin(tf), "conv4: too many input channels (%d) for the staging-transform table", p.Cin);
  int rc;
#define C4_GO(T, S, R) rc = launch4<T, S, R>(p, st)
  if (tf == 0 && stm == 0) { if (res) C4_GO(0, 0, true); else C4_GO(0, 0, false); }
  else if (tf == 0 && stm == 1) { if (res) C4_GO(0, 1, true); else C4_GO(0, 1, false); }
  else if (tf == 1 && stm == 0) { if (res) C4_GO(1, 0, true); else C4_GO(1, 0, false); }
  else if (tf == 1 && stm == 1) { if (res) C4_GO(1, 1, true); else C4_GO(1, 1, false); }
  else if (tf == 0 && stm == 2) C4_GO(0, 2, false);
  else if (tf == 2 && stm == 0) C4_GO(2, 0, false);
  else C4_GO(2, 2, false);
#undef C4_GO
  return rc;
}

}  // namespace kdip
