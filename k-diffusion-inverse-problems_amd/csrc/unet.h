// ADM UNet executor: plan + packed weights + forward (with activation stash) + input-VJP.
#pragma once
#include <array>
#include <map>
#include <string>
#include <vector>
#include "common.h"
#include "det.h"

namespace kdip {

struct UNetConfig {
  int image_size, in_channels, model_channels, out_channels, num_res_blocks;
  std::vector<int> attention_ds;     // downsample rates with attention
  std::vector<int> channel_mult;
  int num_head_channels;
};

struct ConvW {
  int cin = 0, cout = 0, cin_pad = 0, ntaps = 0;
  void* wf = nullptr;      // packed forward weights (device)
  void* wb = nullptr;      // packed dgrad weights (device): cin' = cout (padded to 32), cout' = cin
  void *wf_alt = nullptr, *wb_alt = nullptr;      // DT_F32H3 handles: the same weights in the bf16-headed encoding (DT_F32X3), for calls redone outside the fp16 window
  int cin_pad_b = 0;       // padded K of the dgrad conv (= pad32(cout))
  float* bias = nullptr;   // device fp32 [cout]
};
struct LinW { int in = 0, out = 0; float* w = nullptr; float* b = nullptr; };   // fp32 [out][in]
struct GnW { int C = 0; float* gamma = nullptr; float* beta = nullptr; };

struct Saved {            // per-layer stash for the VJP
  const void* x = nullptr; long ldx = 0;
  void* h2 = nullptr;
  float *coef1 = nullptr, *mr1 = nullptr, *coef2 = nullptr, *mr2 = nullptr;
  void* qkv = nullptr; void* P = nullptr;
  void* ao = nullptr; float* lse = nullptr;      // fused attention: saved attention output and per-row log-sum-exp (instead of P)
  int B = 0, H = 0, W = 0;
};

struct Layer {
  int kind;                // 0 conv, 1 res, 2 attn
  std::string prefix;
  int cin, cout, mode;     // mode: 0 none, 1 down, 2 up
  ConvW conv;              // kind 0
  GnW n1, n2; ConvW c1, c2, skip; LinW emb; int emb_off = 0; bool has_skip = false;   // kind 1 (emb_off: row offset in emb_all)
  GnW norm; ConvW qkv, proj; int heads = 0;                          // kind 2
  Saved sv;
};

struct Arena {
  char* base = nullptr; size_t cap = 0, off = 0, peak = 0;
  bool overflow = false;            // a real pass asked for more than the plan reserved: the pass is aborted by run() / vjp()
  void* alloc(size_t bytes) {
    size_t a = (off + 255) & ~(size_t)255;
    off = a + bytes;
    if (off > peak) peak = off;
    if (!base) return (void*)(uintptr_t)(a + 4096);                   // dry run: fake non-null address
    if (off > cap) {                // never hand out memory past the arena: park the request at the start (garbage results, no
      overflow = true;              // out-of-bounds write as long as the request itself fits) -- the caller checks `overflow`
      return (void*)base;
    }
    return (void*)(base + a);
  }
  void reset() { off = 0; }
};

struct UNet {
  UNetConfig cfg;
  DType dt;                         // activation storage type (DT_F32 | DT_BF16)
  DType cdt;                        // conv arithmetic / packed-weight type: dt, or DT_F32X3 / DT_F32H3 (fp32 storage, split-precision MFMA) when dt == DT_F32
  // DT_F32H3 (fp16-headed split: fast, exact only inside the fp16 window) handles carry the bf16-headed weights as well; x3_alt = 1 makes the
  // next forward / VJP run on them (kdip_unet_x3_head): how a caller redoes a call whose saturation flag came up.  Same tiles, same
  // workspace plan, same stash: only the conv arithmetic changes.
  bool has_alt = false, x3_alt = false;
  bool x3_force_alt = false;         // a weight tensor of the handle sits under the fp16 head's normal range: the handle runs bf16-headed throughout
  bool alt() const { return has_alt && (x3_alt || x3_force_alt); }
  DType ccdt() const { return alt() ? DT_F32X3 : cdt; }
  unsigned* x3_peaks = nullptr; int x3_npeaks = 0;      // fp16-headed passes: one word per conv launch (zeros arena) for the largest staged operand (x3_lowpeak_check at the end of the pass)
  static constexpr int X3_MAX_PEAKS = 1024;
  int device = 0;
  std::vector<std::vector<Layer>> inp, out;
  std::vector<Layer> mid;
  int final_ch = 0;
  LinW te0, te2;
  LinW emb_all;                     // all ResBlock emb_layers.1 stacked: one launch per forward
  int emb_total = 0;
  GnW out_norm; ConvW out_conv; ConvW cov_conv; bool has_cov = false;
  // fp32-storage modes: the few-channel 3x3 convs restated as 1x1 convs (elementwise.hip: im2col3_nchw / tap_gather_nchw; built by finalize()).
  //   in_k1  image conv forward    : K = 9 x in_channels (taps folded into K),   N = first block width
  //   in_n1  image conv dgrad      : K = first block width,                     N = 9 x in_channels (taps folded into N)   -> wb
  //   out_n1 output head forward   : K = final_ch,                              N = 9 x out_channels                      -> wf
  //   out_k1 output head dgrad     : K = 9 x out_channels,                      N = final_ch                               -> wb
  ConvW in_k1, in_n1, out_n1, out_k1; bool tapfold = false;
  std::map<std::string, std::vector<float>> raw;      // host fp32 parameters by reference state_dict name
  std::map<std::string, std::vector<long>> raw_shape;
  bool finalized = false;
  std::vector<void*> dev_allocs;

  Arena persist, scratch, zeros;     // zeros: fp64 statistics accumulators, cleared once per forward / VJP
  size_t zeros_fwd_end = 0;
  unsigned* x3_amax = nullptr;       // split-precision mode: bits of max |cotangent| of the current VJP (fp16 window of the dgrad convs' A operand)
  unsigned* x3_sat = nullptr;        // split-precision mode: sticky device word, bit 0 = some conv launch staged an operand outside the fp16 window (kdip_unet_x3_saturated)
  long x3_weight_sat = 0;            // ... weights of this handle outside the window at pack time
  int x3_window_per_launch = 0;      // 1: every dgrad launch takes its window from a sampled max |g| of its own input instead (kdip_unet_x3_window)
  // deterministic reductions (det.h): on for the fp32-storage modes (f32, bf16x3) -- GroupNorm statistics and split-K partial sums
  // are added in a fixed order, two runs of one call are bitwise equal; the bf16 throughput mode keeps floating-point atomics
  bool det = false;
  float* sk_ws = nullptr; long sk_ws_floats = 0;   // split-K workspace of the small-spatial convs (zeros arena; kept zero by the finalize kernel)
  std::map<std::pair<const void*, int>, double*> fused_stats;   // (tensor, channels) -> GroupNorm sums already accumulated by its producer
  std::map<const double*, DetPending> pending_stats;            // ... sums whose fixed-order finish pass the producer left to the consumer (det.h: DetPending; slab in `persist`)
  int ws_B = 0;                       // largest batch planned so far
  long ws_generation = 0;             // bumped whenever an arena is re-allocated: captured hipGraphs hold raw arena pointers
  std::map<int, std::array<size_t, 3>> planned;   // batch -> (persist, scratch, zeros) peak bytes of a dry forward + VJP at that batch
  bool dry = false;
  // state of the last forward (for the VJP)
  int last_B = 0; bool have_stash = false;
  const void* final_h = nullptr; float *out_coef = nullptr, *out_mr = nullptr;
  std::vector<const void*> hs_ptr; std::vector<int> hs_C;
  std::vector<void*> cat_ptr;

  ~UNet();
  int build_plan();
  int load(const char* name, const float* data, const long* shape, int ndim);
  int finalize();
  int ensure_workspace(int B);
  int run(hipStream_t st, const float* x_nchw, const float* t, int B, float in_scale, float* out_nchw,
          float* cov_nchw, float* feat_nchw, int save);
  int vjp(hipStream_t st, const float* cot_nchw, float* gx_nchw);
  int forward_impl(hipStream_t st, const float* x_nchw, const float* t, int B, float in_scale, float* out_nchw,
                   float* cov_nchw, float* feat_nchw);
  int vjp_impl(hipStream_t st, const float* cot_nchw, float* gx_nchw);
  size_t esize() const { return dt == DT_BF16 ? 2 : 4; }
};

void unet_debug_defer_finish(int on);      // A/B switch of the deferred statistics finish (default on; bit-identical results)
void unet_debug_gn_fold(int on);      // A/B switch of the GroupNorm-coefficient fold (default on)

}  // namespace kdip
