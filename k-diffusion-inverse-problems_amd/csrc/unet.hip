// ADM UNet executor for MI355X: forward with activation stash and a hand-written
// input-VJP (no autograd), all NHWC, all device work in the HIP kernels of this library.
//
// Mirrors UNetModel.__init__/forward (guided_diffusion/unet.py:429-668) for the
// configuration the sampling scripts build (use_scale_shift_norm, resblock_updown,
// num_head_channels=64, legacy attention order, learn_sigma) and replaces
// torch.autograd.grad(..., x) of condition/condition.py:136,146,155,172 with an explicit
// reverse sweep: conv dgrad = the same implicit-GEMM kernel on flipped/transposed packed
// weights; GroupNorm/FiLM/SiLU backward = two streaming passes; attention backward = 4
// batched MFMA GEMMs + a row kernel.
#include <math.h>
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include "kernels.h"
#include "unet.h"

namespace kdip {

#define RUN(call)                     \
  do {                                \
    if (!dry) {                       \
      int _rc = (call);               \
      if (_rc) return _rc;            \
    }                                 \
  } while (0)
#define CK(call)                      \
  do {                                \
    int _rc = (call);                 \
    if (_rc) return _rc;              \
  } while (0)

static inline int pad32(int c) { return (c + 31) / 32 * 32; }

// fp32-storage modes, read once (A/B builds and tools/ only):
//   KDIP_TAPFOLD (default 1): the 3- / 6-channel 3x3 convs (image conv, output head and their input-gradients) run as 1x1 convs with the
//                             taps folded into K or N (unet.h: in_k1 / in_n1 / out_n1 / out_k1) instead of padding K or N to 32
static const int g_tapfold = [] { const char* e = getenv("KDIP_TAPFOLD"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }();

// ------------------------------------------------------------------ tiny fp32 linear ----
// y[b][o] = bias[o] + sum_i act(x[b][i]) * w[o][i]; one wavefront per output (emb MLPs,
// guided_diffusion/unet.py:199-205,472-477: M = batch, latency-bound).
__global__ void linear_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, int B, int in, int out, int silu_in,
                                  float* __restrict__ y, int silu_out) {
  // one wave per 4 output features, 8 batch rows at a time: each weight row is streamed once and every (L2-resident)
  // input row is re-read by out/4 waves instead of out
  constexpr int OC = 4, BT = 8;
  const int oc0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * OC, lane = threadIdx.x & 63;
  if (oc0 >= out) return;
  for (int b0 = 0; b0 < B; b0 += BT) {
    float s[OC][BT];
#pragma unroll
    for (int o = 0; o < OC; ++o)
#pragma unroll
      for (int j = 0; j < BT; ++j) s[o][j] = 0.f;
    for (int i = lane; i < in; i += 64) {
      float wv[OC], xv[BT];
#pragma unroll
      for (int o = 0; o < OC; ++o) wv[o] = oc0 + o < out ? w[(long)(oc0 + o) * in + i] : 0.f;
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        xv[j] = b0 + j < B ? x[(long)(b0 + j) * in + i] : 0.f;
        if (silu_in) xv[j] = xv[j] / (1.f + expf(-xv[j]));
      }
#pragma unroll
      for (int o = 0; o < OC; ++o)
#pragma unroll
        for (int j = 0; j < BT; ++j) s[o][j] += xv[j] * wv[o];
    }
#pragma unroll
    for (int o = 0; o < OC; ++o)
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        const float t = wave_sum(s[o][j]);
        if (lane == 0 && oc0 + o < out && b0 + j < B) {
          const float v = t + bias[oc0 + o];
          y[(long)(b0 + j) * out + oc0 + o] = silu_out ? v / (1.f + expf(-v)) : v;      // (silu_out: the consumer applies SiLU to every
        }                                                                                //  element once per OUTPUT block otherwise)
      }
  }
}
static int linear_f32(hipStream_t st, const float* x, const LinW& L, int B, int silu_in, float* y, int silu_out = 0) {
  hipLaunchKernelGGL(linear_f32_kernel, dim3(cdiv((long)L.out, 16)), dim3(256), 0, st, x, L.w, L.b, B, L.in, L.out,
                     silu_in, y, silu_out);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// ------------------------------------------------------------------------------ plan ----
UNet::~UNet() {
  for (void* p : dev_allocs) (void)hipFree(p);
  if (persist.base) (void)hipFree(persist.base);
  if (scratch.base) (void)hipFree(scratch.base);
  if (zeros.base) (void)hipFree(zeros.base);
}

int UNet::build_plan() {
  // guided_diffusion/unet.py:482-619
  const int mc = cfg.model_channels;
  auto has_attn = [&](int ds) {
    for (int a : cfg.attention_ds) if (a == ds) return true;
    return false;
  };
  auto mkres = [&](const std::string& p, int cin, int cout, int mode) {
    Layer L; L.kind = 1; L.prefix = p; L.cin = cin; L.cout = cout; L.mode = mode; L.has_skip = (cin != cout);
    return L;
  };
  auto mkattn = [&](const std::string& p, int ch) {
    Layer L; L.kind = 2; L.prefix = p; L.cin = L.cout = ch; L.mode = 0; L.heads = ch / cfg.num_head_channels;
    return L;
  };
  KDIP_REQUIRE(cfg.num_head_channels == 64, "only num_head_channels=64 is supported");
  int ch = cfg.channel_mult[0] * mc;
  {
    Layer L; L.kind = 0; L.prefix = "input_blocks.0.0"; L.cin = cfg.in_channels; L.cout = ch; L.mode = 0;
    inp.push_back({L});
  }
  std::vector<int> chans = {ch};
  int ds = 1;
  const int nlev = (int)cfg.channel_mult.size();
  for (int level = 0; level < nlev; ++level) {
    int mult = cfg.channel_mult[level];
    for (int r = 0; r < cfg.num_res_blocks; ++r) {
      std::string bp = "input_blocks." + std::to_string(inp.size());
      std::vector<Layer> ls;
      ls.push_back(mkres(bp + ".0", ch, mult * mc, 0));
      ch = mult * mc;
      if (has_attn(ds)) ls.push_back(mkattn(bp + ".1", ch));
      inp.push_back(ls);
      chans.push_back(ch);
    }
    if (level != nlev - 1) {
      std::string bp = "input_blocks." + std::to_string(inp.size());
      inp.push_back({mkres(bp + ".0", ch, ch, 1)});
      chans.push_back(ch);
      ds *= 2;
    }
  }
  mid.push_back(mkres("middle_block.0", ch, ch, 0));
  mid.push_back(mkattn("middle_block.1", ch));
  mid.push_back(mkres("middle_block.2", ch, ch, 0));
  for (int level = nlev - 1; level >= 0; --level) {
    int mult = cfg.channel_mult[level];
    for (int i = 0; i < cfg.num_res_blocks + 1; ++i) {
      int ich = chans.back(); chans.pop_back();
      std::string bp = "output_blocks." + std::to_string(out.size());
      std::vector<Layer> ls;
      ls.push_back(mkres(bp + ".0", ch + ich, mc * mult, 0));
      ch = mc * mult;
      if (has_attn(ds)) ls.push_back(mkattn(bp + "." + std::to_string(ls.size()), ch));
      if (level && i == cfg.num_res_blocks) {
        ls.push_back(mkres(bp + "." + std::to_string(ls.size()), ch, ch, 2));
        ds /= 2;
      }
      out.push_back(ls);
    }
  }
  final_ch = ch;
  return KDIP_OK;
}

int UNet::load(const char* name, const float* data, const long* shape, int ndim) {
  KDIP_REQUIRE(!finalized, "unet_load after finalize");
  long n = 1;
  std::vector<long> sh;
  for (int i = 0; i < ndim; ++i) { n *= shape[i]; sh.push_back(shape[i]); }
  raw[name] = std::vector<float>(data, data + n);
  raw_shape[name] = sh;
  return KDIP_OK;
}

int UNet::finalize() {
  KDIP_REQUIRE(!finalized, "unet already finalized");
  KDIP_HIP_CHECK(hipSetDevice(device));
  int rc = KDIP_OK;
  const long x3_w0 = x3_weight_saturations(), x3_l0 = x3_weight_subwindow();
  auto need = [&](const std::string& k, long numel) -> const float* {
    auto it = raw.find(k);
    if (it == raw.end()) { rc = set_error(KDIP_ERR_STATE, "missing parameter '%s'", k.c_str()); return nullptr; }
    if ((long)it->second.size() != numel) {
      rc = set_error(KDIP_ERR_ARG, "parameter '%s' has %ld elements, expected %ld", k.c_str(), (long)it->second.size(), numel);
      return nullptr;
    }
    return it->second.data();
  };
  auto upload = [&](const void* host, size_t bytes) -> void* {
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) { rc = set_error(KDIP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes); return nullptr; }
    dev_allocs.push_back(d);
    if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = set_error(KDIP_ERR_HIP, "weight upload failed"); return nullptr; }
    return d;
  };
  auto mkconv = [&](ConvW& c, const std::string& p, int cin, int cout, int ntaps) {
    const float* w = need(p + ".weight", (long)cout * cin * ntaps);
    const float* b = need(p + ".bias", cout);
    if (!w || !b) return;
    c.cin = cin; c.cout = cout; c.ntaps = ntaps; c.cin_pad = pad32(cin); c.cin_pad_b = pad32(cout);
    std::vector<char> buf(packed_weight_bytes(cdt, ntaps, c.cin_pad, cout));
    pack_conv_weight(cdt, w, cout, cin, ntaps, 0, c.cin_pad, buf.data());
    c.wf = upload(buf.data(), buf.size());
    std::vector<char> bufb(packed_weight_bytes(cdt, ntaps, c.cin_pad_b, cin));
    pack_conv_weight(cdt, w, cout, cin, ntaps, 1, c.cin_pad_b, bufb.data());
    c.wb = upload(bufb.data(), bufb.size());
    if (cdt == DT_F32H3) {      // the bf16-headed encoding of the same weights (same size: two 16-bit planes)
      pack_conv_weight(DT_F32X3, w, cout, cin, ntaps, 0, c.cin_pad, buf.data());
      c.wf_alt = upload(buf.data(), buf.size());
      pack_conv_weight(DT_F32X3, w, cout, cin, ntaps, 1, c.cin_pad_b, bufb.data());
      c.wb_alt = upload(bufb.data(), bufb.size());
    }
    c.bias = (float*)upload(b, sizeof(float) * cout);
  };
  auto mkgn = [&](GnW& g, const std::string& p, int C) {
    const float* w = need(p + ".weight", C);
    const float* b = need(p + ".bias", C);
    if (!w || !b) return;
    g.C = C; g.gamma = (float*)upload(w, sizeof(float) * C); g.beta = (float*)upload(b, sizeof(float) * C);
  };
  auto mklin = [&](LinW& l, const std::string& p, int in, int out) {
    const float* w = need(p + ".weight", (long)in * out);
    const float* b = need(p + ".bias", out);
    if (!w || !b) return;
    l.in = in; l.out = out; l.w = (float*)upload(w, sizeof(float) * in * out); l.b = (float*)upload(b, sizeof(float) * out);
  };
  const int ted = cfg.model_channels * 4;
  mklin(te0, "time_embed.0", cfg.model_channels, ted);
  mklin(te2, "time_embed.2", ted, ted);
  std::vector<std::string> emb_names; std::vector<Layer*> emb_layers;
  auto do_layer = [&](Layer& L) {
    const std::string& p = L.prefix;
    if (L.kind == 0) mkconv(L.conv, p, L.cin, L.cout, 9);
    else if (L.kind == 1) {
      mkgn(L.n1, p + ".in_layers.0", L.cin);
      mkconv(L.c1, p + ".in_layers.2", L.cin, L.cout, 9);
      { L.emb.in = ted; L.emb.out = 2 * L.cout; emb_names.push_back(p + ".emb_layers.1"); emb_layers.push_back(&L); }
      mkgn(L.n2, p + ".out_layers.0", L.cout);
      mkconv(L.c2, p + ".out_layers.3", L.cout, L.cout, 9);
      if (L.has_skip) mkconv(L.skip, p + ".skip_connection", L.cin, L.cout, 1);
    } else {
      mkgn(L.norm, p + ".norm", L.cin);
      mkconv(L.qkv, p + ".qkv", L.cin, 3 * L.cin, 1);
      mkconv(L.proj, p + ".proj_out", L.cin, L.cin, 1);
    }
  };
  for (auto& b : inp) for (auto& L : b) do_layer(L);
  for (auto& L : mid) do_layer(L);
  for (auto& b : out) for (auto& L : b) do_layer(L);
  {   // stack all emb_layers.1 weights: [sum 2*cout, ted]
    emb_total = 0;
    for (Layer* L : emb_layers) { L->emb_off = emb_total; emb_total += L->emb.out; }
    std::vector<float> W((size_t)emb_total * ted), Bv(emb_total);
    for (size_t i = 0; i < emb_layers.size(); ++i) {
      Layer* L = emb_layers[i];
      const float* w = need(emb_names[i] + ".weight", (long)L->emb.out * ted);
      const float* b = need(emb_names[i] + ".bias", L->emb.out);
      if (!w || !b) break;
      memcpy(W.data() + (size_t)L->emb_off * ted, w, sizeof(float) * (size_t)L->emb.out * ted);
      memcpy(Bv.data() + L->emb_off, b, sizeof(float) * L->emb.out);
    }
    if (!rc) {
      emb_all.in = ted; emb_all.out = emb_total;
      emb_all.w = (float*)upload(W.data(), sizeof(float) * W.size());
      emb_all.b = (float*)upload(Bv.data(), sizeof(float) * Bv.size());
    }
  }
  mkgn(out_norm, "out.0", final_ch);
  mkconv(out_conv, "out.2", final_ch, cfg.out_channels, 9);
  if (raw.count("out_cov.weight")) {
    mkconv(cov_conv, "out_cov", final_ch, 6, 1);   // OpenAIDenoiserV2.out_cov (k_diffusion/external.py:141)
    has_cov = true;
  }
  if (!rc && g_tapfold && dt != DT_BF16 && 9 * cfg.in_channels <= 32 && 9 * cfg.out_channels <= 64) {
    // logical 3x3 conv Wl[Co][Ci][9] -> (a) taps in K: 1x1 weights [Co][9 Ci] (k = t*Ci + c);  (b) taps in N: 1x1 weights [Np][Ci], row n = t*Co + co
    // (rows >= 9 Co zero: N padded so that the 16-byte-vector epilogue applies).  dgrad: Wl[ci][co][t] = W[co][ci][8 - t] (flipped, transposed).
    auto logical = [&](const float* w, int cout, int cin, bool dgrad) {
      const int Co = dgrad ? cin : cout, Ci = dgrad ? cout : cin;
      std::vector<float> l((size_t)Co * Ci * 9);
      for (int o = 0; o < Co; ++o) for (int i = 0; i < Ci; ++i) for (int t = 0; t < 9; ++t)
        l[((size_t)o * Ci + i) * 9 + t] = dgrad ? w[((size_t)i * cin + o) * 9 + (8 - t)] : w[((size_t)o * cin + i) * 9 + t];
      return l;
    };
    // 1x1 conv weights [N][K] -> packed fragments (K padded to 32), in the handle's encoding and (DT_F32H3) in the bf16-headed one
    auto pack1 = [&](const std::vector<float>& w1, int N, int K, void** main, void** alt) {
      std::vector<char> buf(packed_weight_bytes(cdt, 1, pad32(K), N));
      pack_conv_weight(cdt, w1.data(), N, K, 1, 0, pad32(K), buf.data());
      *main = upload(buf.data(), buf.size());
      if (cdt == DT_F32H3) {
        pack_conv_weight(DT_F32X3, w1.data(), N, K, 1, 0, pad32(K), buf.data());
        *alt = upload(buf.data(), buf.size());
      }
    };
    auto taps_in_k = [&](const std::vector<float>& l, int Co, int Ci, void** main, void** alt) {
      std::vector<float> w1((size_t)Co * 9 * Ci);
      for (int o = 0; o < Co; ++o) for (int i = 0; i < Ci; ++i) for (int t = 0; t < 9; ++t) w1[(size_t)o * 9 * Ci + t * Ci + i] = l[((size_t)o * Ci + i) * 9 + t];
      pack1(w1, Co, 9 * Ci, main, alt);
    };
    auto taps_in_n = [&](const std::vector<float>& l, int Co, int Ci, int Np, void** main, void** alt) {
      std::vector<float> w1((size_t)Np * Ci, 0.f);
      for (int o = 0; o < Co; ++o) for (int i = 0; i < Ci; ++i) for (int t = 0; t < 9; ++t) w1[((size_t)t * Co + o) * Ci + i] = l[((size_t)o * Ci + i) * 9 + t];
      pack1(w1, Np, Ci, main, alt);
    };
    const int ic = cfg.in_channels, oc = cfg.out_channels, c0 = inp[0][0].cout, cf = final_ch;
    const float* wi = need("input_blocks.0.0.weight", (long)c0 * ic * 9);
    const float* wo = need("out.2.weight", (long)oc * cf * 9);
    if (wi && wo) {
      const int nin = pad32(9 * ic), nout = pad32(9 * oc);
      in_k1.ntaps = 1; in_k1.cin = 9 * ic; in_k1.cin_pad = pad32(9 * ic); in_k1.cout = c0; in_k1.bias = inp[0][0].conv.bias;
      taps_in_k(logical(wi, c0, ic, false), c0, ic, &in_k1.wf, &in_k1.wf_alt);
      in_n1.ntaps = 1; in_n1.cin = nin; in_n1.cout = c0; in_n1.cin_pad_b = pad32(c0);                       // conv_b: K = cin_pad_b, N = cin
      taps_in_n(logical(wi, c0, ic, true), ic, c0, nin, &in_n1.wb, &in_n1.wb_alt);
      out_n1.ntaps = 1; out_n1.cin = cf; out_n1.cin_pad = pad32(cf); out_n1.cout = nout; out_n1.bias = nullptr;   // (the bias is added by tap_gather_nchw)
      taps_in_n(logical(wo, oc, cf, false), oc, cf, nout, &out_n1.wf, &out_n1.wf_alt);
      out_k1.ntaps = 1; out_k1.cin = cf; out_k1.cout = 9 * oc; out_k1.cin_pad_b = pad32(9 * oc);
      taps_in_k(logical(wo, oc, cf, true), cf, oc, &out_k1.wb, &out_k1.wb_alt);
      tapfold = true;
    }
  }
  if (!rc && is_x3(cdt)) {
    const unsigned zero = 0;
    x3_amax = (unsigned*)upload(&zero, sizeof(zero));
    x3_sat = (unsigned*)upload(&zero, sizeof(zero));
  }
  if (rc) return rc;
  x3_weight_sat = x3_weight_saturations() - x3_w0;      // (thread-local counter: exactly this handle's weights)
  x3_force_alt = has_alt && x3_weight_subwindow() - x3_l0 > 0;
  raw.clear();
  finalized = true;
  return KDIP_OK;
}

// --------------------------------------------------------------------------- helpers ----
std::atomic<int> g_gn_fold{[] { const char* e = getenv("KDIP_GN_FOLD"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }()};      // 1: conv3 computes the GroupNorm staging coefficients itself (no gn_coef / gn_merge_stats / gn_bwd_coef launches)
void unet_debug_gn_fold(int on) { g_gn_fold.store(on ? 1 : 0); }
// 1: a forward conv with fused deterministic statistics leaves their finish pass to the GroupNorm that consumes them (finish_pending / gn_forward below)
std::atomic<int> g_defer_finish{[] { const char* e = getenv("KDIP_DEFER_FINISH"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }()};
void unet_debug_defer_finish(int on) { g_defer_finish.store(on ? 1 : 0); }
// fp32-storage modes (f32 / bf16x3), read once (A/B builds and tools/ only; the workspace plan of a batch depends on them):
//   KDIP_STORE_DZ (default 0): the backward-statistics epilogue of a dgrad conv leaves dz = dy * silu'(z) instead of dy (the second write of the
//                              tensor measured +24 us on the 128 -> 128 @ 256^2 dgrad launches: the consumers apply silu' themselves instead)
//   KDIP_GNB_FOLD (default 1): the ResBlock-input gradient  gn_bwd_apply(g1) + skip-dgrad + concat gradient  is produced by the 1x1 skip dgrad
//                              conv's epilogue (ConvStats mode 3) on channel-changing blocks: no skip-gradient tensor, no gn_bwd_apply pass
static const int g_store_dz = [] { const char* e = getenv("KDIP_STORE_DZ"); return e ? atoi(e) : 0; }();      // 0 never, 1 only where a TFM-2 dgrad conv consumes the tensor, 2 everywhere
static const int g_gnb_fold = [] { const char* e = getenv("KDIP_GNB_FOLD"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }();
//   KDIP_X3_TF2 (default 0): split-precision mode, the GroupNorm backward between a ResBlock's two dgrad convs is applied while the second
//                            one stages its input (ConvStats::tf_mode 2): the gradient w.r.t. the GroupNorm input is never written.  MEASURED
//                            NEGATIVE (round 6, profiles/r06/ab_gnb_tf2.log, interleaved twice): gn_bwd_apply 7.6 -> 4.1 ms per profiled pass, but the
//                            two-tensor staging makes the 128 x 128-tile class 65.6 -> 78.1 ms (132 B of scratch at the 168-register budget, twice the
//                            staging loads in front of the weight-fragment queue): step 94.8 -> 104.6 ms.  Opt-in; tests/test_x3_gpu.py runs it once.
static const int g_x3_tf2 = [] { const char* e = getenv("KDIP_X3_TF2"); return e ? (atoi(e) != 0 ? 1 : 0) : 0; }();

namespace {
struct Ctx {
  UNet* u; hipStream_t st; bool dry; DType dt; size_t es;
  DType cdt() const { return u->ccdt(); }
  const void* wf(const ConvW& w) const { return u->alt() ? w.wf_alt : w.wf; }
  const void* wb(const ConvW& w) const { return u->alt() ? w.wb_alt : w.wb; }
  unsigned* peak_slot() const {      // fp16-headed pass: this launch's word of the low-side window watch
    if (!u->x3_peaks || u->ccdt() != DT_F32H3) return nullptr;
    const int i = u->x3_npeaks < UNet::X3_MAX_PEAKS - 1 ? u->x3_npeaks : UNet::X3_MAX_PEAKS - 1;
    ++u->x3_npeaks;
    return u->x3_peaks + i;
  }
};

// 1: a forward conv with fused deterministic statistics leaves their finish pass to the GroupNorm that consumes them, which runs it inside its coefficient
// kernel (conv_stats_finish_coef): one launch less in the dependency chain between two convs (kdip_debug / A-B switch: KDIP_DEFER_FINISH=0)
// run the finish pass of `sums` now if its producer deferred it
int finish_pending(Ctx& c, const double* sums, int B) {
  bool dry = c.dry;
  auto it = c.u->pending_stats.find(sums);
  if (it == c.u->pending_stats.end()) return KDIP_OK;
  RUN(conv_stats_finish(c.st, it->second, B, (double*)sums));
  c.u->pending_stats.erase(it);
  return KDIP_OK;
}
double* new_sums(Ctx& c, int B) { return (double*)c.u->zeros.alloc(sizeof(double) * B * 64); }

// deterministic modes: a slab for one launch's block partials (scratch arena: transient, planned by the dry run like every other
// buffer) + the handle's counters; returns false (atomics path) for the bf16 mode
bool det_ws(Ctx& c, size_t slab_bytes, DetWs& w) {
  if (!c.u->det) return false;
  w.slab = c.u->scratch.alloc(slab_bytes); w.slab_bytes = slab_bytes;
  return true;
}
size_t det_gn_bytes(int B, long HW) {      // gn_stats / gn_bwd_stats: [B][chunks][64] doubles, chunks <= min(ceil(1024 / B), HW / 32 + 1) (norm.hip, pick_chunk)
  long chunks = (1024 + B - 1) / B;
  if (HW / 32 + 1 < chunks) chunks = HW / 32 + 1;
  return (size_t)B * (chunks + 1) * 64 * sizeof(double);
}
size_t det_conv_bytes(int B, int H, int W, int Cout) {      // fused conv statistics: [tiles of 128 pixels][Cout / 4 vectors][2] floats
  // launch_cfg2 counts tiles from its TH x TW patch geometry (tilesX * tilesY, TH * TW = 128): for maps whose sides are not both
  // multiples of the patch sides that is more than ceil(HW / 128) -- bounded by cdiv(H, 8) * cdiv(W, 16) and by 2 ceil(HW / 128) + ...;
  // take the patch-geometry bound (equal to HW / 128 for every power-of-two map the kernels accept today)
  const long tiles = std::max<long>(((long)H * W + 127) / 128, (long)cdiv(H, 8) * cdiv(W, 16));
  return (size_t)B * tiles * ((Cout + 127) / 128 * 128 / 4) * 2 * sizeof(float);
}

// pool_W > 0: y / pool_x receive the 2x2-average-pooled activated / raw tensors instead (downsampling ResBlock)
// y == nullptr: statistics + coefficients only (the apply is fused into the consuming conv's input staging, conv3.hip)
// fold: when given (and y == nullptr: the apply is fused into the consuming conv), NO coefficient kernel is launched either -- the
// descriptor tells the conv where the statistics, gamma / beta / FiLM rows and the coef / mr output buffers are (Conv3Fuse::fold_*)
int gn_forward(Ctx& c, const void* x, long ldx, int B, long HW, const GnW& g, const float* film, int silu, void* y,
               long ldy, float** coef_out, float** mr_out, long film_ld = 0, int pool_W = 0, void* pool_x = nullptr, Conv3Fuse* fold = nullptr) {
  bool dry = c.dry;
  float* coef = (float*)c.u->persist.alloc(sizeof(float) * B * g.C * 2);
  float* mr = (float*)c.u->persist.alloc(sizeof(float) * B * 64);
  *coef_out = coef; *mr_out = mr;
  if (y && gn_small_eligible(c.dt, HW, g.C)) {
    RUN(gn_fwd_small(c.st, c.dt, x, ldx, B, HW, g.C, g.gamma, g.beta, film, film_ld, 1e-5f, silu, y, ldy, coef, mr));
    return KDIP_OK;
  }
  const bool do_fold = fold && !y && g_gn_fold.load();
  double* stats = nullptr;
  const double* stats2 = nullptr; int mC1 = 0;          // do_fold: un-merged sums of the two producers of a concat
  auto it = c.u->fused_stats.find(std::make_pair(x, g.C));
  if (it != c.u->fused_stats.end()) stats = it->second;                      // accumulated by the producing conv
  if (!stats && ldx == g.C) {
    // x = [t1 | t2] written in place by two convs that both accumulated statistics: merge them
    for (auto lo = c.u->fused_stats.lower_bound(std::make_pair(x, 0)); lo != c.u->fused_stats.end() && lo->first.first == x; ++lo) {
      const int C1 = lo->first.second, C2 = g.C - C1;
      if (C2 <= 0 || !gn_merge_eligible(C1, C2)) continue;
      auto hi = c.u->fused_stats.find(std::make_pair((const void*)((const char*)x + c.es * C1), C2));
      if (hi == c.u->fused_stats.end()) continue;
      if (do_fold) {      // merged by the conv while it loads its table; the merged sums are still reserved, so that the workspace
        { int rc_ = finish_pending(c, lo->second, B); if (rc_) return rc_; rc_ = finish_pending(c, hi->second, B); if (rc_) return rc_; }
        (void)new_sums(c, B);   // plan of a batch does not depend on the state of the fold switch when it was made (kdip_debug_gn_fold)
        stats = lo->second; stats2 = hi->second; mC1 = C1; break;
      }
      stats = new_sums(c, B);
      { int rc_ = finish_pending(c, lo->second, B); if (rc_) return rc_; rc_ = finish_pending(c, hi->second, B); if (rc_) return rc_; }
      RUN(gn_merge_stats(c.st, lo->second, C1, hi->second, C2, B, stats));
      break;
    }
  }
  if (!stats) {
    stats = new_sums(c, B);
    DetWs dw;
    const bool dd = det_ws(c, det_gn_bytes(B, HW), dw);
    RUN(gn_stats(c.st, c.dt, x, ldx, B, HW, g.C, stats, 1, dd ? &dw : nullptr));
  }
  if (do_fold) {
    { int rc_ = finish_pending(c, stats, B); if (rc_) return rc_; }
    fold->fold_stats = stats; fold->fold_stats2 = stats2; fold->fold_C1 = mC1; fold->fold_gamma = g.gamma; fold->fold_beta = g.beta;
    fold->fold_film = film; fold->fold_film_ld = film_ld; fold->fold_HW = HW; fold->fold_eps = 1e-5f;
    fold->fold_coef_out = coef; fold->fold_mr_out = mr;
    return KDIP_OK;
  }
  {
    auto pd = c.u->pending_stats.find(stats);
    if (pd != c.u->pending_stats.end() && !do_fold) {      // finish pass + coefficients in one launch
      RUN(conv_stats_finish_coef(c.st, pd->second, B, stats, g.gamma, g.beta, film, film_ld, HW, g.C, 1e-5f, coef, mr));
      c.u->pending_stats.erase(pd);
    } else {
      { int rc_ = finish_pending(c, stats, B); if (rc_) return rc_; }
      RUN(gn_coef(c.st, stats, g.gamma, g.beta, film, B, HW, g.C, 1e-5f, coef, mr, film_ld));
    }
  }
  if (!y) return KDIP_OK;
  if (pool_W > 0) RUN(gn_apply_pool2(c.st, c.dt, x, ldx, coef, B, (int)(HW / pool_W), pool_W, g.C, silu, y, ldy, pool_x, g.C));
  else RUN(gn_apply(c.st, c.dt, x, ldx, coef, B, HW, g.C, silu, y, ldy));
  *coef_out = coef; *mr_out = mr;
  return KDIP_OK;
}

int gn_backward(Ctx& c, const void* x, long ldx, const void* dy, long lddy, const float* coef, const float* mr, int B,
                long HW, int C, int silu, const void* addend, long lda, void* dx, long lddx, double* fused_sums = nullptr,
                const void* addend2 = nullptr, long lda2 = 0, int half_lgW = -1) {
  bool dry = c.dry;
  if (half_lgW < 0 && gn_small_eligible(c.dt, HW, C)) {
    RUN(gn_bwd_small(c.st, c.dt, x, ldx, dy, lddy, coef, mr, B, HW, C, silu, addend, lda, dx, lddx, addend2, lda2));
    return KDIP_OK;
  }
  double* sums = fused_sums;
  if (!sums) {
    sums = new_sums(c, B);
    DetWs dw;
    const bool dd = det_ws(c, det_gn_bytes(B, HW), dw);
    RUN(gn_bwd_stats(c.st, c.dt, x, ldx, dy, lddy, coef, mr, B, HW, C, silu, sums, 1, half_lgW, dd ? &dw : nullptr));
  }
  RUN(gn_bwd_apply(c.st, c.dt, x, ldx, dy, lddy, coef, mr, sums, B, HW, C, silu, addend, lda, dx, lddx, addend2, lda2, half_lgW));
  return KDIP_OK;
}

#ifndef KDIP_CONV3
#define KDIP_CONV3 1        // 0: first-generation kernels everywhere (A/B builds)
#endif
// true iff this conv runs on the second-generation kernel (bf16, large maps) and can therefore take a fused GroupNorm transform
#ifndef KDIP_TF2_MAX_COUT
#define KDIP_TF2_MAX_COUT 256
#endif
#ifndef KDIP_CONV3_MIN_BLOCKS
#define KDIP_CONV3_MIN_BLOCKS 192     // its 256-pixel x 128-channel tiles must at least roughly fill the chip (the 32x32 level does not at batch 8)
#endif
// tf: staging transform the caller wants fused (0 none, 1 GroupNorm forward, 2 GroupNorm backward)
bool use_conv3(Ctx& c, const ConvW& w, int B, int H, int W, long ldx, long ldy, bool dgrad, int out_f32, int tf = 0) {
  if (!KDIP_CONV3 || out_f32) return false;
  const int cin = dgrad ? w.cin_pad_b : w.cin_pad, cout = dgrad ? w.cin : w.cout;
  if (!conv3_eligible(c.dt, w.ntaps, H, W, cin, cout, ldx, ldy)) return false;
  if ((long)B * (H / 8) * (W / 32) * (cout / 128) < KDIP_CONV3_MIN_BLOCKS) return false;
  if (tf && cin > conv3_tf_max_cin(tf)) return false;
  if ((long)H * W * ldx * 2 >= (1L << 31)) return false;      // 32-bit staging offsets (conv3_forward's launch precondition): fall back to conv.hip
  // the GroupNorm-backward transform (two tensors, two fmas per element since the producer stores dz) is re-done by every
  // 128-channel output block
  if (tf == 2 && cout > KDIP_TF2_MAX_COUT) return false;
  return true;
}

// in_ups / res_ups: x / res are half-resolution tensors read through a fused nearest x2 upsample (H, W = output size)
// tf_coef: the input is a GroupNorm INPUT and (a, b) [B][cin][2] are its coefficients: silu(a*x + b) is applied while the
// patch is staged (only when use_conv3())
int conv_f(Ctx& c, const ConvW& w, const void* x, long ldx, int B, int H, int W, void* y, long ldy, const void* res,
           long ldr, int out_f32, bool fuse_out_stats = false, int in_ups = 0, int res_ups = 0, const float* tf_coef = nullptr,
           const Conv3Fuse* fold = nullptr) {
  bool dry = c.dry;
  const bool stats_ok = fuse_out_stats && !out_f32 && conv_stats_eligible(H, W, w.cout) && !gn_small_eligible(c.dt, (long)H * W, w.cout);
  if (use_conv3(c, w, B, H, W, ldx, ldy, false, out_f32, tf_coef ? 1 : 0)) {
    Conv3Fuse fu;
    fu.in_ups = in_ups; fu.res_ups = res_ups;
    if (tf_coef) { fu.tf = 1; fu.tf_silu = 1; fu.tf_coef = tf_coef; }
    if (tf_coef && fold && fold->fold_stats) {      // the conv computes (and stores) the coefficients itself
      fu.fold_stats = fold->fold_stats; fu.fold_stats2 = fold->fold_stats2; fu.fold_C1 = fold->fold_C1; fu.fold_gamma = fold->fold_gamma;
      fu.fold_beta = fold->fold_beta; fu.fold_film = fold->fold_film; fu.fold_film_ld = fold->fold_film_ld; fu.fold_HW = fold->fold_HW;
      fu.fold_eps = fold->fold_eps; fu.fold_coef_out = fold->fold_coef_out; fu.fold_mr_out = fold->fold_mr_out;
    }
    if (stats_ok) {
      fu.st_mode = 1;
      fu.st_sums = new_sums(c, B);
      c.u->fused_stats[std::make_pair((const void*)y, w.cout)] = fu.st_sums;
    }
    RUN(conv3_forward(c.st, x, ldx, B, H, W, w.cin_pad, w.wf, w.bias, w.cout, y, ldy, res, ldr, &fu, w.cin));
    return KDIP_OK;
  }
  if (tf_coef && !conv_tf_eligible(c.cdt(), w.ntaps, H, W, w.cin_pad))
    return set_error(KDIP_ERR_STATE, "internal: fused GroupNorm staging requested for a conv no kernel can fuse it into");
  ConvStats stt;
  stt.in_ups = in_ups; stt.res_ups = res_ups;
  if (tf_coef) { stt.tf_coef = tf_coef; stt.tf_silu = 1; }
  DetWs dw;
  DetPending pend;
  if (stats_ok) {
    stt.mode = 1;
    stt.sums = new_sums(c, B);
    c.u->fused_stats[std::make_pair((const void*)y, w.cout)] = stt.sums;
    if (c.u->det) {      // the slab outlives this launch (persist arena, whatever the switch says: the workspace plan of a batch must not depend on it): the
      dw.slab_bytes = det_conv_bytes(B, H, W, w.cout); dw.slab = c.u->persist.alloc(dw.slab_bytes);      // consuming GroupNorm runs the finish pass with its coefficient kernel
      stt.det = &dw;
      if (g_defer_finish.load()) stt.defer = &pend;
    }
  }
  stt.sk_det = c.u->det ? 1 : 0;
  stt.x3_sat = c.u->x3_sat;
  stt.x3_lowpeak = c.peak_slot();
  RUN(conv_forward(c.st, c.cdt(), w.ntaps, x, ldx, B, H, W, w.cin_pad, c.wf(w), w.bias, w.cout, y, ldy, res, ldr, out_f32, 1.f, w.cin,
                   (stt.mode || in_ups || res_ups || tf_coef || stt.sk_det || stt.x3_sat) ? &stt : nullptr, c.u->sk_ws, c.u->sk_ws_floats));
  if (stt.defer) c.u->pending_stats[stt.sums] = pend;      // (dry runs: an empty descriptor keeps the control flow of the consumers identical)
  return KDIP_OK;
}
// input-gradient: x here is dL/d(out) with >= cin_pad_b channels available (zero padded when cout % 32 != 0)
// gn_*: when given, the GroupNorm-backward sums of the produced gradient (w.r.t. the GN whose input is
// gn_x) are accumulated in the epilogue; *sums_out receives the buffer (or nullptr if not eligible).
// tf2_*: g is dL/d(GroupNorm output) of the GroupNorm whose input is tf2_x2 and tf2_coef = (a, b, k0, k1) [B][cout][4]: the
// GroupNorm backward is applied while the patch is staged (only when use_conv3())
int conv_b(Ctx& c, const ConvW& w, const void* g, long ldg, int B, int H, int W, void* y, long ldy, const void* res,
           long ldr, int out_f32, const void* gn_x = nullptr, long gn_ldx = 0, const float* gn_coef = nullptr,
           const float* gn_mr = nullptr, int gn_silu = 0, double** sums_out = nullptr, const float* tf2_coef = nullptr,
           const void* tf2_x2 = nullptr, int* y_is_dz = nullptr, const Conv3Fuse* fold2 = nullptr, const ConvStats* gnb = nullptr, int tf2_silu = 0,
           bool dz_for_tf2 = false) {
  // gnb (fp32 storage, 1x1): mode-3 descriptor (gnb_* fields): y = dgrad + GroupNorm-backward apply (+ addend) in the conv's epilogue
  bool dry = c.dry;
  if (sums_out) *sums_out = nullptr;
  if (y_is_dz) *y_is_dz = 0;          // 1: y holds dz = dy * silu'(z) of the GroupNorm (gn_x, gn_coef): apply its backward with silu = 0
  const bool stats_ok = gn_x && sums_out && !out_f32 && ldy == w.cin && conv_stats_eligible(H, W, w.cin) && !gn_small_eligible(c.dt, (long)H * W, w.cin);
  if (use_conv3(c, w, B, H, W, ldg, ldy, true, out_f32, tf2_coef ? 2 : 0) && !res && (!stats_ok || (gn_silu && y_is_dz))) {
    Conv3Fuse fu;
    if (tf2_coef) { fu.tf = 2; fu.tf_silu = 1; fu.tf_coef = tf2_coef; fu.x2 = tf2_x2; fu.ldx2 = ldg; }
    if (tf2_coef && fold2 && fold2->fold_stats) {   // (a, b, k0, k1) computed by the conv from coef / mr / backward sums: no gn_bwd_coef launch
      fu.fold_stats = fold2->fold_stats; fu.fold_coef = fold2->fold_coef; fu.fold_mr = fold2->fold_mr; fu.fold_HW = fold2->fold_HW;
    }
    if (stats_ok) {
      fu.st_mode = 2; fu.st_silu = 1; fu.st_x = gn_x; fu.st_ldx = gn_ldx; fu.st_coef = gn_coef; fu.st_mr = gn_mr;
      fu.st_sums = new_sums(c, B);
      *sums_out = fu.st_sums;
      if (y_is_dz) *y_is_dz = 1;
    }
    RUN(conv3_forward(c.st, g, ldg, B, H, W, w.cin_pad_b, w.wb, nullptr, w.cin, y, ldy, nullptr, 0, &fu, w.cout));
    return KDIP_OK;
  }
  if (tf2_coef && !(conv_tf_eligible(c.cdt(), w.ntaps, H, W, w.cin_pad_b) && !res && !out_f32))
    return set_error(KDIP_ERR_STATE, "internal: fused GroupNorm-backward staging requested for a conv no kernel can fuse it into");
  ConvStats stt;
  if (tf2_coef) { stt.tf_coef = tf2_coef; stt.tf_mode = 2; stt.tf_x2 = tf2_x2; stt.tf_silu = tf2_silu; }      // split precision: A = a*dz - (k0 + k1*x2) while staging (conv.hip, TFM 2)
  // (split-precision mode) gradients have no natural scale: the fp16 window of the A operand follows max |cotangent| of this VJP
  // (one reduction per VJP) -- enough for networks whose backward gains keep the gradient tensors of one VJP within +-4 decades of
  // it.  kdip_unet_x3_window(u, 1): every dgrad launch takes its own power-of-two scale from a sampled max |g| of its input instead
  // (a word of the zeros arena, cleared with it at the start of the VJP; one ~4 us launch in front of each dgrad conv: +2.3 % per step)
  stt.x3_amax = c.u->x3_amax;
  if (c.u->x3_amax && c.u->x3_window_per_launch) {
    unsigned* aw = (unsigned*)c.u->zeros.alloc(sizeof(unsigned));
    RUN(amax_bits_sampled(c.st, (const float*)g, ((long)B * H * W - 1) * ldg + w.cin_pad_b, aw));      // (the span of a channel-slice view)
    stt.x3_amax = aw;
  }
  DetWs dw;
  if (stats_ok) {
    stt.mode = 2; stt.silu = gn_silu; stt.x = gn_x; stt.ldx = gn_ldx; stt.coef = gn_coef; stt.mr = gn_mr;
    stt.sums = new_sums(c, B);
    *sums_out = stt.sums;
    if (det_ws(c, det_conv_bytes(B, H, W, w.cin), dw)) stt.det = &dw;
    if ((g_store_dz == 2 || (g_store_dz == 1 && dz_for_tf2)) && gn_silu && y_is_dz && c.dt != DT_BF16) { stt.store_dz = 1; *y_is_dz = 1; }      // the sweep that takes the sums leaves dz behind
  }
  if (gnb) {
    if (stats_ok || res || out_f32 || c.dt == DT_BF16 || w.ntaps != 1)
      return set_error(KDIP_ERR_STATE, "internal: GroupNorm-backward epilogue requested for a conv that cannot take it");
    stt.mode = 3; stt.gnb_coef = gnb->gnb_coef; stt.gnb_dz = gnb->gnb_dz; stt.gnb_lddz = gnb->gnb_lddz; stt.gnb_x = gnb->gnb_x; stt.gnb_ldx = gnb->gnb_ldx;
    stt.gnb_add = gnb->gnb_add; stt.gnb_lda = gnb->gnb_lda; stt.gnb_silu = gnb->gnb_silu;
  }
  stt.sk_det = c.u->det ? 1 : 0;
  stt.x3_sat = c.u->x3_sat;
  stt.x3_lowpeak = c.peak_slot();
  RUN(conv_forward(c.st, c.cdt(), w.ntaps, g, ldg, B, H, W, w.cin_pad_b, c.wb(w), nullptr, w.cin, y, ldy, res, ldr, out_f32, 1.f, w.cout,
                   (stt.mode || stt.x3_amax || stt.sk_det || stt.tf_coef) ? &stt : nullptr, c.u->sk_ws, c.u->sk_ws_floats));
  return KDIP_OK;
}
}  // namespace

// upsample with scale (adjoint of avgpool needs 0.25): implemented via copy + scale in one kernel
template <typename T>
__global__ void upsample2s_kernel(const T* __restrict__ x, long ldx, int B, int H, int W, int VP, T* __restrict__ y,
                                  long ldy, float scale) {
  constexpr int EPV = TypeInfo<T>::EPV;
  const int Ho = H * 2, Wo = W * 2;
  long nvec = (long)B * Ho * Wo * VP;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
    int vi = (int)(v % VP);
    long op = v / VP;
    int ox = (int)(op % Wo);
    long t = op / Wo;
    int oy = (int)(t % Ho), b = (int)(t / Ho);
    float f[EPV];
    unpack16<T>(*(const uint4*)(x + (((long)b * H + oy / 2) * W + ox / 2) * ldx + (long)vi * EPV), f);
#pragma unroll
    for (int e = 0; e < EPV; ++e) f[e] *= scale;
    *(uint4*)(y + op * ldy + (long)vi * EPV) = pack16<T>(f);
  }
}
static int upsample2s(hipStream_t st, DType dt, const void* x, long ldx, int B, int H, int W, int C, void* y, long ldy,
                      float scale) {
  int VP = C / (dt == DT_BF16 ? 8 : 4);
  long nvec = (long)B * (H * 2) * (W * 2) * VP;
  long g = (nvec + 255) / 256; if (g > 16384) g = 16384; if (g < 1) g = 1;
  if (dt == DT_BF16)
    hipLaunchKernelGGL(upsample2s_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)x, ldx, B, H, W, VP, (bf16_t*)y, ldy, scale);
  else
    hipLaunchKernelGGL(upsample2s_kernel<float>, dim3((unsigned)g), dim3(256), 0, st, (const float*)x, ldx, B, H, W, VP, (float*)y, ldy, scale);
  KDIP_LAUNCH_CHECK();
  return KDIP_OK;
}

// --------------------------------------------------------------------------- forward ----
// dst/ldd: when given, the block output is written there (a channel slice of a later concat buffer) instead of a fresh tensor
static int res_forward(Ctx& c, Layer& L, const void* x, long ldx, int B, int& H, int& W, const float* film_all, void** outp,
                       void* dst = nullptr, long ldd = 0) {
  bool dry = c.dry;
  UNet* u = c.u;
  const size_t es = c.es;
  u->scratch.reset();
  L.sv.x = x; L.sv.ldx = ldx; L.sv.B = B; L.sv.H = H; L.sv.W = W;
  const long HW = (long)H * W;
  const bool fused_pool = L.mode == 1 && !gn_small_eligible(c.dt, HW, L.cin);   // GN + SiLU + both 2x2 pools in one pass over x
  int Ho = H, Wo = W, ups = 0;
  if (L.mode == 1) { Ho = H / 2; Wo = W / 2; }
  if (L.mode == 2) { Ho = H * 2; Wo = W * 2; ups = 1; }
  const long HWo = (long)Ho * Wo;
  void* o = dst ? dst : u->persist.alloc(es * B * HWo * L.cout);
  const long ldo = dst ? ldd : L.cout;
  // GroupNorm + SiLU applied inside the consuming conv's input staging (second-generation kernel): the activated tensors
  // h1 / h3 are never written.  Not for downsampling blocks (the pool sits between SiLU and conv1) and small maps.
  // (split-precision mode: the same fusion in conv.hip's staging, for maps the streaming GroupNorm kernels would otherwise handle)
  const bool x1 = conv_tf_eligible(c.cdt(), 9, Ho, Wo, L.c1.cin_pad) && !gn_small_eligible(c.dt, HW, L.cin);
  const bool x2 = conv_tf_eligible(c.cdt(), 9, Ho, Wo, L.c2.cin_pad) && !gn_small_eligible(c.dt, HWo, L.cout);
  const bool f1 = L.mode != 1 && (use_conv3(c, L.c1, B, Ho, Wo, ldx, L.cout, false, 0, 1) || x1) && (L.mode != 2 || !gn_small_eligible(c.dt, HW, L.cin));
  const bool f2 = use_conv3(c, L.c2, B, Ho, Wo, L.cout, ldo, false, 0, 1) || x2;
  void* h1 = (fused_pool || f1) ? nullptr : u->scratch.alloc(es * B * HW * L.cin);
  Conv3Fuse fold1, fold2;                              // GroupNorm-coefficient folds of conv1 / conv2 (filled when the apply is fused)
  if (!fused_pool) CK(gn_forward(c, x, ldx, B, HW, L.n1, nullptr, 1, h1, L.cin, &L.sv.coef1, &L.sv.mr1, 0, 0, nullptr, (f1 && c.dt == DT_BF16) ? &fold1 : nullptr));
  const void* cin_ptr = h1; const void* xs = x; long ldxs = ldx;
  if (L.mode == 1) {
    void* h1p = u->scratch.alloc(es * B * Ho * Wo * L.cin);
    void* xp = u->scratch.alloc(es * B * Ho * Wo * L.cin);
    if (fused_pool) {
      CK(gn_forward(c, x, ldx, B, HW, L.n1, nullptr, 1, h1p, L.cin, &L.sv.coef1, &L.sv.mr1, 0, W, xp));
    } else {
      RUN(avgpool2(c.st, c.dt, h1, L.cin, B, H, W, L.cin, h1p, L.cin, 0.25f));
      RUN(avgpool2(c.st, c.dt, x, ldx, B, H, W, L.cin, xp, L.cin, 0.25f));
    }
    cin_ptr = h1p; xs = xp; ldxs = L.cin;
  }
  // (mode 2: Upsample(use_conv=False) of both branches (unet.py:232-235) is folded into the consumers' reads: conv1 and the
  // skip path read the half-resolution tensors at (y >> 1, x >> 1); no 4x-sized copies are written or re-read)
  void* h2 = u->persist.alloc(es * B * HWo * L.cout);
  L.sv.h2 = h2;
  if (f1) CK(conv_f(c, L.c1, x, ldx, B, Ho, Wo, h2, L.cout, nullptr, 0, 0, true, ups, 0, L.sv.coef1, &fold1));
  else CK(conv_f(c, L.c1, cin_ptr, L.cin, B, Ho, Wo, h2, L.cout, nullptr, 0, 0, true, ups, 0));
  const float* film = film_all + L.emb_off;          // row b at film + b * emb_total
  void* h3 = f2 ? nullptr : u->scratch.alloc(es * B * HWo * L.cout);
  CK(gn_forward(c, h2, L.cout, B, HWo, L.n2, film, 1, h3, L.cout, &L.sv.coef2, &L.sv.mr2, u->emb_total, 0, nullptr, (f2 && c.dt == DT_BF16) ? &fold2 : nullptr));      // (the coefficient fold is conv3's)
  const void* S = xs; long ldS = ldxs;
  if (L.has_skip) {
    void* sk = u->scratch.alloc(es * B * HWo * L.cout);
    CK(conv_f(c, L.skip, xs, ldxs, B, Ho, Wo, sk, L.cout, nullptr, 0, 0, false, ups, 0));
    S = sk; ldS = L.cout;
  }
  const int rups = (ups && !L.has_skip) ? 1 : 0;
  if (f2) CK(conv_f(c, L.c2, h2, L.cout, B, Ho, Wo, o, ldo, S, ldS, 0, true, 0, rups, L.sv.coef2, &fold2));
  else CK(conv_f(c, L.c2, h3, L.cout, B, Ho, Wo, o, ldo, S, ldS, 0, true, 0, rups));
  *outp = o; H = Ho; W = Wo;
  return KDIP_OK;
}

#ifndef KDIP_FUSED_ATTN
#define KDIP_FUSED_ATTN 1     // bf16, head width 64, T % 64 == 0: csrc/attention.hip (0: the bgemm + softmax passes)
#endif
static void attn_gemms(const Layer& L, int B, int T, int hc, BGemm& qk, BGemm& pv, const void* qkv, void* S, const void* P, void* a) {
  const int C = L.cin, heads = L.heads;
  const size_t dummy = 0; (void)dummy;
  qk = BGemm();
  qk.A = qkv; qk.sam = 3 * C; qk.sak = 1; qk.sab1 = (long)T * 3 * C; qk.sab2 = 3 * hc;
  qk.Bm = nullptr; qk.sbk = 1; qk.sbn = 3 * C; qk.sbb1 = (long)T * 3 * C; qk.sbb2 = 3 * hc;
  qk.C = S; qk.scm = T; qk.scn = 1; qk.scb1 = (long)heads * T * T; qk.scb2 = (long)T * T;
  qk.M = T; qk.N = T; qk.K = hc; qk.nb1 = B; qk.nb2 = heads; qk.alpha = 1.f / sqrtf((float)hc); qk.c_f32 = 1; qk.a_f32 = 0;
  pv = BGemm();
  pv.A = P; pv.sam = T; pv.sak = 1; pv.sab1 = (long)heads * T * T; pv.sab2 = (long)T * T;
  pv.Bm = nullptr; pv.sbk = 3 * C; pv.sbn = 1; pv.sbb1 = (long)T * 3 * C; pv.sbb2 = 3 * hc;
  pv.C = a; pv.scm = C; pv.scn = 1; pv.scb1 = (long)T * C; pv.scb2 = hc;
  pv.M = T; pv.N = hc; pv.K = T; pv.nb1 = B; pv.nb2 = heads; pv.alpha = 1.f; pv.c_f32 = 0; pv.a_f32 = 0;
}

static int attn_forward(Ctx& c, Layer& L, const void* x, long ldx, int B, int H, int W, void** outp, void* dst = nullptr,
                        long ldd = 0) {
  bool dry = c.dry;
  UNet* u = c.u;
  const size_t es = c.es;
  u->scratch.reset();
  const int C = L.cin, T = H * W, hc = u->cfg.num_head_channels, heads = L.heads;
  L.sv.x = x; L.sv.ldx = ldx; L.sv.B = B; L.sv.H = H; L.sv.W = W;
  void* n = u->scratch.alloc(es * B * T * C);
  CK(gn_forward(c, x, ldx, B, T, L.norm, nullptr, 0, n, C, &L.sv.coef1, &L.sv.mr1));
  void* qkv = u->persist.alloc(es * (size_t)B * T * 3 * C);
  L.sv.qkv = qkv;
  CK(conv_f(c, L.qkv, n, C, B, H, W, qkv, 3 * C, nullptr, 0, 0));
  void* a;
  if (KDIP_FUSED_ATTN && attn_fused_eligible(c.dt, T, hc, 3 * C)) {
    // scores never reach HBM; the VJP recomputes the probabilities from the saved output + log-sum-exp
    a = u->persist.alloc(es * (size_t)B * T * C);
    float* lse = (float*)u->persist.alloc(sizeof(float) * (size_t)B * heads * T);
    void* vt = u->scratch.alloc(es * (size_t)B * T * C);
    L.sv.ao = a; L.sv.lse = lse; L.sv.P = nullptr;
    RUN(attn_fused_forward(c.st, qkv, 3 * C, B, T, heads, vt, a, C, lse));
  } else {
  float* S = (float*)u->scratch.alloc(sizeof(float) * (size_t)B * heads * T * T);
  void* P = u->persist.alloc(es * (size_t)B * heads * T * T);
  L.sv.P = P; L.sv.ao = nullptr;
  a = u->scratch.alloc(es * (size_t)B * T * C);
  BGemm qk, pv;
  attn_gemms(L, B, T, hc, qk, pv, qkv, S, P, a);
  qk.Bm = (const char*)qkv + es * hc;
  pv.Bm = (const char*)qkv + es * 2 * hc;
  RUN(bgemm(c.st, c.dt, qk));
  RUN(softmax_rows(c.st, c.dt, S, (long)B * heads * T, T, P));
  RUN(bgemm(c.st, c.dt, pv));
  }
  void* o = dst ? dst : u->persist.alloc(es * (size_t)B * T * C);
  CK(conv_f(c, L.proj, a, C, B, H, W, o, dst ? ldd : C, x, ldx, 0, true));
  *outp = o;
  return KDIP_OK;
}

int UNet::forward_impl(hipStream_t st, const float* x_nchw, const float* t, int B, float in_scale, float* out_nchw,
                       float* cov_nchw, float* feat_nchw) {
  Ctx c{this, st, dry, dt, esize()};
  const size_t es = esize();
  persist.reset(); scratch.reset(); zeros.reset(); fused_stats.clear(); pending_stats.clear();
  if (!dry && zeros.cap) KDIP_HIP_CHECK(hipMemsetAsync(zeros.base, 0, zeros.cap, st));
  {
    // split-K workspace: room for any 3x3 conv output (forward: cout, dgrad: cin channels) on a <= 16x16 map (the layers whose launches
    // cannot fill the chip).  Only the 3x3 convs split K (conv.hip, launch_cfg2): the 3 x cin channels of the attention qkv 1x1 convs
    // do not count (they used to: 1.5 x the workspace of the FFHQ architecture).
    int maxc = 0;
    auto upd = [&](const std::vector<Layer>& ls) { for (auto& L : ls) if (L.kind != 2) maxc = std::max(maxc, std::max(L.cin, L.cout)); };
    for (auto& b : inp) upd(b);
    upd(mid);
    for (auto& b : out) upd(b);
    if (det) {
      // deterministic modes: one slab per K split, plain stores, nothing to keep zeroed (persist arena: lives through the VJP too)
      sk_ws_floats = (long)B * 256 * maxc * KDIP_SPLITK_MAX;
      sk_ws = (float*)persist.alloc(sizeof(float) * sk_ws_floats);
    } else {
      sk_ws_floats = (long)B * 256 * maxc;
      sk_ws = (float*)zeros.alloc(sizeof(float) * sk_ws_floats);
    }
  }
  // fp16-headed pass: one word per conv launch for the largest operand it staged (cleared with the zeros arena above)
  x3_peaks = ccdt() == DT_F32H3 ? (unsigned*)zeros.alloc(sizeof(unsigned) * X3_MAX_PEAKS) : nullptr;
  x3_npeaks = 0;
  int H = cfg.image_size, W = cfg.image_size;
  const int mc = cfg.model_channels, ted = mc * 4;
  // timestep embedding MLP (fp32)
  float* temb = (float*)persist.alloc(sizeof(float) * B * mc);
  float* e1 = (float*)persist.alloc(sizeof(float) * B * ted);
  float* emb = (float*)persist.alloc(sizeof(float) * B * ted);
  RUN(timestep_embedding(st, t, B, mc, temb));
  RUN(linear_f32(st, temb, te0, B, 0, e1, 1));               // e1 = SiLU(te0(temb)): the activation is applied once, by the producer
  RUN(linear_f32(st, e1, te2, B, 0, emb, 1));                // emb is only ever consumed through SiLU (every ResBlock's emb_layers)
  float* film_all = (float*)persist.alloc(sizeof(float) * (size_t)B * emb_total);
  RUN(linear_f32(st, emb, emb_all, B, 0, film_all));       // every ResBlock's Linear(SiLU(emb)) in one launch
  // input: NCHW fp32 * c_in -> NHWC T, channels padded to 32
  void* xin = persist.alloc(es * (size_t)B * H * W * 32);
  if (tapfold) RUN(im2col3_nchw(st, x_nchw, B, cfg.in_channels, H, W, in_scale, (float*)xin, 32));      // [B,H,W,32]: 9 taps x 3 channels (+ 5 zeros): the image conv is a 1x1 conv over it
  else RUN(nchw_to_nhwc(st, dt, x_nchw, B, cfg.in_channels, H, W, in_scale, xin, 32, 32));
  hs_ptr.clear(); hs_C.clear(); cat_ptr.clear();
  std::vector<int> hs_H;
  const void* h = nullptr; long ldh = 0; int Ch = 0;
  // the last layer of a block writes straight into its slice of the concat buffer it feeds (dst/ldd): no
  // torch.cat copies (guided_diffusion/unet.py:659-662) -- decoder stage j reads cat = [h | hs[nhs-1-j]]
  auto run_layers = [&](std::vector<Layer>& ls, void* dst, long ldd) -> int {
    for (size_t li = 0; li < ls.size(); ++li) {
      Layer& L = ls[li];
      void* d = li + 1 == ls.size() ? dst : nullptr;
      void* o = nullptr;
      if (L.kind == 0) {
        scratch.reset();
        o = d ? d : persist.alloc(es * (size_t)B * H * W * L.cout);
        L.sv.B = B; L.sv.H = H; L.sv.W = W;
        CK(conv_f(c, tapfold ? in_k1 : L.conv, xin, 32, B, H, W, o, d ? ldd : L.cout, nullptr, 0, 0, true));
      } else if (L.kind == 1) {
        CK(res_forward(c, L, h, ldh, B, H, W, film_all, &o, d, ldd));
      } else {
        CK(attn_forward(c, L, h, ldh, B, H, W, &o, d, ldd));
      }
      h = o; ldh = d ? ldd : L.cout; Ch = L.cout;
    }
    return KDIP_OK;
  };
  const int nhs = (int)inp.size();
  KDIP_REQUIRE((int)out.size() == nhs, "internal: %zu decoder stages for %d skips", out.size(), nhs);
  std::vector<void*> cat(nhs, nullptr);
  std::vector<long> cat_ld(nhs, 0);
  std::vector<int> cat_Ch(nhs, 0);                     // channels of the decoder-side half
  for (int si = 0; si < nhs; ++si) {
    // resolution / channels of hs[si] = output of encoder block si
    int Hs = H, Cs = inp[si].back().cout;
    for (auto& L : inp[si]) if (L.kind == 1 && L.mode == 1) Hs /= 2;
    const int j = nhs - 1 - si;
    const int Chf = j == 0 ? mid.back().cout : out[j - 1].back().cout;
    cat[si] = persist.alloc(es * (size_t)B * Hs * Hs * (Chf + Cs));
    cat_ld[si] = Chf + Cs; cat_Ch[si] = Chf;
    CK(run_layers(inp[si], (char*)cat[si] + es * Chf, cat_ld[si]));
    KDIP_REQUIRE(H == Hs, "internal: skip resolution mismatch");
    hs_ptr.push_back(h); hs_C.push_back(Ch); hs_H.push_back(H);
  }
  CK(run_layers(mid, cat[nhs - 1], cat_ld[nhs - 1]));
  for (int j = 0; j < nhs; ++j) {
    const int si = nhs - 1 - j;
    KDIP_REQUIRE(hs_H[si] == H && Ch == cat_Ch[si], "internal: skip resolution / channel mismatch");
    cat_ptr.push_back(cat[si]);
    h = cat[si]; ldh = cat_ld[si]; Ch = (int)cat_ld[si];
    CK(run_layers(out[j], j + 1 < nhs ? cat[si - 1] : nullptr, j + 1 < nhs ? cat_ld[si - 1] : 0));
  }
  final_h = h;
  // head: GN -> SiLU -> conv3x3 (fp32 out, channels padded to 32)
  scratch.reset();
  const long HW = (long)H * W;
  void* hn = scratch.alloc(es * B * HW * final_ch);
  CK(gn_forward(c, h, ldh, B, HW, out_norm, nullptr, 1, hn, final_ch, &out_coef, &out_mr));
  if (tapfold) {      // per-tap partial products [B,H,W,9 x out_channels (+ padding)] from a 1x1 conv, then the nine shifted taps + bias -> NCHW
    float* pt = (float*)scratch.alloc(sizeof(float) * B * HW * out_n1.cout);
    CK(conv_f(c, out_n1, hn, final_ch, B, H, W, pt, out_n1.cout, nullptr, 0, 0));
    RUN(tap_gather_nchw(st, pt, out_n1.cout, B, cfg.out_channels, H, W, out_conv.bias, out_nchw));
  } else {
  float* o32 = (float*)scratch.alloc(sizeof(float) * B * HW * 32);
  CK(conv_f(c, out_conv, hn, final_ch, B, H, W, o32, 32, nullptr, 0, 1));
  RUN(nhwc_to_nchw_f32(st, o32, 32, B, cfg.out_channels, H, W, out_nchw));
  }
  if (cov_nchw) {
    KDIP_REQUIRE(has_cov, "out_cov weights were not loaded");
    float* c32 = (float*)scratch.alloc(sizeof(float) * B * HW * 32);
    CK(conv_f(c, cov_conv, h, ldh, B, H, W, c32, 32, nullptr, 0, 1));
    RUN(nhwc_to_nchw_f32(st, c32, 32, B, 6, H, W, cov_nchw));
  }
  if (feat_nchw) RUN(nhwc_T_to_nchw_f32(st, dt, h, ldh, B, final_ch, H, W, feat_nchw));
  if (x3_peaks) RUN(x3_lowpeak_check(st, x3_peaks, std::min(x3_npeaks, (int)X3_MAX_PEAKS), x3_sat));
  zeros_fwd_end = (zeros.off + 255) & ~(size_t)255;
  return KDIP_OK;
}

// -------------------------------------------------------------------------- backward ----
// add2/lda2: an extra gradient summed into the result (the concat-skip gradient of the tensor this block consumed)
static int res_backward(Ctx& c, Layer& L, const void* G, long ldG, void** gxp, const void* add2 = nullptr, long lda2 = 0) {
  bool dry = c.dry;
  UNet* u = c.u;
  const size_t es = c.es;
  u->scratch.reset();
  const int B = L.sv.B, H = L.sv.H, W = L.sv.W;
  int Ho = H, Wo = W;
  if (L.mode == 1) { Ho = H / 2; Wo = W / 2; }
  if (L.mode == 2) { Ho = H * 2; Wo = W * 2; }
  const long HW = (long)H * W, HWo = (long)Ho * Wo;
  // conv2 dgrad -> grad wrt h3 ; GN2/FiLM/SiLU backward -> grad wrt h2
  void* g3 = u->scratch.alloc(es * B * HWo * L.cout);
  double* sums2 = nullptr;     // GN2-backward sums accumulated by the dgrad epilogue when the shape allows
  int g3_dz = 0;               // g3 holds dz (the dgrad epilogue already applied silu') instead of dy
  const bool fbx_ok = g_x3_tf2 && is_x3(c.cdt()) && conv_tf_eligible(c.cdt(), 9, Ho, Wo, L.c1.cin_pad_b) && !gn_small_eligible(c.dt, HWo, L.cout);
  CK(conv_b(c, L.c2, G, ldG, B, Ho, Wo, g3, L.cout, nullptr, 0, 0, L.sv.h2, L.cout, L.sv.coef2, L.sv.mr2, 1, &sums2, nullptr, nullptr, &g3_dz, nullptr, nullptr, 0, fbx_ok));
  // conv1 dgrad -> grad wrt (resampled) h1.  Second-generation kernel + fused sums: the GN2 backward apply
  // (gh2 = a*dz - (k0 + k1*h2)) happens inside the dgrad conv's input staging; gh2 is never written.
  void* g1p = u->scratch.alloc(es * B * HWo * L.cin);
  double* sums1 = nullptr;
  const bool fbx = fbx_ok && sums2;      // split precision: GroupNorm-backward staging in conv.hip (g3 holds dz or dy: tf2_silu)
  const bool fb = (sums2 && g3_dz && use_conv3(c, L.c1, B, Ho, Wo, L.cout, L.cin, true, 0, 2)) || fbx;
  const void* gh2 = nullptr;
  float* tf2 = nullptr;
  Conv3Fuse foldb;
  if (fb) {
    tf2 = (float*)u->scratch.alloc(sizeof(float) * B * L.cout * 4);
    if (g_gn_fold.load() && !fbx) { foldb.fold_stats = sums2; foldb.fold_coef = L.sv.coef2; foldb.fold_mr = L.sv.mr2; foldb.fold_HW = HWo; }
    else RUN(gn_bwd_coef(c.st, L.sv.coef2, L.sv.mr2, sums2, B, HWo, L.cout, tf2));
  } else {
    void* t = u->scratch.alloc(es * B * HWo * L.cout);
    CK(gn_backward(c, L.sv.h2, L.cout, g3, L.cout, L.sv.coef2, L.sv.mr2, B, HWo, L.cout, g3_dz ? 0 : 1, nullptr, 0, t, L.cout, sums2));
    gh2 = t;
  }
  int g1_dz = 0;
  if (L.mode == 0)
    CK(conv_b(c, L.c1, fb ? g3 : gh2, L.cout, B, Ho, Wo, g1p, L.cin, nullptr, 0, 0, L.sv.x, L.sv.ldx, L.sv.coef1, L.sv.mr1, 1, &sums1, tf2, fb ? L.sv.h2 : nullptr, &g1_dz, &foldb, nullptr, (fbx && !g3_dz) ? 1 : 0));
  else
    CK(conv_b(c, L.c1, fb ? g3 : gh2, L.cout, B, Ho, Wo, g1p, L.cin, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, tf2, fb ? L.sv.h2 : nullptr, nullptr, &foldb, nullptr, (fbx && !g3_dz) ? 1 : 0));
  // Channel-changing block on a streaming-size map, fp32 storage: the input gradient  a*dz - (k0 + k1*x) + W_skip^T G (+ concat gradient)
  // is written by the skip dgrad conv's epilogue (ConvStats mode 3) -- no skip-gradient tensor, no gn_bwd_apply pass over 4 - 5 tensors
  if (g_gnb_fold && L.has_skip && L.mode == 0 && c.dt != DT_BF16 && sums1 && !gn_small_eligible(c.dt, HW, L.cin) && L.cin % 128 == 0 &&
      HW >= 128 && L.sv.ldx % 4 == 0 && lda2 % 4 == 0) {
    float* kc = (float*)u->scratch.alloc(sizeof(float) * B * L.cin * 4);
    RUN(gn_bwd_coef(c.st, L.sv.coef1, L.sv.mr1, sums1, B, HW, L.cin, kc));
    void* gx = u->persist.alloc(es * B * HW * L.cin);
    ConvStats gb;
    gb.gnb_coef = kc; gb.gnb_dz = g1p; gb.gnb_lddz = L.cin; gb.gnb_x = L.sv.x; gb.gnb_ldx = L.sv.ldx; gb.gnb_add = add2; gb.gnb_lda = lda2; gb.gnb_silu = g1_dz ? 0 : 1;
    CK(conv_b(c, L.skip, G, ldG, B, Ho, Wo, gx, L.cin, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, &gb));
    *gxp = gx;
    return KDIP_OK;
  }
  // skip path: grad wrt (resampled) x
  const void* gS = G; long ldgS = ldG;
  if (L.has_skip) {
    void* t = u->scratch.alloc(es * B * HWo * L.cin);
    CK(conv_b(c, L.skip, G, ldG, B, Ho, Wo, t, L.cin, nullptr, 0, 0));
    gS = t; ldgS = L.cin;
  }
  const void* g1 = g1p; const void* gxs = gS; long ldgxs = ldgS;
  int half_lgW = -1;
  if (L.mode == 1 && !gn_small_eligible(c.dt, HW, L.cin) && (W & (W - 1)) == 0 && W >= 2) {
    half_lgW = __builtin_ctz(W);   // the GroupNorm-backward kernels read the half-resolution gradients in place (x 1/4)
  } else if (L.mode == 1) {        // adjoint of 2x2 average pool: nearest upsample * 1/4
    void* a = u->scratch.alloc(es * B * HW * L.cin);
    void* b = u->scratch.alloc(es * B * HW * L.cin);
    RUN(upsample2s(c.st, c.dt, g1p, L.cin, B, Ho, Wo, L.cin, a, L.cin, 0.25f));
    RUN(upsample2s(c.st, c.dt, gS, ldgS, B, Ho, Wo, L.cin, b, L.cin, 0.25f));
    g1 = a; gxs = b; ldgxs = L.cin;
  } else if (L.mode == 2) { // adjoint of nearest x2: 2x2 block sum
    void* a = u->scratch.alloc(es * B * HW * L.cin);
    void* b = u->scratch.alloc(es * B * HW * L.cin);
    RUN(avgpool2(c.st, c.dt, g1p, L.cin, B, Ho, Wo, L.cin, a, L.cin, 1.f));
    RUN(avgpool2(c.st, c.dt, gS, ldgS, B, Ho, Wo, L.cin, b, L.cin, 1.f));
    g1 = a; gxs = b; ldgxs = L.cin;
  }
  void* gx = u->persist.alloc(es * B * HW * L.cin);
  CK(gn_backward(c, L.sv.x, L.sv.ldx, g1, L.cin, L.sv.coef1, L.sv.mr1, B, HW, L.cin, g1_dz ? 0 : 1, gxs, ldgxs, gx, L.cin, sums1, add2, lda2, half_lgW));
  *gxp = gx;
  return KDIP_OK;
}

static int attn_backward(Ctx& c, Layer& L, const void* G, long ldG, void** gxp, const void* add2 = nullptr, long lda2 = 0) {
  bool dry = c.dry;
  UNet* u = c.u;
  const size_t es = c.es;
  u->scratch.reset();
  const int B = L.sv.B, H = L.sv.H, W = L.sv.W, C = L.cin, T = H * W, hc = u->cfg.num_head_channels, heads = L.heads;
  void* ga = u->scratch.alloc(es * (size_t)B * T * C);
  CK(conv_b(c, L.proj, G, ldG, B, H, W, ga, C, nullptr, 0, 0));
  void* dqkv = u->scratch.alloc(es * (size_t)B * T * 3 * C);
  const char* qkv = (const char*)L.sv.qkv;
  if (KDIP_FUSED_ATTN && attn_fused_eligible(c.dt, T, hc, 3 * C)) {
    void* ws = u->scratch.alloc(es * (size_t)B * T * C * 3);
    float* D = (float*)u->scratch.alloc(sizeof(float) * (size_t)B * heads * T);
    RUN(attn_fused_backward(c.st, qkv, 3 * C, ga, C, L.sv.ao, C, L.sv.lse, B, T, heads, ws, D, dqkv, 3 * C));
  } else {
  float* dP = (float*)u->scratch.alloc(sizeof(float) * (size_t)B * heads * T * T);
  void* dS = u->scratch.alloc(es * (size_t)B * heads * T * T);
  const float alpha = 1.f / sqrtf((float)hc);
  BGemm g;
  // dP[t][s] = sum_d ga[t][d] V[s][d]
  g = BGemm(); g.A = ga; g.sam = C; g.sak = 1; g.sab1 = (long)T * C; g.sab2 = hc;
  g.Bm = qkv + es * 2 * hc; g.sbk = 1; g.sbn = 3 * C; g.sbb1 = (long)T * 3 * C; g.sbb2 = 3 * hc;
  g.C = dP; g.scm = T; g.scn = 1; g.scb1 = (long)heads * T * T; g.scb2 = (long)T * T;
  g.M = T; g.N = T; g.K = hc; g.nb1 = B; g.nb2 = heads; g.alpha = 1.f; g.c_f32 = 1;
  RUN(bgemm(c.st, c.dt, g));
  // dV[s][d] = sum_t P[t][s] ga[t][d]
  g = BGemm(); g.A = L.sv.P; g.sam = 1; g.sak = T; g.sab1 = (long)heads * T * T; g.sab2 = (long)T * T;
  g.Bm = ga; g.sbk = C; g.sbn = 1; g.sbb1 = (long)T * C; g.sbb2 = hc;
  g.C = (char*)dqkv + es * 2 * hc; g.scm = 3 * C; g.scn = 1; g.scb1 = (long)T * 3 * C; g.scb2 = 3 * hc;
  g.M = T; g.N = hc; g.K = T; g.nb1 = B; g.nb2 = heads; g.alpha = 1.f; g.c_f32 = 0;
  RUN(bgemm(c.st, c.dt, g));
  RUN(softmax_bwd_rows(c.st, c.dt, L.sv.P, dP, (long)B * heads * T, T, dS));
  // dQ[t][d] = alpha * sum_s dS[t][s] K[s][d]
  g = BGemm(); g.A = dS; g.sam = T; g.sak = 1; g.sab1 = (long)heads * T * T; g.sab2 = (long)T * T;
  g.Bm = qkv + es * hc; g.sbk = 3 * C; g.sbn = 1; g.sbb1 = (long)T * 3 * C; g.sbb2 = 3 * hc;
  g.C = dqkv; g.scm = 3 * C; g.scn = 1; g.scb1 = (long)T * 3 * C; g.scb2 = 3 * hc;
  g.M = T; g.N = hc; g.K = T; g.nb1 = B; g.nb2 = heads; g.alpha = alpha; g.c_f32 = 0;
  RUN(bgemm(c.st, c.dt, g));
  // dK[s][d] = alpha * sum_t dS[t][s] Q[t][d]
  g = BGemm(); g.A = dS; g.sam = 1; g.sak = T; g.sab1 = (long)heads * T * T; g.sab2 = (long)T * T;
  g.Bm = qkv; g.sbk = 3 * C; g.sbn = 1; g.sbb1 = (long)T * 3 * C; g.sbb2 = 3 * hc;
  g.C = (char*)dqkv + es * hc; g.scm = 3 * C; g.scn = 1; g.scb1 = (long)T * 3 * C; g.scb2 = 3 * hc;
  g.M = T; g.N = hc; g.K = T; g.nb1 = B; g.nb2 = heads; g.alpha = alpha; g.c_f32 = 0;
  RUN(bgemm(c.st, c.dt, g));
  }
  void* gn_in = u->scratch.alloc(es * (size_t)B * T * C);
  double* sumsn = nullptr;
  CK(conv_b(c, L.qkv, dqkv, 3 * C, B, H, W, gn_in, C, nullptr, 0, 0, L.sv.x, L.sv.ldx, L.sv.coef1, L.sv.mr1, 0, &sumsn));
  void* gx = u->persist.alloc(es * (size_t)B * T * C);
  CK(gn_backward(c, L.sv.x, L.sv.ldx, gn_in, C, L.sv.coef1, L.sv.mr1, B, T, C, 0, G, ldG, gx, C, sumsn, add2, lda2));
  *gxp = gx;
  return KDIP_OK;
}

int UNet::vjp_impl(hipStream_t st, const float* cot_nchw, float* gx_nchw) {
  Ctx c{this, st, dry, dt, esize()};
  const size_t es = esize();
  const int B = last_B;
  const int H0 = cfg.image_size, W0 = cfg.image_size;
  const long HW0 = (long)H0 * W0;
  scratch.reset();
  zeros.off = zeros_fwd_end;
  if (!dry && zeros.cap > zeros_fwd_end)
    KDIP_HIP_CHECK(hipMemsetAsync(zeros.base + zeros_fwd_end, 0, zeros.cap - zeros_fwd_end, st));
  x3_peaks = ccdt() == DT_F32H3 ? (unsigned*)zeros.alloc(sizeof(unsigned) * X3_MAX_PEAKS) : nullptr;
  x3_npeaks = 0;
  // cotangent NCHW fp32 [B,out_ch,H,W] -> NHWC T padded to 32 channels
  const int cotw = tapfold ? out_k1.cin_pad_b : 32;
  void* cot = persist.alloc(es * B * HW0 * cotw);
  if (tapfold) RUN(im2col3_nchw(st, cot_nchw, B, cfg.out_channels, H0, W0, 1.f, (float*)cot, cotw));      // taps folded into K: the head's dgrad is a 1x1 conv
  else RUN(nchw_to_nhwc(st, dt, cot_nchw, B, cfg.out_channels, H0, W0, 1.f, cot, 32, 32));
  if (x3_amax && !x3_window_per_launch) RUN(amax_bits(st, cot_nchw, (long)B * cfg.out_channels * HW0, x3_amax));
  void* ghn = scratch.alloc(es * B * HW0 * final_ch);
  double* sumsh = nullptr;
  int gh_dz = 0;
  CK(conv_b(c, tapfold ? out_k1 : out_conv, cot, cotw, B, H0, W0, ghn, final_ch, nullptr, 0, 0, final_h, final_ch, out_coef, out_mr, 1, &sumsh, nullptr, nullptr, &gh_dz));
  void* G = persist.alloc(es * B * HW0 * final_ch);
  CK(gn_backward(c, final_h, final_ch, ghn, final_ch, out_coef, out_mr, B, HW0, final_ch, gh_dz ? 0 : 1, nullptr, 0, G, final_ch, sumsh));
  const void* g = G; long ldg = final_ch;
  // add2: gradient summed into the block's input gradient by its first layer's last kernel (the concat-skip
  // gradient of the tensor the block consumed) -- replaces a separate add pass per skip connection
  auto back_layers = [&](std::vector<Layer>& ls, const void* add2 = nullptr, long lda2 = 0) -> int {
    for (int i = (int)ls.size() - 1; i >= 0; --i) {
      Layer& L = ls[i];
      void* gx = nullptr;
      const void* a2 = i == 0 ? add2 : nullptr;
      if (L.kind == 1) CK(res_backward(c, L, g, ldg, &gx, a2, lda2));
      else if (L.kind == 2) CK(attn_backward(c, L, g, ldg, &gx, a2, lda2));
      else return set_error(KDIP_ERR_STATE, "internal: conv layer inside block list");
      g = gx; ldg = L.cin;
    }
    return KDIP_OK;
  };
  const int nhs = (int)hs_ptr.size();
  std::vector<const void*> skip_g(nhs, nullptr);
  std::vector<long> skip_ld(nhs, 0);
  for (int j = (int)out.size() - 1; j >= 0; --j) {
    CK(back_layers(out[j]));
    // g is grad wrt cat [Ch + Cs]; split views
    int si = nhs - 1 - j;
    int Cs = hs_C[si];
    int Ctot = (int)ldg;
    int Chh = Ctot - Cs;
    skip_g[si] = (const char*)g + es * Chh;
    skip_ld[si] = Ctot;
    // g (first Chh channels, ld = Ctot) continues
  }
  // grad wrt hs[i] = chain gradient + concat-skip gradient: the skip part rides along as add2
  CK(back_layers(mid, skip_g[nhs - 1], skip_ld[nhs - 1]));
  for (int i = nhs - 1; i >= 1; --i) CK(back_layers(inp[i], skip_g[i - 1], skip_ld[i - 1]));
  // input_blocks.0: conv3x3(3->ch); g is the complete gradient wrt hs[0]: dgrad to the 3 input channels (fp32)
  {
    Layer& L0 = inp[0][0];
    const long np = (long)B * HW0;
    scratch.reset();
    if (tapfold) {      // taps folded into N: per-tap partial input-gradients from a 1x1 conv, gathered into the NCHW gradient
      float* pt = (float*)scratch.alloc(sizeof(float) * np * in_n1.cin);
      CK(conv_b(c, in_n1, g, ldg, B, H0, W0, pt, in_n1.cin, nullptr, 0, 0));
      RUN(tap_gather_nchw(st, pt, in_n1.cin, B, cfg.in_channels, H0, W0, nullptr, gx_nchw));
    } else {
    float* gx32 = (float*)scratch.alloc(sizeof(float) * np * 32);
    CK(conv_b(c, L0.conv, g, ldg, B, H0, W0, gx32, 32, nullptr, 0, 1));
    RUN(nhwc_to_nchw_f32(st, gx32, 32, B, cfg.in_channels, H0, W0, gx_nchw));
    }
  }
  if (x3_peaks) RUN(x3_lowpeak_check(st, x3_peaks, std::min(x3_npeaks, (int)X3_MAX_PEAKS), x3_sat));
  return KDIP_OK;
}

// ------------------------------------------------------------------------ workspace ----
int UNet::ensure_workspace(int B) {
  // Arena sizes come from a host-only dry run of forward + VJP AT THIS BATCH (tile choices, split-K eligibility and fused
  // statistics depend on B, so peak usage is not monotone in B): every batch is planned once, before any kernel of a call
  // at that batch is enqueued, and the arenas are regrown when a plan does not fit.  Arena::alloc therefore cannot run past
  // `cap` in a real pass; the post-pass checks in run() / vjp() stay as internal assertions.
  auto it = planned.find(B);
  if (it == planned.end()) {
    Arena sp = persist, ss = scratch, sz = zeros;
    persist = Arena(); scratch = Arena(); zeros = Arena();
    dry = true;
    float dummy = 0;
    int rc = forward_impl(nullptr, &dummy, &dummy, B, 1.f, &dummy, has_cov ? &dummy : nullptr, &dummy);
    if (!rc) { last_B = B; rc = vjp_impl(nullptr, &dummy, &dummy); }
    dry = false;
    have_stash = false;
    std::array<size_t, 3> pk = {persist.peak + (1 << 20), scratch.peak + (1 << 20), zeros.peak + 4096};
    persist = sp; scratch = ss; zeros = sz;
    if (rc) return rc;
    it = planned.emplace(B, pk).first;
  }
  const std::array<size_t, 3>& pk = it->second;
  if (pk[0] <= persist.cap && pk[1] <= scratch.cap && pk[2] <= zeros.cap) return KDIP_OK;
  KDIP_HIP_CHECK(hipSetDevice(device));
  KDIP_HIP_CHECK(hipDeviceSynchronize());           // kernels of earlier calls may still use the old arenas
  auto grow = [&](Arena& a, size_t need, const char* what) -> int {
    if (need <= a.cap) return KDIP_OK;
    if (a.base) { KDIP_HIP_CHECK(hipFree(a.base)); }
    a = Arena();
    if (hipMalloc((void**)&a.base, need) != hipSuccess) return set_error(KDIP_ERR_NOMEM, "workspace (%s): hipMalloc(%zu) failed", what, need);
    a.cap = need;
    ++ws_generation;               // every pointer handed out from the old arena is dangling now (graphs.py re-captures)
    return KDIP_OK;
  };
  CK(grow(zeros, pk[2], "zeros"));
  CK(grow(persist, pk[0], "persist"));
  CK(grow(scratch, pk[1], "scratch"));
  if (B > ws_B) ws_B = B;
  return KDIP_OK;
}

int UNet::run(hipStream_t st, const float* x_nchw, const float* t, int B, float in_scale, float* out_nchw,
              float* cov_nchw, float* feat_nchw, int save) {
  KDIP_REQUIRE(finalized, "unet_forward before finalize");
  KDIP_REQUIRE(B >= 1, "batch must be >= 1");
  CK(ensure_workspace(B));
  have_stash = false;
  CK(forward_impl(st, x_nchw, t, B, in_scale, out_nchw, cov_nchw, feat_nchw));
  if (persist.overflow || scratch.overflow || zeros.overflow || persist.peak > persist.cap || scratch.peak > scratch.cap || zeros.peak > zeros.cap)
    return set_error(KDIP_ERR_STATE, "internal: workspace arena overflow (persist %zu/%zu, scratch %zu/%zu, zeros %zu/%zu)", persist.peak,
                     persist.cap, scratch.peak, scratch.cap, zeros.peak, zeros.cap);
  last_B = B;
  have_stash = true;
  (void)save;
  return KDIP_OK;
}

int UNet::vjp(hipStream_t st, const float* cot_nchw, float* gx_nchw) {
  if (!have_stash) return set_error(KDIP_ERR_STATE, "unet_vjp without a preceding unet_forward");
  size_t mark = persist.off;
  int rc = vjp_impl(st, cot_nchw, gx_nchw);
  if (!rc && (persist.overflow || scratch.overflow || zeros.overflow || persist.peak > persist.cap || scratch.peak > scratch.cap || zeros.peak > zeros.cap))
    rc = set_error(KDIP_ERR_STATE, "internal: workspace arena overflow in the VJP");
  persist.off = mark;   // the stash stays valid: the VJP may be called again (tmpd / STSL style)
  return rc;
}

}  // namespace kdip
