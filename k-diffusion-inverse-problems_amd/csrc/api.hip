// C ABI of libkdip_hip (see include/kdip.h for the contract and reference citations).
#include <stdlib.h>
#include <vector>
#include <mutex>
#include "../../include/kdip_internal.h"
#include "kernels.h"
#include "fftops.h"
#include "opctx.h"
#include "unet.h"

using namespace kdip;

struct kdip_unet { UNet u; };
struct kdip_op { OpCtx c; };

#define API_CK(call) do { int _rc = (call); if (_rc) return _rc; } while (0)
namespace {
// device buffers of a test hook: freed on every return path
struct DevPool {
  std::vector<void*> bufs;
  ~DevPool() { for (void* b : bufs) (void)hipFree(b); }
  int get(void** out, size_t bytes) {
    if (hipMalloc(out, bytes) != hipSuccess) return kdip::set_error(KDIP_ERR_NOMEM, "test hook: hipMalloc(%zu) failed", bytes);
    bufs.push_back(*out);
    return KDIP_OK;
  }
};
}  // namespace
#define ST(s) ((hipStream_t)(s))

extern "C" {

const char* kdip_last_error(void) { return g_last_error.c_str(); }
int kdip_version(void) { return 100; }

// ------------------------------------------------------------------------------ UNet ----
int kdip_unet_create(int device, int dtype, int image_size, int in_channels, int model_channels, int out_channels,
                     int num_res_blocks, const int* attention_ds, int n_attention_ds, const int* channel_mult,
                     int n_channel_mult, int num_head_channels, kdip_unet** out) {
  KDIP_REQUIRE(out, "null output handle");
  KDIP_REQUIRE(dtype == KDIP_F32 || dtype == KDIP_BF16 || dtype == KDIP_BF16X3 || dtype == KDIP_F16X3, "dtype %d", dtype);
  KDIP_REQUIRE(in_channels == 3 && out_channels <= 32, "in_channels must be 3, out_channels <= 32");
  KDIP_REQUIRE(model_channels % 32 == 0, "model_channels must be a multiple of 32");
  int ndev = 0;
  KDIP_HIP_CHECK(hipGetDeviceCount(&ndev));
  KDIP_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
  kdip_unet* h = new kdip_unet();
  h->u.device = device;
  h->u.dt = dtype == KDIP_BF16 ? DT_BF16 : DT_F32;
  h->u.cdt = dtype == KDIP_BF16X3 ? DT_F32X3 : (dtype == KDIP_F16X3 ? DT_F32H3 : h->u.dt);
  h->u.has_alt = dtype == KDIP_F16X3;      // ... carries the bf16-headed weights too (kdip_unet_x3_head)
  h->u.det = dtype != KDIP_BF16;        // fp32-storage modes: fixed-order reductions (det.h)
  if (const char* e = getenv("KDIP_DET")) { if (atoi(e) == 0) h->u.det = false; }      // A/B timing aid (same as kdip_unet_deterministic(u, 0))
  h->u.cfg.image_size = image_size; h->u.cfg.in_channels = in_channels; h->u.cfg.model_channels = model_channels;
  h->u.cfg.out_channels = out_channels; h->u.cfg.num_res_blocks = num_res_blocks;
  h->u.cfg.attention_ds.assign(attention_ds, attention_ds + n_attention_ds);
  h->u.cfg.channel_mult.assign(channel_mult, channel_mult + n_channel_mult);
  h->u.cfg.num_head_channels = num_head_channels;
  int rc = h->u.build_plan();
  if (rc) { delete h; return rc; }
  *out = h;
  return KDIP_OK;
}
void kdip_unet_destroy(kdip_unet* u) { delete u; }
int kdip_unet_load(kdip_unet* u, const char* name, const float* data_host, const long* shape, int ndim) {
  KDIP_REQUIRE(u && name && data_host && shape, "null argument");
  return u->u.load(name, data_host, shape, ndim);
}
int kdip_unet_finalize(kdip_unet* u) { KDIP_REQUIRE(u, "null handle"); return u->u.finalize(); }
int kdip_unet_forward(kdip_unet* u, void* stream, const float* x_dev, const float* t_dev, int B, float in_scale,
                      float* out_dev, float* cov_out_dev, float* feature_dev) {
  KDIP_REQUIRE(u && x_dev && t_dev && out_dev, "null argument");
  KDIP_HIP_CHECK(hipSetDevice(u->u.device));
  return u->u.run(ST(stream), x_dev, t_dev, B, in_scale, out_dev, cov_out_dev, feature_dev, 1);
}
int kdip_unet_vjp(kdip_unet* u, void* stream, const float* cot_dev, int B, float* gx_dev) {
  KDIP_REQUIRE(u && cot_dev && gx_dev, "null argument");
  KDIP_REQUIRE(B == u->u.last_B, "unet_vjp: cotangent batch %d does not match the batch %d of the last forward", B, u->u.last_B);
  KDIP_HIP_CHECK(hipSetDevice(u->u.device));
  return u->u.vjp(ST(stream), cot_dev, gx_dev);
}
namespace {
__global__ void checksum_kernel(const unsigned* __restrict__ w, size_t n, unsigned long long* out) {
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += w[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
}  // namespace
// Debug / test aid: word sum of the activation stash the VJP reads (persist arena up to the end of the last forward).  Tests use it
// to assert that nothing between unet_forward and unet_vjp (solver, operators, host code) writes into the handle's workspace.
int kdip_unet_debug_stash_checksum(kdip_unet* u, void* stream, unsigned long long* sum_host) {
  KDIP_REQUIRE(u && sum_host, "null argument");
  KDIP_REQUIRE(u->u.have_stash, "no forward stash");
  KDIP_HIP_CHECK(hipSetDevice(u->u.device));
  unsigned long long* d = nullptr;
  KDIP_HIP_CHECK(hipMalloc((void**)&d, 8));
  hipError_t e = hipMemsetAsync(d, 0, 8, ST(stream));
  const size_t n = u->u.persist.off / 4;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(checksum_kernel, dim3(2048), dim3(256), 0, ST(stream), (const unsigned*)u->u.persist.base, n, d);
    e = hipMemcpyAsync(sum_host, d, 8, hipMemcpyDeviceToHost, ST(stream));
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ST(stream));
  (void)hipFree(d);                       // (freed on every path)
  KDIP_HIP_CHECK(e);
  return KDIP_OK;
}
long kdip_unet_workspace_generation(kdip_unet* u) { return u ? u->u.ws_generation : -1; }
int kdip_unet_x3_window(kdip_unet* u, int per_launch) {
  KDIP_REQUIRE(u, "null handle");
  KDIP_REQUIRE(is_x3(u->u.cdt), "x3_window: the handle was not created with KDIP_BF16X3 / KDIP_F16X3");
  if (u->u.x3_window_per_launch != (per_launch ? 1 : 0)) {
    u->u.x3_window_per_launch = per_launch ? 1 : 0;
    u->u.planned.clear();              // the per-launch words live in the zeros arena: re-plan every batch
    u->u.have_stash = false;           // ... and a VJP needs a forward made under the new plan
    ++u->u.ws_generation;              // graphs captured under the other window mode replay its kernels / arena offsets: invalidate them (graphs.py keys on this)
  }
  return KDIP_OK;
}
int kdip_unet_x3_saturated(kdip_unet* u, void* stream, int reset, int* flags_host) {
  KDIP_REQUIRE(u && flags_host, "null argument");
  KDIP_REQUIRE(is_x3(u->u.cdt) && u->u.x3_sat, "x3_saturated: the handle was not created with KDIP_BF16X3 / KDIP_F16X3 (or is not finalized)");
  KDIP_HIP_CHECK(hipSetDevice(u->u.device));
  unsigned w = 0;
  KDIP_HIP_CHECK(hipMemcpyAsync(&w, u->u.x3_sat, sizeof(w), hipMemcpyDeviceToHost, ST(stream)));
  if (reset) KDIP_HIP_CHECK(hipMemsetAsync(u->u.x3_sat, 0, sizeof(unsigned), ST(stream)));
  KDIP_HIP_CHECK(hipStreamSynchronize(ST(stream)));
  *flags_host = (int)(w & 5u) | (u->u.x3_weight_sat > 0 ? 2 : 0) | (u->u.x3_force_alt ? 8 : 0);
  return KDIP_OK;
}
int kdip_debug_x3_peaks(kdip_unet* u, void* stream, float* peaks_host, int max, int* n_host) {
  KDIP_REQUIRE(u && peaks_host && n_host, "null argument");
  const int n = u->u.x3_peaks ? std::min(std::min(u->u.x3_npeaks, (int)UNet::X3_MAX_PEAKS), max) : 0;
  *n_host = n;
  if (n > 0) {
    KDIP_HIP_CHECK(hipMemcpyAsync(peaks_host, u->u.x3_peaks, sizeof(float) * n, hipMemcpyDeviceToHost, ST(stream)));
    KDIP_HIP_CHECK(hipStreamSynchronize(ST(stream)));
  }
  return KDIP_OK;
}
int kdip_unet_x3_head(kdip_unet* u, int bf16_head) {
  if (!u) return set_error(KDIP_ERR_ARG, "null handle");
  if (!u->u.has_alt) return set_error(KDIP_ERR_STATE, "x3_head: the handle was not created with KDIP_F16X3");
  const int prev = u->u.x3_alt ? 1 : 0;
  u->u.x3_alt = bf16_head != 0;
  return prev;
}
int kdip_unet_deterministic(kdip_unet* u, int on) {
  KDIP_REQUIRE(u, "null handle");
  KDIP_REQUIRE(u->u.dt != DT_BF16 || !on, "deterministic reductions are instantiated for the fp32-storage modes (KDIP_F32, KDIP_BF16X3) only");
  const int prev = u->u.det ? 1 : 0;
  if (prev != (on ? 1 : 0)) {
    u->u.det = on != 0;
    u->u.planned.clear();              // slabs / counters / split-K workspace move between arenas: re-plan every batch
    u->u.have_stash = false;
    ++u->u.ws_generation;              // captured hipGraphs replay the old mode's kernels and offsets: invalidate them
  }
  return prev;
}
long kdip_unet_workspace_bytes(kdip_unet* u, int B) {
  if (!u) return -1;
  if (B > 0 && u->u.finalized) { int rc = u->u.ensure_workspace(B); if (rc) return rc; }
  return (long)(u->u.persist.cap + u->u.scratch.cap);
}

// ------------------------------------------------------------------ CU-masked streams (chip partitioning) ----
// Two part-batches of a GPU's batch can each be given their own set of compute units: the persistent conv kernels of one stream
// then never lock the other stream's latency-bound small-map kernels out of the chip (every CU's LDS is full while they run).
int kdip_stream_create_cu_mask(int device, const unsigned* mask_words, int nwords, void** stream_out) {
  KDIP_REQUIRE(mask_words && nwords > 0 && stream_out, "null argument");
  KDIP_HIP_CHECK(hipSetDevice(device));
  hipStream_t st = nullptr;
  KDIP_HIP_CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask_words));
  // measured on MI355X (tools/cu_census.py): each group of 8 mask bits addresses ONE CU index in all 8 XCDs (a CU index is
  // enabled when any of its 8 bits is set), so the stream owns 8 CUs per non-zero byte
  int ncus = 0;
  for (int w = 0; w < nwords; ++w)
    for (int b = 0; b < 4; ++b) ncus += ((mask_words[w] >> (8 * b)) & 0xffu) ? 8 : 0;
  stream_register_cus(st, ncus);
  *stream_out = (void*)st;
  return KDIP_OK;
}
int kdip_stream_destroy(void* stream) {
  stream_register_cus(ST(stream), 0);
  if (stream) KDIP_HIP_CHECK(hipStreamDestroy(ST(stream)));
  return KDIP_OK;
}
namespace {
__global__ void cu_census_kernel(unsigned* out, int spin) {
  // one record per block: HW_ID (wave / simd / cu / sh / se ids) and XCC_ID
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);          // HW_REG_HW_ID, all 32 bits
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);      // HW_REG_XCC_ID, bits 3:0
  }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);                     // keep the block resident so that others spread out
}
}  // namespace
// Debug aid: launches `blocks` one-wave blocks that each hold a 64 KB LDS allocation for a moment and records where they ran:
// out_host[2 b] = HW_ID, out_host[2 b + 1] = XCC_ID of block b.  Used to learn which CU-mask bit addresses which XCD / CU.
int kdip_debug_cu_census(void* stream, int blocks, unsigned* out_host) {
  KDIP_REQUIRE(blocks > 0 && blocks <= 4096 && out_host, "bad argument");
  unsigned* d = nullptr;
  KDIP_HIP_CHECK(hipMalloc((void**)&d, sizeof(unsigned) * 2 * blocks));
  hipLaunchKernelGGL(cu_census_kernel, dim3(blocks), dim3(64), 65536, ST(stream), d, 200);
  hipError_t e = hipMemcpyAsync(out_host, d, sizeof(unsigned) * 2 * blocks, hipMemcpyDeviceToHost, ST(stream));
  if (e == hipSuccess) e = hipStreamSynchronize(ST(stream));
  (void)hipFree(d);
  KDIP_HIP_CHECK(e);
  return KDIP_OK;
}

// -------------------------------------------------------------------------- operators ----
int kdip_op_create(int device, int kind, int image_size, int scale_factor, float sigma_s, kdip_op** out) {
  KDIP_REQUIRE(out, "null output handle");
  KDIP_REQUIRE(kind >= 0 && kind <= 2, "operator kind %d", kind);
  int ndev = 0;
  KDIP_HIP_CHECK(hipGetDeviceCount(&ndev));
  KDIP_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
  kdip_op* h = new kdip_op();
  h->c.device = device; h->c.kind = kind; h->c.N = image_size; h->c.sf = kind == OP_SR ? scale_factor : 1;
  h->c.sigma_s = sigma_s;
  int rc = h->c.init();
  if (rc) { delete h; return rc; }
  *out = h;
  return KDIP_OK;
}
void kdip_op_destroy(kdip_op* op) { delete op; }
int kdip_op_set_psf(kdip_op* op, const float* psf_host, int kh, int kw) { KDIP_REQUIRE(op && psf_host, "null argument"); return op->c.set_psf(psf_host, kh, kw); }
int kdip_op_set_separable(kdip_op* op, const float* kr, const float* kc, int taps) { KDIP_REQUIRE(op && kr && kc, "null argument"); return op->c.set_separable(kr, kc, taps); }
int kdip_op_set_mask(kdip_op* op, const float* mask_host) { KDIP_REQUIRE(op && mask_host, "null argument"); return op->c.set_mask(mask_host); }
int kdip_op_set_ortho(kdip_op* op, int t) { KDIP_REQUIRE(op, "null handle"); return op->c.set_ortho(t); }
int kdip_op_get_otf(kdip_op* op, void* stream, float* otf_dev) {
  KDIP_REQUIRE(op && otf_dev && op->c.FB, "no OTF");
  KDIP_HIP_CHECK(hipMemcpyAsync(otf_dev, op->c.FB, sizeof(float2) * op->c.N * op->c.N, hipMemcpyDeviceToDevice, ST(stream)));
  return KDIP_OK;
}
int kdip_op_apply(kdip_op* op, void* stream, const float* x_dev, int B, int adjoint, float* out_dev) {
  KDIP_REQUIRE(op && x_dev && out_dev, "null argument");
  OpCtx& c = op->c;
  KDIP_HIP_CHECK(hipSetDevice(c.device));
  const long nn = (long)c.N * c.N;
  if (c.kind == OP_INPAINT) { KDIP_REQUIRE(c.mask, "no mask"); return mul_planes(ST(stream), x_dev, c.mask, 3L * B * nn, 3 * nn, out_dev); }
  if (c.kind == OP_BLUR) return c.apply_A(ST(stream), x_dev, out_dev, B, adjoint);
  if (adjoint) return c.sr_transpose(ST(stream), x_dev, out_dev, B);
  API_CK(c.ensure_ws(B));
  API_CK(c.apply_A(ST(stream), x_dev, c.rbuf[5], B, 0));
  return strided_down(ST(stream), c.rbuf[5], c.N, c.sf, 3L * B, out_dev);
}
int kdip_op_solve(kdip_op* op, void* stream, const float* y_dev, const float* x0_dev, float var_scalar,
                  const float* var_tensor_dev, int B, float* mat_dev, int* cg_iters_host, int* cg_info_host) {
  KDIP_REQUIRE(op && y_dev && x0_dev && mat_dev, "null argument");
  KDIP_HIP_CHECK(hipSetDevice(op->c.device));
  return op->c.solve(ST(stream), y_dev, x0_dev, var_scalar, var_tensor_dev, B, mat_dev, cg_iters_host, cg_info_host);
}
int kdip_op_set_cg_fixed_trips(kdip_op* op, int trips) {
  KDIP_REQUIRE(op && trips >= 0 && trips <= 1000, "cg fixed trips must be in [0, 1000]");
  op->c.cg_fixed_trips = trips;
  return KDIP_OK;
}
int kdip_op_cg_unconverged(kdip_op* op, void* stream, int* count_host) {
  KDIP_REQUIRE(op && count_host, "null argument");
  *count_host = 0;
  if (!op->c.cg.unconverged) return KDIP_OK;            // no CG workspace yet: nothing ran
  KDIP_HIP_CHECK(hipSetDevice(op->c.device));
  KDIP_HIP_CHECK(hipMemcpyAsync(count_host, op->c.cg.unconverged, sizeof(int), hipMemcpyDeviceToHost, ST(stream)));
  KDIP_HIP_CHECK(hipMemsetAsync(op->c.cg.unconverged, 0, sizeof(int), ST(stream)));
  KDIP_HIP_CHECK(hipStreamSynchronize(ST(stream)));
  return KDIP_OK;
}

// One Type-I guided denoiser call in one entry point (the ~10 C calls of kdip_amd.condition._type_I_guidance_impl, same kernels in
// the same order): UNet forward, p_mean_variance epilogue, mat-solver, cotangent, UNet VJP, likelihood-score assembly, combine.
long kdip_guided_ws_floats(int B, int S) { return 33L * B * S * S; }
// workspace regions of kdip_guided_call_v1, in floats from ws_dev (KDIP_GWS_* order); the caller's stepwise continuation (a second
// VJP for tmpd / STSL after a fused call) reads x0_raw from here instead of knowing the layout
int kdip_guided_ws_layout(int B, int S, long* offsets, int count) {
  KDIP_REQUIRE(offsets && count == KDIP_GWS_COUNT && B >= 1 && S >= 1, "guided_ws_layout: bad arguments (count must be KDIP_GWS_COUNT = %d)", KDIP_GWS_COUNT);
  const long n3 = 3L * B * S * S;
  offsets[KDIP_GWS_OUT6] = 0;          offsets[KDIP_GWS_X0_MEAN] = 2 * n3;  offsets[KDIP_GWS_X0_RAW] = 3 * n3;
  offsets[KDIP_GWS_VAR] = 4 * n3;      offsets[KDIP_GWS_MAT] = 5 * n3;      offsets[KDIP_GWS_COT] = 6 * n3;
  offsets[KDIP_GWS_G_RAW] = 8 * n3;    offsets[KDIP_GWS_UG] = 9 * n3;       offsets[KDIP_GWS_SCORE] = 10 * n3;
  return KDIP_OK;
}
long kdip_op_workspace_generation(kdip_op* op) { return op ? op->c.ws_generation : -1; }
int kdip_guided_call_v1(kdip_unet* u, kdip_op* op, void* stream, const float* x_dev, const float* t_dev, const float* y_dev, int B,
                        const float* t7, float sigma, float var_scalar, int tensor_var, float* ws, float* hat_dev,
                        int* cg_iters_host, int* cg_info_host) {
  KDIP_REQUIRE(u && op && x_dev && t_dev && y_dev && t7 && ws && hat_dev, "null argument");
  KDIP_REQUIRE(u->u.cfg.out_channels == 6 && u->u.cfg.in_channels == 3, "guided_call_v1: the V1 path needs a learn-sigma UNet (3 -> 6 channels)");
  KDIP_REQUIRE(op->c.N == u->u.cfg.image_size, "guided_call_v1: operator size %d != UNet image size %d", op->c.N, u->u.cfg.image_size);
  hipStream_t st = ST(stream);
  const int S = u->u.cfg.image_size;
  const long HW = (long)S * S, n3 = 3L * B * HW;
  long off[KDIP_GWS_COUNT];
  API_CK(kdip_guided_ws_layout(B, S, off, KDIP_GWS_COUNT));       // the one definition of the layout
  float* out6 = ws + off[KDIP_GWS_OUT6];          // [B,6,S,S]
  float* x0_mean = ws + off[KDIP_GWS_X0_MEAN];    // [B,3,S,S] each
  float* x0_raw = ws + off[KDIP_GWS_X0_RAW];
  float* var = ws + off[KDIP_GWS_VAR];
  float* mat = ws + off[KDIP_GWS_MAT];
  float* cot = ws + off[KDIP_GWS_COT];            // [B,6,S,S]
  float* g_raw = ws + off[KDIP_GWS_G_RAW];
  float* ug = ws + off[KDIP_GWS_UG];
  float* score = ws + off[KDIP_GWS_SCORE];
  KDIP_HIP_CHECK(hipSetDevice(u->u.device));
  API_CK(u->u.run(st, x_dev, t_dev, B, t7[0], out6, nullptr, nullptr, 1));
  API_CK(kdip_x0_epilogue_v1(stream, out6, x_dev, B, HW, t7, x0_mean, x0_raw, tensor_var ? var : nullptr));
  API_CK(op->c.solve(st, y_dev, x0_mean, var_scalar, tensor_var ? var : nullptr, B, mat, cg_iters_host, cg_info_host));
  API_CK(vjp_cotangent_v1(st, mat, x0_raw, B, HW, t7[2], cot, g_raw));
  API_CK(u->u.vjp(st, cot, ug));
  API_CK(axpby(st, g_raw, t7[0] * t7[1], ug, t7[0], n3, score));                     // c_in * (a_t * g_raw + unet_vjp)
  return guidance_combine(st, x0_mean, score, 1.f, nullptr, 0.f, sigma * sigma, n3, hat_dev);
}

int kdip_op_ortho(kdip_op* op, void* stream, const float* x_dev, int B, int inverse, float* out_dev) {
  KDIP_REQUIRE(op && x_dev && out_dev, "null argument");
  KDIP_HIP_CHECK(hipSetDevice(op->c.device));
  return inverse ? op->c.ortho_inv(ST(stream), x_dev, out_dev, B) : op->c.ortho_fwd(ST(stream), x_dev, out_dev, B);
}

int kdip_gather(void* stream, const float* x, const long* idx, long nidx, long per, int B, float* out) { return gather_idx(ST(stream), x, idx, nidx, per, B, out); }
int kdip_scatter(void* stream, const float* y, const long* idx, long nidx, long per, int B, float* out) { return scatter_idx(ST(stream), y, idx, nidx, per, B, out); }
int kdip_mask_mul(void* stream, const float* x, const float* m, int B, long chw, float* out) { return mul_planes(ST(stream), x, m, (long)B * chw, chw, out); }
int kdip_resize_axis(void* stream, const float* x, const float* w, const int* fov, int taps, int n_in, int n_out, int other,
                     int axis, long planes, int adjoint, float* out) {
  return adjoint ? resize_axis_adj(ST(stream), x, w, fov, taps, n_in, n_out, other, axis, planes, out)
                 : resize_axis(ST(stream), x, w, fov, taps, n_in, n_out, other, axis, planes, out);
}
int kdip_blur_dense(void* stream, const float* x, const float* psf, int ks, int S, long planes, int adjoint, float* out) {
  return blur_dense_circ(ST(stream), x, psf, ks, S, planes, adjoint, out);
}
int kdip_fft2(void* stream, int S, const float* in, int real_in, float* out, int real_out, long planes, int inverse, float* tmp) {
  // twiddle table per device, created once under a lock (this entry point may be called from several host threads / devices)
  static std::mutex mu;
  static float2* tw_by_dev[64] = {nullptr};
  int dev = 0;
  KDIP_HIP_CHECK(hipGetDevice(&dev));
  KDIP_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
  float2* tw;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!tw_by_dev[dev]) {
      float2 host[128];
      make_twiddles256(host);
      KDIP_HIP_CHECK(hipMalloc((void**)&tw_by_dev[dev], sizeof(host)));
      KDIP_HIP_CHECK(hipMemcpy(tw_by_dev[dev], host, sizeof(host), hipMemcpyHostToDevice));
    }
    tw = tw_by_dev[dev];
  }
  return fft2(ST(stream), tw, S, in, real_in, (float2*)tmp, out, real_out, planes, inverse);
}

// --------------------------------------------------------------------------- guidance ----
int kdip_x0_epilogue_v1(void* stream, const float* uo, const float* x, int B, long HW, const float* t7, float* x0,
                        float* x0_raw, float* var) {
  KDIP_REQUIRE(uo && x && t7 && x0, "null argument");
  X0Params p;
  p.c_in = t7[0]; p.sqrt_recip = t7[1]; p.sqrt_recipm1 = t7[2]; p.log_beta = t7[3]; p.log_post_var = t7[4];
  p.post_var = t7[5]; p.coef1 = t7[6]; p.want_var = var ? 1 : 0;
  return x0_epilogue_v1(ST(stream), uo, x, B, HW, p, x0, x0_raw, var);
}
int kdip_x0_epilogue_v2(void* stream, const float* uo, const float* co, const float* x, int B, long HW, float sigma,
                        int want_var, float* x0, float* xv, float* tv) {
  KDIP_REQUIRE(uo && x && x0 && (!want_var || (co && xv && tv)), "null argument");
  return x0_epilogue_v2(ST(stream), uo, co, x, B, HW, sigma, want_var, x0, xv, tv);
}
int kdip_vjp_cotangent_v1(void* stream, const float* gh, const float* xraw, int B, long HW, float srm1, float* cot, float* graw) {
  return vjp_cotangent_v1(ST(stream), gh, xraw, B, HW, srm1, cot, graw);
}
int kdip_vjp_cotangent_v2(void* stream, const float* gh, int B, long HW, float* cot) { return vjp_cotangent_v2(ST(stream), gh, B, HW, cot); }
int kdip_guidance_combine(void* stream, const float* x0, const float* gd, float a, const float* uv, float b, float coef, long n, float* hat) {
  return guidance_combine(ST(stream), x0, gd, a, uv, b, coef, n, hat);
}
int kdip_axpby(void* stream, const float* x, float a, const float* y, float b, long n, float* out) { return axpby(ST(stream), x, a, y, b, n, out); }
int kdip_mul(void* stream, const float* x, const float* y, long n, float* out) { return mul_elem(ST(stream), x, y, n, out); }
int kdip_clamp(void* stream, const float* x, long n, float* out) { return clamp_pm1(ST(stream), x, n, out); }
int kdip_dps_normalize(void* stream, const float* x, long per_x, const float* r, long per_r, float zeta, int B, float* out,
                       float* nrm, double* tmp) {
  API_CK(norm_per_sample(ST(stream), r, B, per_r, nrm, tmp));
  return scale_per_sample_inv(ST(stream), x, nrm, zeta, B, per_x, out);
}

int kdip_sampler_add_noise(void* stream, const float* x, const float* eps, float s, long n, float* out) { return sampler_add_noise(ST(stream), x, eps, s, n, out); }
int kdip_sampler_euler(void* stream, const float* x, const float* den, float sh, float dt, long n, float* out) { return sampler_euler(ST(stream), x, den, sh, dt, n, out); }
int kdip_sampler_heun(void* stream, const float* x, const float* d1, const float* x2, const float* d2, float sh, float sn, float dt, long n, float* out) {
  return sampler_heun(ST(stream), x, d1, x2, d2, sh, sn, dt, n, out);
}

// ------------------------------------------------------------------------- test hooks ----
static inline int pad32i(int c) { return (c + 31) / 32 * 32; }

int kdip_debug_conv_timing(void* dev_buf, int H, int cin, int cout, int st_mode) { return conv_debug_timing(dev_buf, H, cin, cout, st_mode); }

int kdip_test_conv(void* stream, int dtype, int ntaps, const float* x_nchw, int B, int Cin, int H, int W, const float* w_host,
                   const float* bias_host, int Cout, int transpose_flip, float* y_nchw, int storage_out) {
  hipStream_t st = ST(stream);
  const DType cdt = dtype == KDIP_BF16 ? DT_BF16 : (dtype == KDIP_BF16X3 ? DT_F32X3 : (dtype == KDIP_F16X3 ? DT_F32H3 : DT_F32));      // conv arithmetic
  const DType dt = storage_dtype(cdt);
  size_t es = dt == DT_BF16 ? 2 : 4;
  // logical conv after optional transpose: Ci -> Co
  const int Ci = transpose_flip ? Cout : Cin, Co = transpose_flip ? Cin : Cout;
  const int cpad = pad32i(Ci);
  std::vector<char> buf(packed_weight_bytes(cdt, ntaps, cpad, Co));
  pack_conv_weight(cdt, w_host, Cout, Cin, ntaps, transpose_flip, cpad, buf.data());
  void *wp = nullptr, *xin = nullptr, *ys = nullptr; float *bias = nullptr, *y32 = nullptr, *skws = nullptr;
  KDIP_HIP_CHECK(hipMalloc(&wp, buf.size()));
  KDIP_HIP_CHECK(hipMemcpy(wp, buf.data(), buf.size(), hipMemcpyHostToDevice));
  if (bias_host) { KDIP_HIP_CHECK(hipMalloc((void**)&bias, sizeof(float) * Co)); KDIP_HIP_CHECK(hipMemcpy(bias, bias_host, sizeof(float) * Co, hipMemcpyHostToDevice)); }
  KDIP_HIP_CHECK(hipMalloc(&xin, es * (size_t)B * H * W * cpad));
  const int opad = pad32i(Co);
  int rc = nchw_to_nhwc(st, dt, x_nchw, B, Ci, H, W, 1.f, xin, cpad, cpad);
  // split-precision mode: a dgrad input is a gradient (no natural scale), so -- as the UNet's VJP does -- the fp16 window of the A
  // operand follows max |x|; forward inputs are taken at O(1) scale
  ConvStats stt;
  unsigned* amax = nullptr;
  if (is_x3(cdt) && transpose_flip) {
    KDIP_HIP_CHECK(hipMalloc((void**)&amax, sizeof(unsigned)));
    if (!rc) rc = amax_bits(st, x_nchw, (long)B * Ci * H * W, amax);
    stt.x3_amax = amax;
  }
  if (storage_out) {     // the UNet-internal epilogues: output in the storage dtype
    KDIP_HIP_CHECK(hipMalloc(&ys, es * (size_t)B * H * W * opad));
    const long wsf = (long)B * H * W * Co;             // zeroed split-K workspace: under-filled shapes take the split-K path
    KDIP_HIP_CHECK(hipMalloc((void**)&skws, sizeof(float) * wsf));
    KDIP_HIP_CHECK(hipMemsetAsync(skws, 0, sizeof(float) * wsf, st));
    if (!rc) rc = conv_forward(st, cdt, ntaps, xin, cpad, B, H, W, cpad, wp, bias, Co, ys, opad, nullptr, 0, 0, 1.f, 0, amax ? &stt : nullptr, skws, wsf);
    if (!rc) rc = nhwc_T_to_nchw_f32(st, dt, ys, opad, B, Co, H, W, y_nchw);
  } else {               // the fp32-output epilogue of the output heads / final input gradient
    KDIP_HIP_CHECK(hipMalloc((void**)&y32, sizeof(float) * (size_t)B * H * W * opad));
    if (!rc) rc = conv_forward(st, cdt, ntaps, xin, cpad, B, H, W, cpad, wp, bias, Co, y32, opad, nullptr, 0, 1, 1.f, 0, amax ? &stt : nullptr);
    if (!rc) rc = nhwc_to_nchw_f32(st, y32, opad, B, Co, H, W, y_nchw);
  }
  hipError_t e = hipStreamSynchronize(st);
  if (amax) (void)hipFree(amax);
  (void)hipFree(wp); (void)hipFree(xin); if (y32) (void)hipFree(y32); if (ys) (void)hipFree(ys); if (skws) (void)hipFree(skws); if (bias) (void)hipFree(bias);
  if (rc) return rc;
  KDIP_HIP_CHECK(e);
  return KDIP_OK;
}

// ------------------------------------------------------------------ stand-alone conv layers (LPIPS backbone) ----
struct kdip_conv { DType dt, cdt; int cin, cout, cin_pad, ntaps, device; void* w = nullptr; float* bias = nullptr; };
int kdip_conv_create(int device, int dtype, const float* w_host, const float* bias_host, int Cout, int Cin, int ntaps, kdip_conv** out) {
  KDIP_REQUIRE(out && w_host && (ntaps == 9 || ntaps == 1) && Cout > 0 && Cin > 0, "conv_create: bad arguments");
  KDIP_HIP_CHECK(hipSetDevice(device));
  kdip_conv* c = new kdip_conv;
  c->cdt = dtype == KDIP_BF16 ? DT_BF16 : (dtype == KDIP_BF16X3 ? DT_F32X3 : (dtype == KDIP_F16X3 ? DT_F32H3 : DT_F32)); c->dt = storage_dtype(c->cdt); c->cin = Cin; c->cout = Cout; c->cin_pad = pad32i(Cin); c->ntaps = ntaps; c->device = device;
  std::vector<char> buf(packed_weight_bytes(c->cdt, ntaps, c->cin_pad, Cout));
  pack_conv_weight(c->cdt, w_host, Cout, Cin, ntaps, 0, c->cin_pad, buf.data());
  bool ok = hipMalloc(&c->w, buf.size()) == hipSuccess && hipMemcpy(c->w, buf.data(), buf.size(), hipMemcpyHostToDevice) == hipSuccess;
  if (ok && bias_host)
    ok = hipMalloc((void**)&c->bias, sizeof(float) * Cout) == hipSuccess &&
         hipMemcpy(c->bias, bias_host, sizeof(float) * Cout, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {                               // nothing of a half-built layer survives an error
    kdip_conv_destroy(c);
    return set_error(KDIP_ERR_NOMEM, "conv_create: device allocation / upload failed");
  }
  *out = c;
  return KDIP_OK;
}
void kdip_conv_destroy(kdip_conv* c) {
  if (!c) return;
  if (c->w) (void)hipFree(c->w);
  if (c->bias) (void)hipFree(c->bias);
  delete c;
}
// y = conv(x) (+ bias); x [B,Cin,H,W] fp32 NCHW, y [B,Cout,H,W] fp32 NCHW; ws: caller-provided device scratch of
// kdip_conv_workspace_bytes(c, B, H, W) bytes (no allocation here: capture-safe)
long kdip_conv_workspace_bytes(kdip_conv* c, int B, int H, int W) {
  if (!c) return -1;
  const size_t es = c->dt == DT_BF16 ? 2 : 4;
  return (long)(es * (size_t)B * H * W * c->cin_pad + sizeof(float) * (size_t)B * H * W * pad32i(c->cout) + 512);
}
int kdip_conv_apply(kdip_conv* c, void* stream, const float* x_nchw, int B, int H, int W, float* y_nchw, void* ws) {
  KDIP_REQUIRE(c && x_nchw && y_nchw && ws, "conv_apply: null argument");
  hipStream_t st = ST(stream);
  const size_t es = c->dt == DT_BF16 ? 2 : 4;
  char* xin = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* y32 = (float*)(xin + ((es * (size_t)B * H * W * c->cin_pad + 255) & ~(size_t)255));
  const int opad = pad32i(c->cout);
  API_CK(nchw_to_nhwc(st, c->dt, x_nchw, B, c->cin, H, W, 1.f, xin, c->cin_pad, c->cin_pad));
  API_CK(conv_forward(st, c->cdt, c->ntaps, xin, c->cin_pad, B, H, W, c->cin_pad, c->w, c->bias, c->cout, y32, opad, nullptr, 0, 1, 1.f, c->cin));
  return nhwc_to_nchw_f32(st, y32, opad, B, c->cout, H, W, y_nchw);
}
int kdip_gauss_nll_mean(void* stream, const float* pred, const float* target, const float* logvar, int B, long per, int accumulate, float* out) {
  return gauss_nll_mean(ST(stream), pred, target, logvar, B, per, accumulate, out);
}
int kdip_relu_maxpool(void* stream, const float* x, long planes, int H, int W, int pool, float* y) { return relu_maxpool_planes(ST(stream), x, planes, H, W, pool, y); }
int kdip_lpips_layer(void* stream, const float* f0, const float* f1, const float* lin_w, int B, int C, long HW, float* out_accum) {
  return lpips_layer(ST(stream), f0, f1, lin_w, B, C, HW, out_accum);
}

// Test hook of the fused attention (csrc/attention.hip): device fp32 qkv [B][T][3C] (head h at channels 192 h + q|k|v) and output
// cotangent dO [B][T][C] are rounded to bf16, forward + VJP run, results come back as fp32 (o [B][T][C], dqkv [B][T][3C]).
int kdip_test_attention(void* stream, const float* qkv_dev, const float* dO_dev, int B, int T, int heads, float* o_dev, float* dqkv_dev) {
  hipStream_t st = ST(stream);
  const int C = heads * 64;
  const size_t nq = (size_t)B * T * 3 * C, no = (size_t)B * T * C;
  void *qkv = nullptr, *dO = nullptr, *o = nullptr, *dq = nullptr, *ws = nullptr; float *lse = nullptr, *D = nullptr;
  DevPool pool;
  API_CK(pool.get(&qkv, 2 * nq)); API_CK(pool.get(&dO, 2 * no)); API_CK(pool.get(&o, 2 * no)); API_CK(pool.get(&dq, 2 * nq));
  API_CK(pool.get(&ws, 2 * no * 3)); API_CK(pool.get((void**)&lse, sizeof(float) * B * heads * T)); API_CK(pool.get((void**)&D, sizeof(float) * B * heads * T));
  int rc = f32_to_bf16(st, qkv_dev, (long)nq, qkv);
  if (!rc) rc = f32_to_bf16(st, dO_dev, (long)no, dO);
  if (!rc) rc = attn_fused_forward(st, qkv, 3 * C, B, T, heads, ws, o, C, lse);
  if (!rc) rc = attn_fused_backward(st, qkv, 3 * C, dO, C, o, C, lse, B, T, heads, ws, D, dq, 3 * C);
  if (!rc) rc = T_to_f32(st, DT_BF16, o, (long)no, o_dev);
  if (!rc) rc = T_to_f32(st, DT_BF16, dq, (long)nq, dqkv_dev);
  hipError_t e = hipStreamSynchronize(st);         // (the pool frees after the stream has drained)
  if (rc) return rc;
  KDIP_HIP_CHECK(e);
  return KDIP_OK;
}

// conv3.hip through the C ABI: every tensor argument is a device fp32 NCHW tensor (converted to the bf16 NHWC storage
// layout here), coefficient tables are device fp32 arrays in the layouts of Conv3Fuse; reps > 1 re-runs the conv and
// reports the mean HIP-event time per launch in *avg_us_host.
int kdip_test_conv3(void* stream, const float* x_nchw, const float* x2_nchw, int B, int Cin, int H, int W, const float* w_host,
                    const float* bias_host, int Cout, int transpose_flip, int tf, const float* tf_coef_dev, const float* res_nchw,
                    int in_ups, int res_ups, int st_mode, const float* stx_nchw, const float* st_coef_dev, const float* st_mr_dev,
                    float* y_nchw, double* sums_dev, int reps, float* avg_us_host) {
  hipStream_t st = ST(stream);
  const DType dt = DT_BF16;
  const size_t es = 2;
  const int Ci = transpose_flip ? Cout : Cin, Co = transpose_flip ? Cin : Cout;
  const int cpad = pad32i(Ci);
  KDIP_REQUIRE(Ci == cpad && Co % 128 == 0, "test_conv3: Cin must be a multiple of 32 and Cout of 128");
  std::vector<char> buf(packed_weight_bytes(dt, 9, cpad, Co));
  pack_conv_weight(dt, w_host, Cout, Cin, 9, transpose_flip, cpad, buf.data());
  void *wp = nullptr, *xin = nullptr, *x2 = nullptr, *res = nullptr, *stx = nullptr, *ys = nullptr; float* bias = nullptr;
  const int Hi = in_ups ? H / 2 : H, Wi = in_ups ? W / 2 : W, Hr = res_ups ? H / 2 : H, Wr = res_ups ? W / 2 : W;
  DevPool pool;
  API_CK(pool.get(&wp, buf.size()));
  KDIP_HIP_CHECK(hipMemcpy(wp, buf.data(), buf.size(), hipMemcpyHostToDevice));
  if (bias_host) { API_CK(pool.get((void**)&bias, sizeof(float) * Co)); KDIP_HIP_CHECK(hipMemcpy(bias, bias_host, sizeof(float) * Co, hipMemcpyHostToDevice)); }
  API_CK(pool.get(&xin, es * (size_t)B * Hi * Wi * cpad));
  API_CK(pool.get(&ys, es * (size_t)B * H * W * Co));
  int rc = nchw_to_nhwc(st, dt, x_nchw, B, Ci, Hi, Wi, 1.f, xin, cpad, cpad);
  if (x2_nchw) { API_CK(pool.get(&x2, es * (size_t)B * H * W * cpad)); if (!rc) rc = nchw_to_nhwc(st, dt, x2_nchw, B, Ci, H, W, 1.f, x2, cpad, cpad); }
  if (res_nchw) { API_CK(pool.get(&res, es * (size_t)B * Hr * Wr * Co)); if (!rc) rc = nchw_to_nhwc(st, dt, res_nchw, B, Co, Hr, Wr, 1.f, res, Co, Co); }
  if (stx_nchw) { API_CK(pool.get(&stx, es * (size_t)B * H * W * Co)); if (!rc) rc = nchw_to_nhwc(st, dt, stx_nchw, B, Co, H, W, 1.f, stx, Co, Co); }
  Conv3Fuse fu;
  fu.in_ups = in_ups; fu.res_ups = res_ups; fu.tf = tf; fu.tf_silu = 1; fu.tf_coef = tf_coef_dev; fu.x2 = x2; fu.ldx2 = cpad;
  fu.st_mode = st_mode; fu.st_silu = 1; fu.st_sums = sums_dev; fu.st_x = stx; fu.st_ldx = Co; fu.st_coef = st_coef_dev; fu.st_mr = st_mr_dev;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (reps > 1) { KDIP_HIP_CHECK(hipEventCreate(&e0)); KDIP_HIP_CHECK(hipEventCreate(&e1)); }
  for (int r = 0; r < (reps > 1 ? reps + 1 : 1) && !rc; ++r) {
    if (r == 1) KDIP_HIP_CHECK(hipEventRecord(e0, st));          // launch 0 is the warm-up
    if (sums_dev) KDIP_HIP_CHECK(hipMemsetAsync(sums_dev, 0, sizeof(double) * B * 64, st));
    rc = conv3_forward(st, xin, cpad, B, H, W, cpad, wp, bias, Co, ys, Co, res, Co, &fu, Ci);
  }
  if (reps > 1 && !rc) {
    KDIP_HIP_CHECK(hipEventRecord(e1, st));
    KDIP_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0; KDIP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (avg_us_host) *avg_us_host = ms * 1e3f / reps;
  }
  if (!rc) rc = nhwc_T_to_nchw_f32(st, dt, ys, Co, B, Co, H, W, y_nchw);
  hipError_t e = hipStreamSynchronize(st);
  if (e0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
  if (rc) return rc;
  KDIP_HIP_CHECK(e);
  return KDIP_OK;
}

int kdip_debug_conv3_timing(void* dev_buf) { return conv3_debug_timing(dev_buf); }
int kdip_debug_gn_fold(int on) { unet_debug_gn_fold(on); return KDIP_OK; }
int kdip_debug_defer_finish(int on) { unet_debug_defer_finish(on); return KDIP_OK; }

int kdip_test_groupnorm(void* stream, int dtype, const float* x_nchw, int B, int C, int H, int W, const float* gamma_host,
                        const float* beta_host, const float* film_host, int silu, float* y_nchw, const float* dy_nchw,
                        float* dx_nchw) {
  hipStream_t st = ST(stream);
  DType dt = dtype == KDIP_BF16 ? DT_BF16 : DT_F32;
  size_t es = dt == DT_BF16 ? 2 : 4;
  const long HW = (long)H * W;
  void *x = nullptr, *y = nullptr, *dy = nullptr, *dx = nullptr;
  float *gamma, *beta, *film = nullptr, *coef, *mr; double *stats, *sums;
  KDIP_HIP_CHECK(hipMalloc(&x, es * B * HW * C)); KDIP_HIP_CHECK(hipMalloc(&y, es * B * HW * C));
  KDIP_HIP_CHECK(hipMalloc(&dy, es * B * HW * C)); KDIP_HIP_CHECK(hipMalloc(&dx, es * B * HW * C));
  KDIP_HIP_CHECK(hipMalloc((void**)&gamma, 4 * C)); KDIP_HIP_CHECK(hipMalloc((void**)&beta, 4 * C));
  KDIP_HIP_CHECK(hipMalloc((void**)&coef, 8 * B * C)); KDIP_HIP_CHECK(hipMalloc((void**)&mr, 4 * B * 64));
  KDIP_HIP_CHECK(hipMalloc((void**)&stats, 8 * B * 64)); KDIP_HIP_CHECK(hipMalloc((void**)&sums, 8 * B * 64));
  KDIP_HIP_CHECK(hipMemcpy(gamma, gamma_host, 4 * C, hipMemcpyHostToDevice));
  KDIP_HIP_CHECK(hipMemcpy(beta, beta_host, 4 * C, hipMemcpyHostToDevice));
  if (film_host) { KDIP_HIP_CHECK(hipMalloc((void**)&film, 8 * B * C)); KDIP_HIP_CHECK(hipMemcpy(film, film_host, 8 * B * C, hipMemcpyHostToDevice)); }
  int rc = nchw_to_nhwc(st, dt, x_nchw, B, C, H, W, 1.f, x, C, C);
  const bool small = gn_small_eligible(dt, HW, C);      // same choice as the UNet executor
  if (small) {
    if (!rc) rc = gn_fwd_small(st, dt, x, C, B, HW, C, gamma, beta, film, 0, 1e-5f, silu, y, C, coef, mr);
  } else {
    if (!rc) rc = gn_stats(st, dt, x, C, B, HW, C, stats);
    if (!rc) rc = gn_coef(st, stats, gamma, beta, film, B, HW, C, 1e-5f, coef, mr);
    if (!rc) rc = gn_apply(st, dt, x, C, coef, B, HW, C, silu, y, C);
  }
  if (!rc) rc = nhwc_T_to_nchw_f32(st, dt, y, C, B, C, H, W, y_nchw);
  if (!rc && dy_nchw && dx_nchw) {
    rc = nchw_to_nhwc(st, dt, dy_nchw, B, C, H, W, 1.f, dy, C, C);
    if (small) {
      if (!rc) rc = gn_bwd_small(st, dt, x, C, dy, C, coef, mr, B, HW, C, silu, nullptr, 0, dx, C);
    } else {
      if (!rc) rc = gn_bwd_stats(st, dt, x, C, dy, C, coef, mr, B, HW, C, silu, sums);
      if (!rc) rc = gn_bwd_apply(st, dt, x, C, dy, C, coef, mr, sums, B, HW, C, silu, nullptr, 0, dx, C);
    }
    if (!rc) rc = nhwc_T_to_nchw_f32(st, dt, dx, C, B, C, H, W, dx_nchw);
  }
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(x); (void)hipFree(y); (void)hipFree(dy); (void)hipFree(dx); (void)hipFree(gamma); (void)hipFree(beta);
  (void)hipFree(coef); (void)hipFree(mr); (void)hipFree(stats); (void)hipFree(sums); if (film) (void)hipFree(film);
  if (rc) return rc;
  KDIP_HIP_CHECK(e);
  return KDIP_OK;
}

}  // extern "C"
