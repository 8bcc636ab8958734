#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>
namespace kdip {
thread_local std::string g_last_error;
int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

static std::mutex g_cus_mu;
static std::vector<std::pair<hipStream_t, int>> g_stream_cus;
void stream_register_cus(hipStream_t st, int ncus) {
  std::lock_guard<std::mutex> lk(g_cus_mu);
  for (size_t i = 0; i < g_stream_cus.size(); ++i)
    if (g_stream_cus[i].first == st) { g_stream_cus.erase(g_stream_cus.begin() + i); break; }
  if (ncus > 0) g_stream_cus.push_back({st, ncus});
}
int stream_cus(hipStream_t st, int device_cus) {
  std::lock_guard<std::mutex> lk(g_cus_mu);
  for (auto& e : g_stream_cus) if (e.first == st) return e.second < device_cus ? e.second : device_cus;
  return device_cus;
}

bool g_prof_on = false;
struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; const char* tag; long d[4]; };
static std::vector<ProfRec> g_recs;            // guarded by g_prof_mu (launches may come from several host threads)
static std::mutex g_prof_mu;
static thread_local ProfRec g_cur;             // the launch being bracketed on this host thread
void prof_begin(hipStream_t st, int cls, double flops, double bytes, const char* tag, long d0, long d1, long d2, long d3) {
  if (!g_prof_on) return;
  g_cur.cls = cls; g_cur.flops = flops; g_cur.bytes = bytes; g_cur.tag = tag; g_cur.d[0] = d0; g_cur.d[1] = d1; g_cur.d[2] = d2; g_cur.d[3] = d3;
  (void)hipEventCreate(&g_cur.a); (void)hipEventCreate(&g_cur.b);
  (void)hipEventRecord(g_cur.a, st);
}
void prof_end(hipStream_t st) {
  if (!g_prof_on) return;
  (void)hipEventRecord(g_cur.b, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_recs.push_back(g_cur);
}
static const char* kClassNames[PC_COUNT] = {"conv3x3_igemm_128x128", "conv3x3_igemm_128x64", "conv3x3_igemm_128x32",
                                            "conv1x1_igemm_128x128", "conv1x1_igemm_128x64", "conv1x1_igemm_128x32",
                                            "gn_stats", "gn_apply", "gn_bwd_stats", "gn_bwd_apply",
                                            "op_blur", "op_fft2", "op_otf", "op_dwt", "op_gather_scatter_mask", "op_resize", "pointwise"};
}  // namespace kdip

extern "C" {
int kdip_profile_enable(int on) {
  using namespace kdip;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_recs.clear();
  g_prof_on = on != 0;
  return 0;
}
int kdip_profile_dump(const char* path) {
  using namespace kdip;
  FILE* f = fopen(path, "w");
  if (!f) return set_error(KDIP_ERR_ARG, "cannot open %s", path);
  fprintf(f, "class,tag,d0,d1,d2,d3,gflop,mbytes,us\n");
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.b);
    float t = 0; (void)hipEventElapsedTime(&t, r.a, r.b);
    fprintf(f, "%s,%s,%ld,%ld,%ld,%ld,%.3f,%.3f,%.2f\n", kClassNames[r.cls], r.tag ? r.tag : "", r.d[0], r.d[1], r.d[2], r.d[3],
            r.flops / 1e9, r.bytes / 1e6, t * 1e3);
  }
  fclose(f);
  return 0;
}
int kdip_profile_num_classes(void) { return kdip::PC_COUNT; }
const char* kdip_profile_class_name(int cls) { return (cls >= 0 && cls < kdip::PC_COUNT) ? kdip::kClassNames[cls] : "?"; }
int kdip_profile_report(double* ms, double* flops, double* bytes, long* launches) {
  using namespace kdip;
  for (int i = 0; i < PC_COUNT; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; launches[i] = 0; }
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return set_error(KDIP_ERR_HIP, "profile: event sync failed");
    float t = 0;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return set_error(KDIP_ERR_HIP, "profile: elapsed failed");
    ms[r.cls] += t; flops[r.cls] += r.flops; bytes[r.cls] += r.bytes; launches[r.cls] += 1;
  }
  return 0;
}
}
