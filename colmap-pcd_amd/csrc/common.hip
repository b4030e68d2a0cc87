// common.hip -- errors, device check, profiler, misc C-ABI entry points.
#include "common.h"

namespace pcd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

static bool is_gfx950(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
  return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

pcd_status require_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (%s); libpcdhip has no CPU fallback",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return PCD_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    set_error("device %d out of range (0..%d)", device, n - 1);
    return PCD_ERR_INVALID;
  }
  if (!is_gfx950(device)) {
    set_error("device %d is not gfx950 (MI355X); libpcdhip ships gfx950 code objects only", device);
    return PCD_ERR_NO_DEVICE;
  }
  PCD_HIP_TRY(hipSetDevice(device));
  return PCD_OK;
}

Profiler& Profiler::get() {
  static Profiler p;
  return p;
}

void Profiler::begin(const char* name, hipStream_t s) {
  Rec r;
  r.name = name;
  (void)hipEventCreate(&r.a);
  (void)hipEventCreate(&r.b);
  (void)hipEventRecord(r.a, s);
  std::lock_guard<std::mutex> g(mu);
  pending.push_back(r);
}

void Profiler::end(hipStream_t s) {
  std::lock_guard<std::mutex> g(mu);
  // the most recent record without an end event on this thread
  (void)hipEventRecord(pending.back().b, s);
}

void Profiler::collect() {
  std::lock_guard<std::mutex> g(mu);
  for (auto& r : pending) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
    bool hit = false;
    for (auto& t : totals)
      if (t.name == r.name) { t.launches++; t.ms += ms; hit = true; break; }
    if (!hit) totals.push_back({r.name, 1, (double)ms});
  }
  pending.clear();
}

void Profiler::reset() {
  collect();
  std::lock_guard<std::mutex> g(mu);
  totals.clear();
}

}  // namespace pcd

extern "C" {

const char* pcd_last_error(void) { return pcd::get_error(); }
int pcd_version(void) { return PCDHIP_VERSION_MAJOR * 100 + PCDHIP_VERSION_MINOR; }

int pcd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int d = 0; d < n; ++d) ok += pcd::is_gfx950(d) ? 1 : 0;
  return ok;
}

pcd_status pcd_profile_enable(int on) {
  pcd::Profiler::get().enabled = on != 0;
  return PCD_OK;
}
pcd_status pcd_profile_only(const char* scope) {
  pcd::Profiler::get().only = scope ? scope : "";
  return PCD_OK;
}
pcd_status pcd_profile_reset(void) {
  pcd::Profiler::get().reset();
  return PCD_OK;
}
pcd_status pcd_profile_get(pcd_kernel_time* entries, int cap, int* count) {
  auto& p = pcd::Profiler::get();
  p.collect();
  std::lock_guard<std::mutex> g(p.mu);
  int n = (int)p.totals.size();
  if (count) *count = n;
  for (int i = 0; i < n && i < cap && entries; ++i) {
    std::snprintf(entries[i].name, sizeof entries[i].name, "%s", p.totals[i].name.c_str());
    entries[i].launches = p.totals[i].launches;
    entries[i].total_ms = p.totals[i].ms;
  }
  return PCD_OK;
}

static const int kCamNumParams[11] = {3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12};
int pcd_camera_num_params(int model_id) { return (model_id >= 0 && model_id < 11) ? kCamNumParams[model_id] : -1; }

pcd_status pcd_camera_param_groups(int model_id, uint8_t* group) {
  PCD_REQUIRE(group && model_id >= 0 && model_id < 11, "camera model id / null pointer");
  // base/camera_models.h: models with one focal length (f cx cy ...) and with two (fx fy cx cy ...)
  const bool single_f = model_id == 0 || model_id == 2 || model_id == 3 || model_id == 8 || model_id == 9;
  const int nf = single_f ? 1 : 2;
  for (int k = 0; k < kCamNumParams[model_id]; ++k) group[k] = k < nf ? 0 : (k < nf + 2 ? 1 : 2);
  return PCD_OK;
}

pcd_status pcd_search_range_schedule(const int32_t* global_opt_num, uint64_t n, double kd_max, double kd_min,
                                     double drop_speed, double* out) {
  // sfm/incremental_mapper.cc:1159-1163, 1423-1427 (host-side scalar schedule;
  // the result is the per-query max_range array of pcd_associate)
  PCD_REQUIRE(out && (global_opt_num || n == 0), "null pointer");
  for (uint64_t i = 0; i < n; ++i) {
    double r = kd_max - global_opt_num[i] * drop_speed;
    if (r <= kd_min) r = kd_min;
    out[i] = r;
  }
  return PCD_OK;
}

}  // extern "C"
