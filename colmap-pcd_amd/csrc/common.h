// common.h -- shared host-side plumbing of libpcdhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <exception>
#include <new>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pcdhip.h"

namespace pcd {

// ----------------------------------------------------------------- errors --
void set_error(const char* fmt, ...);

// pcdhip.h promises that no entry point throws: the ones that allocate host memory (std::vector, new) run their body
// inside guard(), which turns std::bad_alloc into PCD_ERR_OOM and anything else into PCD_ERR_INVALID.
template <typename F>
inline pcd_status guard(F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    set_error("host memory allocation failed");
    return PCD_ERR_OOM;
  } catch (const std::exception& e) {
    set_error("unexpected exception: %s", e.what());
    return PCD_ERR_INVALID;
  } catch (...) {
    set_error("unexpected exception");
    return PCD_ERR_INVALID;
  }
}
const char* get_error();

#define PCD_HIP_TRY(expr)                                                              \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      ::pcd::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return (_e == hipErrorOutOfMemory) ? PCD_ERR_OOM : PCD_ERR_HIP;                  \
    }                                                                                  \
  } while (0)

#define PCD_TRY(expr)                    \
  do {                                   \
    pcd_status _s = (expr);              \
    if (_s != PCD_OK) return _s;         \
  } while (0)

#define PCD_REQUIRE(cond, msg)                                        \
  do {                                                                \
    if (!(cond)) {                                                    \
      ::pcd::set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, msg); \
      return PCD_ERR_INVALID;                                         \
    }                                                                 \
  } while (0)

// Checks that `device` exists and is a gfx950 part.  No CPU fallback exists.
pcd_status require_device(int device);

// The *_device entry points are written for EAGER launches on the caller's stream: they grow per-handle scratch
// (hipFree + hipMalloc), build per-cloud tables behind a hipStreamSynchronize on first use and time scopes with
// events.  None of that may be recorded into a graph -- a replay would write through scratch addresses a later eager
// call has freed (the memory fault seen when round 3's step was captured with torch.cuda.graph and replayed) -- so a
// capturing stream is refused before anything is touched.  (Asking about the legacy stream while another stream of
// the device captures in global mode is an error of the caller's; it is reported the same way.)
inline pcd_status refuse_capture(hipStream_t s, const char* fn) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  const hipError_t e = hipStreamIsCapturing(s, &st);
  if (e != hipSuccess || st != hipStreamCaptureStatusNone) {
    if (e != hipSuccess) (void)hipGetLastError();
    set_error("%s: the stream is capturing a graph; this entry point allocates scratch and may synchronise, "
              "it can only be launched eagerly", fn);
    return PCD_ERR_UNSUPPORTED;
  }
  return PCD_OK;
}
#define PCD_REFUSE_CAPTURE(stream) PCD_TRY(::pcd::refuse_capture((hipStream_t)(stream), __func__))

// ------------------------------------------------------- device buffers ----
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;  // capacity in elements
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  // grow-only (keeps contents undefined)
  pcd_status reserve(size_t count) {
    if (count <= n) return PCD_OK;
    release();
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), bytes);
    if (e != hipSuccess) {
      p = nullptr;
      set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
      return PCD_ERR_OOM;
    }
    n = count;
    return PCD_OK;
  }
};

// pinned host buffer (hipHostMalloc), grow-only
template <typename T>
struct PinnedBuf {
  T* p = nullptr;
  size_t n = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { release(); }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = 0;
  }
  pcd_status reserve(size_t count) {
    if (count <= n) return PCD_OK;
    release();
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
      p = nullptr;
      set_error("hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
      return PCD_ERR_OOM;
    }
    n = count;
    return PCD_OK;
  }
};

// --------------------------------------------------------------- profiler --
// When enabled, every kernel launch through PCD_LAUNCH is bracketed by HIP
// events on its own stream; bench.py reads the per-kernel totals.
struct Profiler {
  static Profiler& get();
  bool enabled = false;
  std::string only;   // non-empty: time just the scope of this name (pcd_profile_only)
  struct Rec { std::string name; hipEvent_t a, b; };
  std::vector<Rec> pending;
  struct Tot { std::string name; uint64_t launches; double ms; };
  std::vector<Tot> totals;
  std::mutex mu;
  void begin(const char* name, hipStream_t s);
  void end(hipStream_t s);
  void collect();  // syncs events, folds pending into totals
  void reset();
};

struct ScopedKernelTimer {
  hipStream_t s;
  bool on;
  ScopedKernelTimer(const char* name, hipStream_t stream) : s(stream), on(Profiler::get().enabled) {
    if (on && !Profiler::get().only.empty() && Profiler::get().only != name) on = false;
    if (on) Profiler::get().begin(name, s);
  }
  ~ScopedKernelTimer() {
    if (on) Profiler::get().end(s);
  }
};

inline unsigned div_up(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace pcd
