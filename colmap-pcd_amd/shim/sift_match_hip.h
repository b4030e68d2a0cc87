// sift_match_hip.h -- adapter with the interface of lib/SiftGPU's SiftMatchGPU (lib/SiftGPU/SiftGPU.h:268-352)
// as far as the reference uses it (feature/sift.cc:1170-1266 CreateSiftGPUMatcher / MatchSiftFeaturesGPU):
//   SiftMatchHIP m(max_sift); m.gpu_index = i; m.VerifyContextGL(); m.SetMaxSift(n); m.GetMaxSift();
//   m.SetDescriptors(0|1, num, const unsigned char*); m.GetSiftMatch(max_match, buf, distmax, ratiomax, mutual)
// The result is the exact brute-force match set of feature/sift.cc:55-144 (what SiftGPU approximates), in
// ascending index of set 0.  Guided matching (GetGuidedSiftMatch) is not provided.  No CPU fallback: without a
// gfx950 device VerifyContextGL() returns 0 and GetSiftMatch() returns -1, the value the reference treats as
// "matching failed" (feature/sift.cc:1261-1266).
#pragma once
#include <cstdint>

#include "../../include/pcdhip.h"

namespace colmap_hip {

class SiftMatchHIP {
 public:
  int gpu_index = 0;
  explicit SiftMatchHIP(int max_sift = 4096) : max_sift_(max_sift) {}
  ~SiftMatchHIP() { pcd_sift_matcher_destroy(m_); }
  SiftMatchHIP(const SiftMatchHIP&) = delete;
  SiftMatchHIP& operator=(const SiftMatchHIP&) = delete;

  int VerifyContextGL() { return Ensure() ? 1 : 0; }
  int CreateContextGL() { return VerifyContextGL(); }
  bool Allocate(int max_sift, int /*mbm*/) { SetMaxSift(max_sift); return Ensure(); }
  void SetMaxSift(int max_sift) {
    max_sift_ = max_sift;
    if (m_) pcd_sift_matcher_set_max_sift(m_, max_sift);
  }
  int GetMaxSift() const { return max_sift_; }
  void SetDescriptors(int index, int num, const unsigned char* descriptors, int /*id*/ = -1) {
    if (Ensure()) ok_ = pcd_sift_matcher_set_descriptors(m_, index, num, descriptors) == PCD_OK;
  }
  int GetSiftMatch(int max_match, uint32_t match_buffer[][2], float distmax = 0.7f, float ratiomax = 0.8f,
                   int mutual_best_match = 1) {
    if (!Ensure() || !ok_) return -1;
    int32_t n = 0;
    if (pcd_sift_matcher_match(m_, max_match, &match_buffer[0][0], distmax, ratiomax, mutual_best_match, &n) != PCD_OK)
      return -1;
    return n;
  }

 private:
  bool Ensure() {
    if (!m_ && pcd_sift_matcher_create(gpu_index, max_sift_, &m_) != PCD_OK) m_ = nullptr;
    return m_ != nullptr;
  }
  int max_sift_;
  bool ok_ = true;
  pcd_sift_matcher* m_ = nullptr;
};

}  // namespace colmap_hip
