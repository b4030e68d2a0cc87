// exhaustive_matcher_hip.h -- the two host-side pieces of the reference's exhaustive feature matching that sit
// directly above the matcher kernel, with the reference's shapes:
//   ExhaustiveBlocks(num_images, block_size)   the image-pair lists of ExhaustiveFeatureMatcher::Run
//                                              (feature/matching.cc:902-960), one per block pair, in its order;
//   SiftBlockMatcherHIP::Match(image_pairs)    SiftFeatureMatcher::Match(image_pairs) (feature/matching.cc:798-880)
//                                              for one block: all pairs go to the device in ONE
//                                              pcd_sift_match_batch call instead of being queued to per-pair
//                                              workers (:358-380 CPU, :403-440 GPU).
// The matches of a pair are the exact brute-force result of feature/sift.cc:55-144 (MatchSiftFeaturesCPU), ascending
// in the first image's descriptor index.  No CPU fallback: Match returns false without a gfx950 device.
#pragma once
#include <cstdint>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/pcdhip.h"

namespace colmap_hip {

using image_t = uint32_t;
using ImagePairs = std::vector<std::pair<image_t, image_t>>;

// feature/matching.cc:921-953: blocks of `block_size` consecutive images; inside a block pair, (idx1, idx2) is taken
// when (idx1 > idx2 && idx1 % B <= idx2 % B) || (idx1 < idx2 && idx1 % B < idx2 % B) -- every unordered pair once.
inline std::vector<ImagePairs> ExhaustiveBlocks(const std::vector<image_t>& image_ids, size_t block_size) {
  std::vector<ImagePairs> lists;
  const size_t n = image_ids.size(), B = block_size;
  if (B == 0) return lists;
  const size_t nblocks = (n + B - 1) / B;
  lists.reserve(nblocks * nblocks);
  for (size_t bi = 0; bi < nblocks; ++bi)
    for (size_t bj = 0; bj < nblocks; ++bj) {
      const size_t i_end = std::min(n, (bi + 1) * B), j_end = std::min(n, (bj + 1) * B);
      ImagePairs list;
      for (size_t i = bi * B; i < i_end; ++i)
        for (size_t j = bj * B; j < j_end; ++j) {
          // position inside the block decides which of (i, j) / (j, i) is taken, so that each unordered pair of the
          // two blocks -- and each pair inside a diagonal block -- appears in exactly one list
          const size_t pi = i % B, pj = j % B;
          const bool take = i > j ? pi <= pj : (i < j && pi < pj);
          if (take) list.emplace_back(image_ids[i], image_ids[j]);
        }
      lists.push_back(std::move(list));
    }
  return lists;
}

struct SiftBlockOptions {   // SiftMatchingOptions as far as the brute-force matcher reads them (feature/sift.h:118-150)
  double max_ratio = 0.8;
  double max_distance = 0.7;
  bool cross_check = true;
  int gpu_index = 0;
};

class SiftBlockMatcherHIP {
 public:
  using Descriptors = std::pair<const uint8_t*, uint32_t>;   // rows x 128 uint8, row-major (FeatureDescriptors)
  using FeatureMatches = std::vector<std::pair<uint32_t, uint32_t>>;

  explicit SiftBlockMatcherHIP(const SiftBlockOptions& options = SiftBlockOptions()) : options_(options) {}

  // `descriptors(image_id)` plays FeatureMatcherCache::GetDescriptors.  results[p] = matches of image_pairs[p].
  template <typename GetDescriptors>
  bool Match(const ImagePairs& image_pairs, GetDescriptors&& descriptors, std::vector<FeatureMatches>* results) {
    results->assign(image_pairs.size(), FeatureMatches());
    if (image_pairs.empty()) return true;
    // the block's images, each once, in one arena
    std::unordered_map<image_t, uint32_t> slot;
    std::vector<uint8_t> arena;
    std::vector<uint64_t> first_row{0};
    std::vector<uint32_t> pairs;
    pairs.reserve(2 * image_pairs.size());
    uint64_t capacity = 0;
    auto slot_of = [&](image_t id) {
      auto it = slot.find(id);
      if (it != slot.end()) return it->second;
      const Descriptors d = descriptors(id);
      arena.insert(arena.end(), d.first, d.first + (size_t)d.second * 128);
      first_row.push_back(first_row.back() + d.second);
      const uint32_t s = (uint32_t)slot.size();
      slot.emplace(id, s);
      return s;
    };
    for (const auto& pr : image_pairs) {
      const uint32_t a = slot_of(pr.first), b = slot_of(pr.second);
      pairs.push_back(a); pairs.push_back(b);
      capacity += first_row[a + 1] - first_row[a];   // at most one match per descriptor of the first image
    }
    std::vector<uint32_t> matches(2 * (size_t)std::max<uint64_t>(capacity, 1));
    std::vector<uint64_t> list_offset(image_pairs.size() + 1, 0);
    if (pcd_sift_match_batch(options_.gpu_index, arena.empty() ? nullptr : arena.data(), first_row.data(), (int)slot.size(),
                             pairs.data(), (int)image_pairs.size(), (float)options_.max_ratio, (float)options_.max_distance,
                             options_.cross_check ? 1 : 0, matches.data(), capacity, list_offset.data()) != PCD_OK)
      return false;
    for (size_t p = 0; p < image_pairs.size(); ++p) {
      FeatureMatches& m = (*results)[p];
      m.reserve((size_t)(list_offset[p + 1] - list_offset[p]));
      for (uint64_t k = list_offset[p]; k < list_offset[p + 1]; ++k) m.emplace_back(matches[2 * k], matches[2 * k + 1]);
    }
    return true;
  }

 private:
  SiftBlockOptions options_;
};

}  // namespace colmap_hip
