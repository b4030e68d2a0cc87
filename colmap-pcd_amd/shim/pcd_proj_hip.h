// pcd_proj_hip.h -- reference-signature adapter of lidar::PcdProj (lidar/pcd_projection.h:48-185) over the C ABI.
//
//   PcdProjectionOptions        lidar/pcd_projection.h:31-47 (same member names; the save-image members are kept
//                               for source compatibility and ignored: SaveDepthImage needs OpenCV and is GUI/debug)
//   PcdProj::BuildSubMap        .cc:223-255  (here: from the device cloud PointCloudProcess already holds)
//   PcdProj::SetNewImage #1     .cc:13-89    std::map<point3D_t, 6-vector>
//   PcdProj::SetNewImage #2     .cc:102-220  pt_xys[i].second + pt_xyzs
//   PcdProj::SetNewImages       batched form of #1 for the loop of BundleAdjustmentConfig::Project2Image
//                               (optim/bundle_adjustment.cc:262-280): one GPU call for all images of a BA window
// Image / Camera are template parameters: any type with .qvec[4], .tvec[3], .points2D (each with .xy[2],
// HasPoint3D(), .point3D_id) and .params, .width, .height works (shim/ba_problem.h's mirrors do).
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pcdhip.h"

namespace colmap_hip {
namespace lidar {

struct PcdProjectionOptions {
  double depth_image_scale = 0.2;
  bool if_save_depth_image = false;
  std::string depth_image_folder;
  std::string original_image_folder;
  bool if_save_lidar_frame = false;
  std::string lidar_frame_folder;
  int max_proj_scale = 10;
  int min_proj_scale = 2;
  double min_proj_dist = 2;
  float submap_length = 1.0;
  float submap_width = 1.0;
  float submap_height = 1.0;
  float choose_meter = 40.0;
  double min_lidar_proj_dist = 0.0;
};

class PcdProj {
 public:
  using Vector6 = std::array<double, 6>;
  explicit PcdProj(PcdProjectionOptions options) : options_(std::move(options)) {}
  ~PcdProj() { pcd_proj_destroy(proj_); }
  PcdProj(const PcdProj&) = delete;
  PcdProj& operator=(const PcdProj&) = delete;

  // The reference takes the PCL cloud; the device cloud (already axis-swapped, ply.cc:38-54) is the same data.
  bool BuildSubMap(pcd_cloud* cloud) {
    pcd_proj_options o;
    o.depth_image_scale = options_.depth_image_scale;
    o.max_proj_scale = options_.max_proj_scale;
    o.min_proj_scale = options_.min_proj_scale;
    o.min_proj_dist = options_.min_proj_dist;
    o.submap_length = options_.submap_length;
    o.submap_width = options_.submap_width;
    o.submap_height = options_.submap_height;
    o.choose_meter = options_.choose_meter;
    o.min_lidar_proj_dist = options_.min_lidar_proj_dist;
    pcd_proj_destroy(proj_);
    proj_ = nullptr;
    return pcd_proj_create(cloud, &o, &proj_) == PCD_OK;
  }

  template <typename ImageT, typename CameraT>
  void SetNewImage(const ImageT& image, const CameraT& camera, std::map<uint64_t, Vector6>& map) {
    std::vector<const ImageT*> im{&image};
    std::vector<const CameraT*> cam{&camera};
    std::vector<std::map<uint64_t, Vector6>*> out{&map};
    SetNewImages(im, cam, out);
  }

  // all images of one Project2Image sweep in a single device call
  template <typename ImageT, typename CameraT>
  bool SetNewImages(const std::vector<const ImageT*>& images, const std::vector<const CameraT*>& cameras,
                    const std::vector<std::map<uint64_t, Vector6>*>& maps) {
    if (!proj_ || images.size() != cameras.size() || images.size() != maps.size()) return false;
    std::vector<pcd_proj_image> desc(images.size());
    std::vector<double> feat;
    std::vector<uint64_t> ids;
    for (size_t i = 0; i < images.size(); ++i) {
      Describe(*images[i], *cameras[i], &desc[i]);
      desc[i].feat_begin = ids.size();
      for (const auto& p2 : images[i]->points2D) {
        if (!p2.HasPoint3D()) continue;
        feat.push_back(p2.xy[0]);
        feat.push_back(p2.xy[1]);
        ids.push_back(p2.point3D_id);
      }
      desc[i].feat_end = ids.size();
    }
    std::vector<uint8_t> found(ids.size());
    std::vector<double> l6(6 * ids.size());
    if (pcd_proj_set_new_images(proj_, desc.size(), desc.data(), ids.size(), feat.data(), found.data(), nullptr,
                                nullptr, l6.data(), nullptr) != PCD_OK)
      return false;
    for (size_t i = 0; i < images.size(); ++i)
      for (uint64_t f = desc[i].feat_begin; f < desc[i].feat_end; ++f) {
        if (!found[f]) continue;
        Vector6 v;
        for (int k = 0; k < 6; ++k) v[k] = l6[6 * f + k];
        maps[i]->insert({ids[f], v});   // insert, not assign: the first Point2D of a 3D point wins (.cc:75)
      }
    return true;
  }

  template <typename ImageT, typename CameraT>
  void SetNewImage(const ImageT& image, const CameraT& camera,
                   std::vector<std::pair<std::array<double, 2>, bool>>& pt_xys,
                   std::vector<std::array<double, 3>>& pt_xyzs) {
    if (!proj_) return;
    pcd_proj_image d;
    Describe(image, camera, &d);
    d.feat_begin = 0;
    d.feat_end = pt_xys.size();
    std::vector<double> feat(2 * pt_xys.size()), cam(3 * pt_xys.size());
    std::vector<uint8_t> found(pt_xys.size());
    for (size_t i = 0; i < pt_xys.size(); ++i) {
      feat[2 * i] = pt_xys[i].first[0];
      feat[2 * i + 1] = pt_xys[i].first[1];
    }
    if (pcd_proj_set_new_images(proj_, 1, &d, pt_xys.size(), feat.data(), found.data(), nullptr, nullptr, nullptr,
                                cam.data()) != PCD_OK)
      return;
    for (size_t i = 0; i < pt_xys.size(); ++i) {
      pt_xys[i].second = found[i] != 0;
      pt_xyzs.push_back({cam[3 * i], cam[3 * i + 1], cam[3 * i + 2]});   // zeros when unmatched (.cc:176, 213)
    }
  }

  pcd_proj* handle() const { return proj_; }

 private:
  template <typename ImageT, typename CameraT>
  static void Describe(const ImageT& image, const CameraT& camera, pcd_proj_image* d) {
    for (int k = 0; k < 4; ++k) d->qvec[k] = image.qvec[k];
    for (int k = 0; k < 3; ++k) d->tvec[k] = image.tvec[k];
    for (int k = 0; k < 8; ++k) d->params[k] = k < (int)camera.params.size() ? camera.params[k] : 0.0;
    d->width = camera.width;
    d->height = camera.height;
  }
  PcdProjectionOptions options_;
  pcd_proj* proj_ = nullptr;
};

}  // namespace lidar
}  // namespace colmap_hip
