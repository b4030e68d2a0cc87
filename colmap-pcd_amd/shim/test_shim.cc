// test_shim.cc -- exercises the reference-signature adapters (lidar_hip.h, ba_problem.h).
//   ./test_shim            structure tests only (no GPU needed)
//   ./test_shim --gpu      also the GPU paths (fails if no gfx950 device)
// Expected values marked [ref] are the reference's own (src/optim/bundle_adjustment_test.cc); the
// others are derived for this fork's defaults (intrinsics constant, optim/bundle_adjustment.h:79-81).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include <hip/hip_runtime_api.h>   // the sharded-cloud test's caller-side reduction moves device buffers

#include "ba_problem.h"
#include "ceres_adapter.h"
#include "pose_reader.h"
#include "pose_writer.h"
#include "sift_match_hip.h"
#include "exhaustive_matcher_hip.h"
#include "lidar_hip.h"

using namespace colmap_hip;

static int g_fail = 0;
#define CHECK_EQ(a, b)                                                                         \
  do {                                                                                         \
    auto _a = (a); auto _b = (b);                                                              \
    if (!(_a == _b)) { std::printf("FAIL %s:%d: %s == %s (%g vs %g)\n", __FILE__, __LINE__, #a, #b, (double)_a, (double)_b); ++g_fail; } \
  } while (0)
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

// shape of bundle_adjustment_test.cc:123-184 GenerateReconstruction: every image observes every point,
// SIMPLE_RADIAL f = 1200, 1000 x 1000, identity rotation, t = (U(-1,1), U(-1,1), 10), +-2 px noise
static void GenerateReconstruction(size_t num_images, size_t num_points, Reconstruction* rec) {
  std::mt19937 rng(0);
  std::uniform_real_distribution<double> u(-1.0, 1.0), px(-2.0, 2.0);
  for (point3D_t p = 1; p <= num_points; ++p) {
    Point3D pt;
    for (double& c : pt.xyz) c = u(rng);
    rec->points3D[p] = pt;
  }
  for (image_t i = 0; i < num_images; ++i) {
    Camera cam;
    cam.model_id = PCD_CAM_SIMPLE_RADIAL;
    cam.params = {1200.0, 500.0, 500.0, 0.0};
    rec->cameras[i] = cam;
    Image im;
    im.camera_id = i;
    im.tvec[0] = u(rng); im.tvec[1] = u(rng); im.tvec[2] = 10;
    for (point3D_t p = 1; p <= num_points; ++p) {
      const Point3D& pt = rec->points3D[p];
      const double X = pt.xyz[0] + im.tvec[0], Y = pt.xyz[1] + im.tvec[1], Z = pt.xyz[2] + im.tvec[2];
      Point2D p2;
      p2.xy[0] = 1200.0 * X / Z + 500.0 + px(rng);
      p2.xy[1] = 1200.0 * Y / Z + 500.0 + px(rng);
      p2.point3D_id = p;
      im.points2D.push_back(p2);
      rec->points3D[p].track.push_back({i, (point2D_t)(p - 1)});
    }
    rec->images[i] = im;
  }
}

static void DeleteObservation(Reconstruction* rec, image_t image_id, point2D_t idx) {
  Point2D& p2 = rec->images[image_id].points2D[idx];
  auto& tr = rec->points3D[p2.point3D_id].track;
  tr.erase(std::remove_if(tr.begin(), tr.end(), [&](const TrackElement& t) { return t.image_id == image_id && t.point2D_idx == idx; }), tr.end());
  p2.point3D_id = kInvalidPoint3DId;
}

static void TestConfigNumObservations() {   // [ref] bundle_adjustment_test.cc:186-210
  Reconstruction rec;
  GenerateReconstruction(4, 100, &rec);
  BundleAdjustmentConfig config;
  config.AddImage(0);
  config.AddImage(1);
  CHECK_EQ(config.NumResiduals(rec), 400u);
  config.AddVariablePoint(1);
  CHECK_EQ(config.NumResiduals(rec), 404u);
  config.AddConstantPoint(2);
  CHECK_EQ(config.NumResiduals(rec), 408u);
  config.AddImage(2);
  CHECK_EQ(config.NumResiduals(rec), 604u);
  config.AddImage(3);
  CHECK_EQ(config.NumResiduals(rec), 800u);
}

// The reference's BA tests were written for upstream COLMAP's defaults refine_focal_length = true,
// refine_principal_point = false, refine_extra_params = true; this fork defaults all three to false
// (optim/bundle_adjustment.h:73-81), so the [ref] counts are reproduced with the upstream values set explicitly
// and the fork's own defaults are checked beside them.
static BundleAdjustmentOptions UpstreamOptions() {
  BundleAdjustmentOptions o;
  o.refine_focal_length = true; o.refine_principal_point = false; o.refine_extra_params = true;
  return o;
}

static void TestTwoView() {   // [ref] bundle_adjustment_test.cc:212-245: 400 residuals, 309 parameters
  Reconstruction rec;
  GenerateReconstruction(2, 100, &rec);
  BundleAdjustmentConfig config;
  config.AddImage(0); config.AddImage(1);
  config.SetConstantPose(0);
  config.SetConstantTvec(1, {0});
  BundleAdjusterHip ba(UpstreamOptions(), config);
  ba.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::NoLidar);
  CHECK_EQ(ba.NumResiduals(), 400u);
  CHECK_EQ(ba.NumResidualsReduced(), 400u);      // [ref]
  CHECK_EQ(ba.NumEffectiveParameters(), 309u);   // [ref] 100 x 3 + 5 (pose of image 1) + 2 x 2 (f, k of each camera)
  CHECK_EQ(ba.NumConstantPoints(), 0u);
  BundleAdjusterHip fork(BundleAdjustmentOptions(), config);   // the fork's defaults: intrinsics constant
  fork.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::NoLidar);
  CHECK_EQ(fork.NumEffectiveParameters(), 305u);
}

static void TestTwoViewConstantCamera() {   // [ref] bundle_adjustment_test.cc:247-285: 400 / 302
  Reconstruction rec;
  GenerateReconstruction(2, 100, &rec);
  BundleAdjustmentConfig config;
  config.AddImage(0); config.AddImage(1);
  config.SetConstantPose(0); config.SetConstantPose(1);
  config.SetConstantCamera(0);
  BundleAdjusterHip ba(UpstreamOptions(), config);
  ba.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::NoLidar);
  CHECK_EQ(ba.NumResidualsReduced(), 400u);      // [ref]
  CHECK_EQ(ba.NumEffectiveParameters(), 302u);   // [ref] 100 x 3 + 2 parameters of camera 1
  CHECK(!ba.CameraVariable(0) && ba.CameraVariable(1));
}

static void TestPartiallyContainedTracks() {   // [ref] bundle_adjustment_test.cc:287-328: 400 / 7
  Reconstruction rec;
  GenerateReconstruction(3, 100, &rec);
  const point3D_t variable_point = rec.images[2].points2D[0].point3D_id;
  DeleteObservation(&rec, 2, 0);
  BundleAdjustmentConfig config;
  config.AddImage(0); config.AddImage(1);
  config.SetConstantPose(0); config.SetConstantPose(1);
  BundleAdjusterHip ba(UpstreamOptions(), config);
  ba.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::NoLidar);
  CHECK_EQ(ba.NumResiduals(), 400u);
  CHECK_EQ(ba.NumResidualsReduced(), 400u);     // [ref] (the camera blocks keep every residual in)
  CHECK_EQ(ba.NumConstantPoints(), 99u);        // every track that also lives in image 2 is held constant (:1107-1131)
  CHECK_EQ(ba.NumEffectiveParameters(), 7u);    // [ref] 1 x 3 point + 2 x 2 camera parameters
  for (size_t i = 0; i < ba.point_ids_.size(); ++i) CHECK((ba.point_ids_[i] == variable_point) == (ba.point_const_[i] == 0));
  BundleAdjusterHip fork(BundleAdjustmentOptions(), config);
  fork.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::NoLidar);
  CHECK_EQ(fork.NumEffectiveParameters(), 3u);
  CHECK_EQ(fork.NumResidualsReduced(), 4u);     // only the two observations of the one variable point remain
}

static void TestForceToOptimizePoint() {   // [ref] bundle_adjustment_test.cc:330-392: 402 / 10
  Reconstruction rec;
  GenerateReconstruction(3, 100, &rec);
  const point3D_t add_variable = rec.images[2].points2D[1].point3D_id;
  const point3D_t add_constant = rec.images[2].points2D[2].point3D_id;
  DeleteObservation(&rec, 2, 0);
  BundleAdjustmentConfig config;
  config.AddImage(0); config.AddImage(1);
  config.SetConstantPose(0); config.SetConstantPose(1);
  config.AddVariablePoint(add_variable);
  config.AddConstantPoint(add_constant);
  BundleAdjusterHip ba(UpstreamOptions(), config);
  ba.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::NoLidar);
  // un-reduced: 400 + 2 (variable point seen from image 2) + 2 (constant point seen from image 2)
  CHECK_EQ(ba.NumResiduals(), 404u);
  // [ref] 402: the block of the constant point in image 2 has only constant parameters (camera 2 is made constant
  // when AddPointToProblem meets it, :951-954) and is dropped
  CHECK_EQ(ba.NumResidualsReduced(), 402u);
  CHECK_EQ(ba.NumEffectiveParameters(), 10u);   // [ref] 2 x 3 points + 2 x 2 camera parameters
  CHECK_EQ(ba.NumConstantPoints(), 98u);
  CHECK(ba.cam_model_.size() == 3 && !ba.CameraVariable(2));
}

static void TestConstantPoints() {   // [ref] bundle_adjustment_test.cc:394-440: 400 / 298
  Reconstruction rec;
  GenerateReconstruction(2, 100, &rec);
  BundleAdjustmentConfig config;
  config.AddImage(0); config.AddImage(1);
  config.SetConstantPose(0); config.SetConstantPose(1);
  config.AddConstantPoint(1); config.AddConstantPoint(2);
  BundleAdjusterHip ba(UpstreamOptions(), config);
  ba.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::NoLidar);
  CHECK_EQ(ba.NumResidualsReduced(), 400u);      // [ref]
  CHECK_EQ(ba.NumEffectiveParameters(), 298u);   // [ref] 98 x 3 + 2 x 2
  CHECK_EQ(ba.NumConstantPoints(), 2u);
  for (size_t i = 0; i < ba.point_ids_.size(); ++i)
    CHECK((ba.point_ids_[i] == 1 || ba.point_ids_[i] == 2) == (ba.point_const_[i] != 0));
}

static void TestVariableImage() {   // [ref] bundle_adjustment_test.cc:442-484: 600 / 317
  Reconstruction rec;
  GenerateReconstruction(3, 100, &rec);
  BundleAdjustmentConfig config;
  config.AddImage(0); config.AddImage(1); config.AddImage(2);
  config.SetConstantPose(0);
  config.SetConstantTvec(1, {0});
  BundleAdjusterHip ba(UpstreamOptions(), config);
  ba.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::NoLidar);
  CHECK_EQ(ba.NumResidualsReduced(), 600u);      // [ref]
  CHECK_EQ(ba.NumEffectiveParameters(), 317u);   // [ref] 100 x 3 + 5 + 6 + 3 x 2
  CHECK(ba.image_const_pose_[0] == 1 && ba.image_const_tvec_[1] == 1 && ba.image_const_pose_[2] == 0 && ba.image_const_tvec_[2] == 0);
}

static void TestCameraParameterGroups() {   // [ref] bundle_adjustment_test.cc:486-657: 307 / 313 / 307
  for (int variant = 0; variant < 3; ++variant) {
    Reconstruction rec;
    GenerateReconstruction(2, 100, &rec);
    BundleAdjustmentConfig config;
    config.AddImage(0); config.AddImage(1);
    config.SetConstantPose(0);
    config.SetConstantTvec(1, {0});
    BundleAdjustmentOptions o = UpstreamOptions();
    if (variant == 0) o.refine_focal_length = false;      // TestConstantFocalLength
    if (variant == 1) o.refine_principal_point = true;    // TestVariablePrincipalPoint
    if (variant == 2) o.refine_extra_params = false;      // TestConstantExtraParam
    BundleAdjusterHip ba(o, config);
    ba.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::NoLidar);
    CHECK_EQ(ba.NumResidualsReduced(), 400u);   // [ref]
    CHECK_EQ(ba.NumEffectiveParameters(), variant == 1 ? 313u : 307u);   // [ref] 305 + 2 / + 8 / + 2
    // SIMPLE_RADIAL f cx cy k: which entries of each camera are optimised
    for (int c = 0; c < 2; ++c) {
      const uint8_t* m = ba.cam_refine_.data() + ba.cam_off_[c];
      CHECK_EQ((int)m[0], variant == 0 ? 0 : 1);
      CHECK_EQ((int)m[1], variant == 1 ? 1 : 0);
      CHECK_EQ((int)m[2], variant == 1 ? 1 : 0);
      CHECK_EQ((int)m[3], variant == 2 ? 0 : 1);
    }
  }
}

static void TestLidarBlocks() {   // optim/bundle_adjustment.cc:993-1040 weights / NaN guard, :601-682 phrases
  Reconstruction rec;
  GenerateReconstruction(2, 10, &rec);
  BundleAdjustmentConfig config;
  config.AddImage(0); config.AddImage(1);
  for (point3D_t p = 1; p <= 10; ++p) config.AddVariablePoint(p);
  LidarPoint a; a.type = LidarPointType::Icp; a.abcd = {0, 1, 0, -0.5};
  LidarPoint b; b.type = LidarPointType::IcpGround; b.abcd = {0, 1, 0, -0.5};
  LidarPoint c; c.type = LidarPointType::Proj; c.abcd = {0, 1, 0, -0.5};
  LidarPoint d; d.type = LidarPointType::Icp; d.abcd = {0, NAN, 0, -0.5};
  config.AddLidarPoint(1, a); config.AddLidarPoint(2, b); config.AddLidarPoint(3, c); config.AddLidarPoint(4, d);
  BundleAdjusterHip ba(BundleAdjustmentOptions(), config);
  ba.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::WholeMap);
  CHECK_EQ(ba.lidar_point_.size(), 3u);         // NaN plane dropped
  CHECK_EQ(ba.NumResiduals(), 40u + 3u);
  CHECK_EQ(ba.lidar_w_[0], 100.0); CHECK_EQ(ba.lidar_w_[1], 1000.0); CHECK_EQ(ba.lidar_w_[2], 1.0);
  BundleAdjustmentOptions off; off.if_add_lidar_constraint = false;
  BundleAdjusterHip ba2(off, config);
  ba2.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::Local);
  CHECK_EQ(ba2.lidar_point_.size(), 0u);
  // Global phrase: only points flagged IfInSphere enter (AddImageInSphereToProblem :734)
  for (point3D_t p = 1; p <= 5; ++p) rec.points3D[p].if_in_sphere = true;
  BundleAdjusterHip ba3(BundleAdjustmentOptions(), config);
  ba3.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::Global);
  CHECK_EQ(ba3.obs_image_.size(), 10u);
}

// PLY ingest (lidar/ply.cc:14 pcl::io::loadPLYFile contract): ascii and binary, extra properties, nx/ny/nz
static std::string WritePly(bool binary, bool short_names, const std::vector<float>& xyz, const std::vector<float>& nrm) {
  const std::string path = std::string("/tmp/pcdhip_test_") + (binary ? "bin" : "ascii") + (short_names ? "_nx" : "") + ".ply";
  FILE* f = std::fopen(path.c_str(), "wb");
  const size_t n = xyz.size() / 3;
  std::fprintf(f, "ply\nformat %s 1.0\ncomment test\nelement vertex %zu\n", binary ? "binary_little_endian" : "ascii", n);
  std::fprintf(f, "property float x\nproperty float y\nproperty float z\nproperty uchar intensity\n");
  if (short_names) std::fprintf(f, "property double nx\nproperty double ny\nproperty double nz\n");
  else std::fprintf(f, "property float normal_x\nproperty float normal_y\nproperty float normal_z\n");
  std::fprintf(f, "element face 0\nproperty list uchar int vertex_indices\nend_header\n");
  for (size_t i = 0; i < n; ++i) {
    const unsigned char inten = (unsigned char)(i & 255);
    if (binary) {
      std::fwrite(&xyz[3 * i], 4, 3, f);
      std::fwrite(&inten, 1, 1, f);
      if (short_names) { double d[3] = {nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]}; std::fwrite(d, 8, 3, f); }
      else std::fwrite(&nrm[3 * i], 4, 3, f);
    } else {
      std::fprintf(f, "%.9g %.9g %.9g %d %.9g %.9g %.9g\n", xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], (int)inten,
                   nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    }
  }
  std::fclose(f);
  return path;
}

// MatchVariablePoint2LidarPoint (optim/bundle_adjustment.cc:288-350) on hand-computed candidates
static void TestMatchVariablePoint() {
  std::map<uint32_t, std::map<uint64_t, std::array<double, 6>>> searched;
  const double X[3] = {1.0, 2.0, 3.0};
  // image 5: lidar point straight below X along its normal -> |cos| = 1;  image 6: offset sideways -> smaller |cos|
  searched[5][42] = {1.0, 2.0, 2.5, 0, 0, 2.0};
  searched[6][42] = {0.0, 2.0, 2.5, 0, 0, -4.0};
  searched[7][99] = {9, 9, 9, 1, 0, 0};                 // other point only
  std::vector<uint32_t> track = {5, 7, 6, 8};
  LidarPoint lp;
  CHECK(MatchVariablePoint2LidarPoint(searched, 42, X, track, &lp));
  CHECK(lp.type == LidarPointType::Proj && lp.color[0] == 255 && lp.color[1] == 0);
  CHECK(lp.xyz[0] == 0.0 && lp.xyz[2] == 2.5);                       // image 6 wins: cos = 0.5/sqrt(1.25) < 1
  CHECK(lp.abcd[2] == -1.0 && lp.abcd[3] == 2.5);                    // normalised plane z = 2.5
  CHECK(std::fabs(lp.dist - 0.5) < 1e-15);                           // point-to-plane
  CHECK(std::fabs(lp.angle - 0.5 / std::sqrt(1.25)) < 1e-15);
  CHECK(!MatchVariablePoint2LidarPoint(searched, 43, X, track, &lp));
  std::vector<uint32_t> only5 = {5};
  CHECK(MatchVariablePoint2LidarPoint(searched, 42, X, only5, &lp) && lp.xyz[0] == 1.0 && lp.angle == 1.0);
}

// LoadPose (controllers/incremental_mapper.cc:920-996) on hand-computed poses
static void TestPoseReader() {
  const std::string path = "/tmp/pcdhip_pose_test.ply";
  {
    std::ofstream f(path);
    f << "ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nend_header\n";
    f << "1 2 3 0 0 0\n";                       // image 1: pure translation
    f << "0 0 0 nan 0 0\n";                     // image 2: skipped, but the id advances
    f << "0 0 0 0 0 1.5707963267948966\n";      // image 3: yaw +90 deg (turn left)
    f << "5 0 0 0 0 3.141592653589793\n";       // image 4: trace <= 0 branch of the quaternion extraction
  }
  std::map<uint32_t, std::array<double, 7>> poses;
  CHECK(LoadPosePly(path, &poses));
  CHECK_EQ(poses.size(), 3u);
  CHECK(poses.count(1) && !poses.count(2) && poses.count(3) && poses.count(4));
  // image 1: R = I, t_wc = (-2, -3, 1) -> t_cw = (2, 3, -1), q = (1, 0, 0, 0)
  CHECK(poses[1][0] == 2 && poses[1][1] == 3 && poses[1][2] == -1 && poses[1][3] == 1 && poses[1][4] == 0);
  // image 3: R_wc = Ry(-90 deg): the camera's optical axis (0,0,1) looks along world (-1,0,0) = LiDAR +y (left)
  const double h = std::sqrt(0.5);
  CHECK(std::fabs(poses[3][3] - h) < 1e-15 && std::fabs(poses[3][5] - h) < 1e-15);      // q_cw = (c, 0, +s, 0)
  CHECK(std::fabs(poses[3][4]) < 1e-15 && std::fabs(poses[3][6]) < 1e-15);
  // image 4: half turn about the vertical axis; t_wc = (0, 0, 5), R_cw t = (0,0,-5) -> t_cw = (0, 0, 5)
  CHECK(std::fabs(std::fabs(poses[4][5]) - 1.0) < 1e-15 && std::fabs(poses[4][3]) < 1e-15);
  CHECK(std::fabs(poses[4][2] - 5.0) < 1e-12 && std::fabs(poses[4][0]) < 1e-12);
  std::remove(path.c_str());
  CHECK(!LoadPosePly("/tmp/pcdhip_no_such_pose.ply", &poses));
}

// SaveImagePoses (ui/main_window.cc:1078-1182): hand-computed poses, the "nan" rows, and the round trip through LoadPose
static void TestPoseWriter() {
  const std::string path = "/tmp/pcdhip_pose_writer_test.ply";
  const double h = std::sqrt(0.5), kPi = 3.14159265358979323846;
  std::map<uint32_t, std::array<double, 7>> poses;
  poses[1] = {2, 3, -1, 1, 0, 0, 0};        // R = I, t_cw = (2, 3, -1): camera at t_wc = (-2, -3, 1) = LiDAR (1, 2, 3)
  poses[3] = {0, 0, 0, h, 0, h, 0};         // q_cw = rotation by +90 deg about y: the camera looks along LiDAR +y, yaw = +pi/2
  poses[4] = {0, 0, 5, 0, 0, 1, 0};         // half turn about the vertical axis, camera at LiDAR x = 5 (pose_reader test, image 4)
  poses[6] = {0, 0, 0, std::cos(0.25), 0, 0, std::sin(0.25)};   // R_cw = Rz(0.5): R_wc = Rz(-0.5), roll = -0.5
  poses[7] = {0, 0, 0, std::cos(0.2), std::sin(0.2), 0, 0};     // R_cw = Rx(0.4): R_wc = Rx(-0.4), Euler x angle -0.4, pitch = +0.4
  CHECK(SaveImagePosesPly(path, 8, poses));
  std::vector<std::string> lines;
  {
    std::ifstream f(path);
    std::string l;
    while (std::getline(f, l)) lines.push_back(l);
  }
  CHECK_EQ(lines.size(), 10u + 8u);
  CHECK(lines[0] == "ply" && lines[1] == "format ascii 1.0" && lines[2] == "element vertex 8");
  CHECK(lines[3] == "property float x" && lines[6] == "property float roll" && lines[7] == "property float pitch" &&
        lines[8] == "property float yaw" && lines[9] == "end_header");
  CHECK(lines[10 + 1] == "nan nan nan nan nan nan" && lines[10 + 4] == "nan nan nan nan nan nan" &&
        lines[10 + 7] == "nan nan nan nan nan nan");
  auto parse = [&](int id) {
    std::array<double, 6> v{};
    std::stringstream ss(lines[10 + id - 1]);
    for (double& x : v) ss >> x;
    return v;
  };
  auto near6 = [&](int id, std::array<double, 6> e) {
    const auto v = parse(id);
    bool ok = true;
    for (int k = 0; k < 6; ++k) ok = ok && std::fabs(v[k] - e[k]) < 1e-5 * std::max(1.0, std::fabs(e[k]));   // 6 significant digits in the file
    return ok;
  };
  CHECK(near6(1, {1, 2, 3, 0, 0, 0}));
  CHECK(near6(3, {0, 0, 0, 0, 0, kPi / 2}));          // through the fold: eulerAngles gives (pi/2, -pi, +-pi)
  {
    const auto v = parse(4);                          // a half turn: yaw = +-pi, the camera at x = 5
    CHECK(std::fabs(v[0] - 5) < 1e-5 && std::fabs(v[1]) < 1e-5 && std::fabs(v[2]) < 1e-5);
    CHECK(std::fabs(std::fabs(v[5]) - kPi) < 1e-5 && std::fabs(v[3]) < 1e-5 && std::fabs(v[4]) < 1e-5);
  }
  CHECK(near6(6, {0, 0, 0, -0.5, 0, 0}));
  CHECK(near6(7, {0, 0, 0, 0, 0.4, 0}));
  // the reader gives the poses back (float text: 6 significant digits)
  std::map<uint32_t, std::array<double, 7>> back;
  CHECK(LoadPosePly(path, &back));
  CHECK_EQ(back.size(), poses.size());
  for (const auto& kv : poses) {
    CHECK(back.count(kv.first));
    const auto& a = kv.second;
    const auto& b = back[kv.first];
    const double sgn = (a[3] * b[3] + a[4] * b[4] + a[5] * b[5] + a[6] * b[6]) < 0 ? -1.0 : 1.0;   // q and -q are one rotation
    for (int k = 0; k < 3; ++k) CHECK(std::fabs(a[k] - b[k]) < 2e-5);
    for (int k = 3; k < 7; ++k) CHECK(std::fabs(a[k] - sgn * b[k]) < 2e-5);
  }
  // round trip from the LiDAR side over a grid of angles (|pitch| < pi/2, angles away from +-pi where the sign is free)
  std::map<uint32_t, std::array<double, 7>> grid;
  std::vector<std::array<double, 6>> src;
  uint32_t id = 0;
  for (double roll : {-2.9, -1.0, 0.0, 0.7, 3.0})
    for (double pitch : {-1.5, -0.6, 0.0, 0.3, 1.4})
      for (double yaw : {-3.0, -1.8, -0.2, 0.0, 0.9, 2.6}) {
        const std::array<double, 6> p = {1.5 + id, -2.0 + 0.25 * id, 0.125 * id, roll, pitch, yaw};
        src.push_back(p);
        grid[++id] = LidarPoseToColmap(p.data());
      }
  CHECK(SaveImagePosesPly(path, (int)id, grid));
  lines.clear();
  {
    std::ifstream f(path);
    std::string l;
    while (std::getline(f, l)) lines.push_back(l);
  }
  CHECK_EQ(lines.size(), 10u + id);
  for (uint32_t i = 1; i <= id; ++i) CHECK(near6((int)i, src[i - 1]));
  std::remove(path.c_str());
  CHECK(!SaveImagePosesPly("/tmp/pcdhip_no_such_dir/pose.ply", 1, poses));
}

static void TestPlyReader() {
  std::vector<float> xyz, nrm;
  std::mt19937 rng(3);
  std::uniform_real_distribution<float> u(-50.f, 50.f);
  for (int i = 0; i < 1000; ++i) for (int k = 0; k < 3; ++k) { xyz.push_back(u(rng)); nrm.push_back(u(rng) / 50.f); }
  xyz[3 * 17 + 1] = NAN;   // NaN rows survive the reader; the axis swap / filter drops them later
  for (int variant = 0; variant < 4; ++variant) {
    const std::string p = WritePly(variant & 1, variant & 2, xyz, nrm);
    std::vector<float> x2, n2;
    CHECK(ReadPlyXYZNormal(p, &x2, &n2));
    CHECK_EQ(x2.size(), xyz.size());
    bool same = x2.size() == xyz.size();
    for (size_t i = 0; same && i < xyz.size(); ++i) {
      same = (std::isnan(xyz[i]) ? std::isnan(x2[i]) : x2[i] == xyz[i]) && n2[i] == nrm[i];
    }
    CHECK(same);
    std::remove(p.c_str());
  }
  std::vector<float> a, b;
  CHECK(!ReadPlyXYZNormal("/tmp/pcdhip_does_not_exist.ply", &a, &b));   // load failure -> false (ply.cc:14-17)
  {  // ascii: signed non-finite tokens keep their sign; a header that promises more vertices than the file
     // can hold is a failed load, not an allocation of that size
    const std::string p = "/tmp/pcdhip_test_tokens.ply";
    const char* hdr = "ply\nformat ascii 1.0\nelement vertex %s\nproperty float x\nproperty float y\nproperty float z\n"
                      "property float normal_x\nproperty float normal_y\nproperty float normal_z\nend_header\n";
    FILE* f = std::fopen(p.c_str(), "wb");
    std::fprintf(f, hdr, "2");
    std::fprintf(f, "-inf 1 2 -nan 0 1\n3 inf -0.5 0 nan 1e-3\n");
    std::fclose(f);
    CHECK(ReadPlyXYZNormal(p, &a, &b));
    CHECK_EQ(a.size(), 6u);
    CHECK(a.size() == 6 && std::isinf(a[0]) && a[0] < 0 && a[1] == 1.f && std::isnan(b[0]) && std::isinf(a[4]) && a[4] > 0 &&
          a[5] == -0.5f && std::isnan(b[4]) && b[5] == 1e-3f);
    f = std::fopen(p.c_str(), "wb");
    std::fprintf(f, hdr, "4000000000000");
    std::fprintf(f, "0 0 0 0 0 1\n");
    std::fclose(f);
    CHECK(!ReadPlyXYZNormal(p, &a, &b));
    f = std::fopen(p.c_str(), "wb");
    std::fprintf(f, hdr, "2");
    std::fprintf(f, "0 0 0 0 0 1\n1 2 3\n");    // short row
    std::fclose(f);
    CHECK(!ReadPlyXYZNormal(p, &a, &b));
    std::remove(p.c_str());
  }
}

// SiftMatchGPU-shaped adapter on the hand-made descriptors of feature/sift_test.cc (two unit descriptors each:
// expected 2 matches (0,0),(1,1) [ref]; with the second set's rows swapped the matches follow)
static int TestGpuSiftMatcher() {
  std::vector<unsigned char> d1(2 * 128, 0), d2(2 * 128, 0);
  for (int k = 0; k < 4; ++k) {            // two orthogonal descriptors of norm 510 (~ the 512 of real SIFT rows):
    d1[k] = d2[k] = 255;                   // dot 260100 -> acos(0.992) = 0.13 < 0.7, second best 0 -> ratio passes
    d1[128 + 4 + k] = d2[128 + 4 + k] = 255;
  }
  SiftMatchHIP m(4096);
  CHECK(m.VerifyContextGL());
  m.SetDescriptors(0, 2, d1.data());
  m.SetDescriptors(1, 2, d2.data());
  uint32_t buf[8][2];
  CHECK_EQ(m.GetSiftMatch(8, buf), 2);
  CHECK(buf[0][0] == 0 && buf[0][1] == 0 && buf[1][0] == 1 && buf[1][1] == 1);
  std::swap_ranges(d2.begin(), d2.begin() + 128, d2.begin() + 128);
  m.SetDescriptors(1, 2, d2.data());       // slot 0 stays resident
  CHECK_EQ(m.GetSiftMatch(8, buf), 2);
  CHECK(buf[0][0] == 0 && buf[0][1] == 1 && buf[1][0] == 1 && buf[1][1] == 0);
  CHECK_EQ(m.GetSiftMatch(1, buf), 1);     // max_match clips
  m.SetDescriptors(1, 0, nullptr);
  CHECK_EQ(m.GetSiftMatch(8, buf), 0);     // empty set: no matches
  m.SetMaxSift(1);
  m.SetDescriptors(0, 2, d1.data());       // clipped to one descriptor each
  m.SetDescriptors(1, 2, d2.data());       // (rows swapped above: first rows are orthogonal)
  CHECK_EQ(m.GetSiftMatch(8, buf), 0);
  m.SetDescriptors(1, 2, d1.data());
  CHECK_EQ(m.GetSiftMatch(8, buf), 1);
  return 0;
}

// ExhaustiveFeatureMatcher::Run's block enumeration (feature/matching.cc:921-953): every unordered pair exactly once,
// num_blocks^2 lists, at most block_size^2 pairs per list
static void TestExhaustiveBlocks() {
  for (const auto& cfg : std::vector<std::pair<size_t, size_t>>{{1, 50}, {2, 50}, {23, 5}, {50, 50}, {51, 50}, {120, 50}, {450, 50}}) {
    const size_t n = cfg.first, B = cfg.second;
    std::vector<image_t> ids(n);
    for (size_t i = 0; i < n; ++i) ids[i] = (image_t)(100 + 3 * i);   // image ids are not positions
    const auto blocks = ExhaustiveBlocks(ids, B);
    const size_t nb = (n + B - 1) / B;
    CHECK_EQ(blocks.size(), nb * nb);
    std::vector<uint8_t> seen(n * n, 0);
    size_t total = 0;
    for (const auto& pairs : blocks) {
      CHECK(pairs.size() <= B * B);
      for (const auto& pr : pairs) {
        const size_t a = (pr.first - 100) / 3, b = (pr.second - 100) / 3;
        CHECK(a != b && a < n && b < n);
        CHECK(!seen[a * n + b] && !seen[b * n + a]);
        seen[a * n + b] = 1;
        ++total;
      }
    }
    CHECK_EQ(total, n * (n - 1) / 2);
  }
}

// SiftFeatureMatcher::Match(image_pairs) for a block in one batched call == the SiftMatchGPU-shaped matcher pair by pair
static int TestGpuSiftBlockMatcher() {
  std::mt19937 rng(17);
  const int n_img = 7;
  const int sizes[n_img] = {300, 0, 129, 1, 511, 256, 77};
  std::vector<std::vector<uint8_t>> desc(n_img);
  std::vector<uint8_t> pool(600 * 128);
  for (auto& v : pool) v = (uint8_t)(rng() % 64);
  for (int i = 0; i < n_img; ++i) {
    desc[i].resize((size_t)sizes[i] * 128);
    for (int r = 0; r < sizes[i]; ++r) {
      const int src = (int)(rng() % 600);
      for (int k = 0; k < 128; ++k) desc[i][(size_t)r * 128 + k] = (uint8_t)std::min(255, pool[(size_t)src * 128 + k] + (int)(rng() % 3));
    }
  }
  std::vector<image_t> ids;
  for (int i = 0; i < n_img; ++i) ids.push_back((image_t)(10 + i));
  SiftBlockOptions opt;
  opt.max_ratio = 0.95; opt.max_distance = 1.2;
  SiftBlockMatcherHIP block(opt);
  auto get = [&](image_t id) { return SiftBlockMatcherHIP::Descriptors(desc[id - 10].data(), (uint32_t)sizes[id - 10]); };
  size_t total = 0;
  for (const auto& pairs : ExhaustiveBlocks(ids, 4)) {
    std::vector<SiftBlockMatcherHIP::FeatureMatches> res;
    CHECK(block.Match(pairs, get, &res));
    CHECK_EQ(res.size(), pairs.size());
    for (size_t p = 0; p < pairs.size(); ++p) {
      const int a = (int)pairs[p].first - 10, b = (int)pairs[p].second - 10;
      SiftMatchHIP m(4096);
      CHECK(m.VerifyContextGL());
      m.SetDescriptors(0, sizes[a], desc[a].data());
      m.SetDescriptors(1, sizes[b], desc[b].data());
      std::vector<uint32_t> buf(2 * (size_t)std::max(sizes[a], 1));
      const int k = m.GetSiftMatch(std::max(sizes[a], 1), reinterpret_cast<uint32_t(*)[2]>(buf.data()), (float)opt.max_distance,
                                   (float)opt.max_ratio, 1);
      CHECK_EQ((size_t)std::max(k, 0), res[p].size());
      for (int j = 0; j < k; ++j) CHECK(res[p][j].first == buf[2 * j] && res[p][j].second == buf[2 * j + 1]);
      total += res[p].size();
    }
  }
  CHECK(total > 100);
  return 0;
}

// PcdProj mirror: wall z = 10 m in front of a camera at the origin (pinhole 3039 px, 4032 x 3024)
static int TestGpuProjection() {
  std::vector<float> xyz, nrm;
  for (int i = -100; i <= 100; ++i)
    for (int j = -60; j <= 60; ++j) {
      const float vx = 0.05f * i, vy = 0.05f * j, vz = 10.0f;         // visual frame
      xyz.insert(xyz.end(), {vz, -vx, -vy});
      nrm.insert(nrm.end(), {-1.f, 0.f, 0.f});                         // visual normal (0,0,-1)
    }
  lidar::PointCloudProcess pcp;
  CHECK(pcp.InitializeFromRawCloud(xyz.data(), nrm.data(), xyz.size() / 3));
  lidar::PcdProjectionOptions pp;
  pp.min_lidar_proj_dist = 0.5;
  CHECK(pcp.BuildProjector(pp));
  CHECK(pcd_proj_num_submaps(pcp.pcd_proj_->handle()) > 0);
  Camera cam;
  cam.model_id = 4;
  cam.params = {3039, 3039, 2016, 1512, 0, 0, 0, 0};
  cam.width = 4032; cam.height = 3024;
  Image img;
  auto add = [&](double u, double v, point3D_t id) { Point2D p; p.xy[0] = u; p.xy[1] = v; p.point3D_id = id; img.points2D.push_back(p); };
  add(2016, 1512, 11);            // centre: hits the wall
  add(2016 + 1000, 1512, 12);     // 3.29 m to the right at 10 m: still on the wall (|x| <= 5)
  add(2016 + 1900, 1512, 13);     // 6.25 m: beyond the wall's edge (and beyond any splat)
  add(100, 100, kInvalidPoint3DId);   // no 3D point: not a feature of overload #1
  add(2016, 1512, 14);
  std::map<uint64_t, lidar::PcdProj::Vector6> map;
  pcp.pcd_proj_->SetNewImage(img, cam, map);
  CHECK_EQ(map.size(), 3u);
  CHECK(map.count(11) && map.count(12) && map.count(14) && !map.count(13));
  CHECK(map[11][2] == 10.0 && map[11][5] == -1.0 && std::fabs(map[11][0]) < 0.5);
  CHECK(map[11] == map[14]);
  CHECK(std::fabs(map[12][0] - 3.29) < 0.5);
  // overload #2: every pt_xy, plane/ray intersection in the camera frame
  std::vector<std::pair<std::array<double, 2>, bool>> pt_xys = {{{2016, 1512}, false}, {{3016, 1512}, false},
                                                                {{3916, 1512}, true}, {{-50, 10}, true}};
  std::vector<std::array<double, 3>> pt_xyzs;
  pcp.pcd_proj_->SetNewImage(img, cam, pt_xys, pt_xyzs);
  CHECK_EQ(pt_xyzs.size(), 4u);
  CHECK(pt_xys[0].second && pt_xys[1].second && !pt_xys[2].second && !pt_xys[3].second);
  CHECK(pt_xyzs[0][2] == 10.0 && pt_xyzs[0][0] == 0.0);
  CHECK(std::fabs(pt_xyzs[1][0] - 10.0 * 1000 / 3039) < 1e-12 && pt_xyzs[1][2] == 10.0);
  CHECK(pt_xyzs[2][0] == 0.0 && pt_xyzs[2][2] == 0.0 && pt_xyzs[3][2] == 0.0);
  return 0;
}


// ---- Ceres adapter (shim/ceres_adapter.h) -----------------------------------------------------------------------
// block shapes = the template arguments of the reference's AutoDiffCostFunction instantiations
// (optim/bundle_adjustment.cc:858-893, :967-983, :1031-1037); no GPU needed
static void TestCeresBlockShapes() {
  HipBlockBuffers buf;
  static const double res5[5] = {1, 2, 3, 4, 5};
  buf.b.residuals = res5;
  HipReprojectionBlock var(&buf, 0, false, 4), cst(&buf, 1, true, 8);
  buf.num_obs = 2;
  HipLidarBlock lid(&buf, 0);
  CHECK_EQ(var.num_residuals(), 2); CHECK_EQ(cst.num_residuals(), 2); CHECK_EQ(lid.num_residuals(), 1);
  CHECK((var.parameter_block_sizes() == std::vector<int32_t>{4, 3, 3, 4}));
  CHECK((cst.parameter_block_sizes() == std::vector<int32_t>{3, 8}));
  CHECK((lid.parameter_block_sizes() == std::vector<int32_t>{3}));
  double r[2] = {0, 0};
  CHECK(var.Evaluate(nullptr, r, nullptr) && r[0] == 1 && r[1] == 2);   // residual-only evaluation
  CHECK(cst.Evaluate(nullptr, r, nullptr) && r[0] == 3 && r[1] == 4);
  CHECK(lid.Evaluate(nullptr, r, nullptr) && r[0] == 5);
  double j[8]; double* jac[4] = {j, nullptr, nullptr, nullptr};
  CHECK(!var.Evaluate(nullptr, r, jac));   // Jacobians requested but the callback prepared none
}

// End-to-end replay of the reference's batch call order (controllers/bundle_adjustment.cc:76-204):
// PLY -> PointCloudProcess::Initialize -> association of EVERY 3D point (batched, controller gate) ->
// AddVariablePoint / AddLidarPoint -> BundleAdjuster SetUp (WholeMap) -> what ceres::Solve does per iteration:
// EvaluationCallback::PrepareForEvaluation, then CostFunction::Evaluate of every residual block.
static int TestGpuCeresAdapterEndToEnd() {
  // LiDAR map: floor y = 0.5 (visual frame) on a 5 cm lattice, normals (0,1,0), written as a binary PLY
  std::vector<float> xyz, nrm;
  for (int i = -60; i <= 60; ++i)
    for (int j = -60; j <= 60; ++j) {
      const float vx = 0.05f * i, vy = 0.5f, vz = 0.05f * j;
      xyz.insert(xyz.end(), {vz, -vx, -vy});                           // raw LiDAR frame = (z', -x', -y')
      nrm.insert(nrm.end(), {0.f, 0.f, -1.f});
    }
  const std::string ply = WritePly(true, false, xyz, nrm);
  lidar::PointCloudProcess pcp(ply);
  CHECK(pcp.Initialize());                                             // :124 LoadPointcloud
  std::remove(ply.c_str());
  Reconstruction rec;
  GenerateReconstruction(3, 80, &rec);
  rec.cameras[2].params[3] = 0.01;                                     // some distortion on one camera
  BundleAdjustmentConfig config;
  for (image_t i = 0; i < 3; ++i) config.AddImage(i);                  // :109
  config.SetConstantPose(0);
  config.SetConstantTvec(1, {0});
  // :130-185, batched: every point3D is a variable point; associated ones get a LidarPoint
  std::vector<uint64_t> ids; std::vector<double> pts;
  for (point3D_t p = 1; p <= 80; ++p) {
    config.AddVariablePoint(p);
    ids.push_back(p);
    pts.insert(pts.end(), rec.points3D[p].xyz, rec.points3D[p].xyz + 3);
  }
  std::unordered_map<uint64_t, LidarPoint> maps;
  CHECK(MatchClosestLidarPoints(pcp, ids, pts, {0.0}, PCD_GATE_CONTROLLER, &maps));
  size_t expect = 0;
  for (point3D_t p = 1; p <= 80; ++p) expect += std::fabs(rec.points3D[p].xyz[1] - 0.5) <= 1.0;   // dist2plane gate
  CHECK_EQ(maps.size(), expect);
  CHECK(expect > 40 && expect < 80);
  for (const auto& kv : maps) {
    CHECK(kv.second.type == LidarPointType::IcpGround);                // |ny/nx| = inf > 10
    config.AddLidarPoint(kv.first, kv.second);                         // :179
  }
  BundleAdjustmentOptions options = UpstreamOptions();                 // refined f + k: camera blocks are live
  options.loss_function_type = PCD_LOSS_SOFT_L1;                       // Ceres applies the loss itself: raw blocks
  BundleAdjusterHip ba(options, config);
  ba.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::WholeMap);         // :198 SetOptimazePhrase(WholeMap), :479
  CHECK_EQ(ba.NumResiduals(), 2u * 240u + expect);
  CHECK(ba.Create(0));
  ShimParameterSource src(&rec);
  HipEvaluation<> cb(&ba, src);
  // the residual blocks exactly as AddResidualBlock receives them
  struct Block { std::unique_ptr<ceres::CostFunction> f; std::vector<double*> params; std::vector<bool> constant; };
  std::vector<Block> blocks;
  for (size_t o = 0; o < ba.obs_image_.size(); ++o) {
    const int im = ba.obs_image_[o], pt = ba.obs_point_[o], cm = ba.image_cam_[im];
    Block b;
    b.f.reset(cb.ReprojectionBlock(o));
    if (!ba.image_const_pose_[im]) {
      b.params = {src.Qvec(ba.image_ids_[im]), src.Tvec(ba.image_ids_[im])};
      b.constant = {false, false};
    }
    b.params.push_back(src.XYZ(ba.point_ids_[pt])); b.constant.push_back(ba.point_const_[pt] != 0);
    b.params.push_back(src.Params(ba.camera_ids_[cm])); b.constant.push_back(!ba.CameraVariable(cm));
    CHECK_EQ(b.f->parameter_block_sizes().size(), b.params.size());
    blocks.push_back(std::move(b));
  }
  for (size_t l = 0; l < ba.lidar_point_.size(); ++l) {
    Block b;
    b.f.reset(cb.LidarBlock(l));
    b.params = {src.XYZ(ba.point_ids_[ba.lidar_point_[l]])};
    b.constant = {false};
    blocks.push_back(std::move(b));
  }
  // reference values: one direct evaluation through the C ABI
  const size_t O = ba.obs_image_.size(), L = ba.lidar_point_.size();
  std::vector<double> res(2 * O + L), jq(8 * O), jt(6 * O), jx(6 * O), jl(3 * L), jc(2 * PCD_CAM_JAC_STRIDE * O);
  pcd_ba_out direct{};
  direct.residuals = res.data(); direct.jac_q = jq.data(); direct.jac_t = jt.data(); direct.jac_X = jx.data();
  direct.jac_lidar = jl.data(); direct.jac_cam = jc.data();
  CHECK_EQ((int)pcd_ba_evaluate(ba.handle(), &direct), (int)PCD_OK);
  auto evaluate_all = [&](bool with_jac, std::vector<double>* all_r, std::vector<std::vector<double>>* all_j) {
    all_r->clear(); all_j->clear();
    bool ok = true;
    for (Block& b : blocks) {
      double r[2] = {0, 0};
      std::vector<std::vector<double>> j(b.params.size());
      std::vector<double*> jp(b.params.size(), nullptr);
      for (size_t k = 0; k < b.params.size(); ++k) {
        j[k].assign((size_t)b.f->num_residuals() * b.f->parameter_block_sizes()[k], -777.0);
        if (!b.constant[k]) jp[k] = j[k].data();                       // Ceres passes NULL for constant blocks
      }
      ok &= b.f->Evaluate(b.params.data(), r, with_jac ? jp.data() : nullptr);
      for (int k = 0; k < b.f->num_residuals(); ++k) all_r->push_back(r[k]);
      for (auto& v : j) all_j->push_back(v);
    }
    return ok;
  };
  std::vector<double> r1; std::vector<std::vector<double>> j1;
  cb.PrepareForEvaluation(/*evaluate_jacobians=*/true, /*new_evaluation_point=*/true);
  CHECK(cb.ok());
  CHECK(evaluate_all(true, &r1, &j1));
  CHECK(r1 == res);                                                    // every residual, bit for bit
  size_t jb = 0, checked = 0, untouched = 0;
  for (size_t o = 0; o < O; ++o) {
    const int im = ba.obs_image_[o], cm = ba.image_cam_[im], K = pcd_camera_num_params(ba.cam_model_[cm]);
    if (!ba.image_const_pose_[im]) {
      CHECK(std::equal(j1[jb].begin(), j1[jb].end(), jq.begin() + 8 * o)); ++jb;
      CHECK(std::equal(j1[jb].begin(), j1[jb].end(), jt.begin() + 6 * o)); ++jb;
      checked += 2;
    }
    CHECK(std::equal(j1[jb].begin(), j1[jb].end(), jx.begin() + 6 * o)); ++jb;
    if (ba.CameraVariable(cm)) {
      for (int row = 0; row < 2; ++row)
        CHECK(std::equal(j1[jb].begin() + row * K, j1[jb].begin() + (row + 1) * K, jc.begin() + (2 * o + row) * PCD_CAM_JAC_STRIDE));
      ++checked;
    } else {
      CHECK(j1[jb][0] == -777.0); ++untouched;                         // NULL Jacobian pointer: nothing written
    }
    ++jb;
  }
  for (size_t l = 0; l < L; ++l) { CHECK(std::equal(j1[jb].begin(), j1[jb].end(), jl.begin() + 3 * l)); ++jb; }
  CHECK_EQ(jb, j1.size());
  CHECK(checked > O && untouched == 0);
  // residual-only evaluation (LM trial step): Evaluate(params, r, NULL)
  std::vector<double> r2; std::vector<std::vector<double>> j2;
  cb.PrepareForEvaluation(false, false);
  CHECK(evaluate_all(false, &r2, &j2) && r2 == res);
  CHECK(!evaluate_all(true, &r2, &j2));                                // Jacobians were not prepared: blocks say so
  // the solver moves the parameters IN PLACE (Reconstruction memory); the next evaluation must see them
  rec.images[2].tvec[0] += 0.05; rec.points3D[7].xyz[2] -= 0.02; rec.cameras[1].params[0] *= 1.001;
  cb.PrepareForEvaluation(true, true);
  std::vector<double> r3; std::vector<std::vector<double>> j3;
  CHECK(evaluate_all(true, &r3, &j3) && r3 != res);
  BundleAdjusterHip fresh(options, config);                            // same problem assembled from the moved state
  fresh.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::WholeMap);
  CHECK(fresh.Create(0));
  std::vector<double> res_fresh(2 * O + L);
  pcd_ba_out of{}; of.residuals = res_fresh.data();
  CHECK_EQ((int)pcd_ba_evaluate(fresh.handle(), &of), (int)PCD_OK);
  CHECK(r3 == res_fresh);
  CHECK_EQ(cb.num_evaluations(), 3u);
  // ---- the recorder route (integration/colmap-pcd-hip.patch): the call sites keep creating the blocks themselves, in
  // their own order -- first half of the reprojection blocks, then the lidar blocks, then the rest, as SetUpLocalByLidar
  // adds the constant points' blocks behind the lidar terms -- and only the addresses of the parameter blocks are noted
  {
    HipBlockRecorder recd;
    std::vector<std::unique_ptr<ceres::CostFunction>> rb(O + L);
    auto add_obs = [&](size_t o) {
      const int im = ba.obs_image_[o], pt = ba.obs_point_[o], cm = ba.image_cam_[im];
      rb[o].reset(recd.AddReprojection(ba.cam_model_[cm], src.Qvec(ba.image_ids_[im]), src.Tvec(ba.image_ids_[im]),
                                       src.XYZ(ba.point_ids_[pt]), src.Params(ba.camera_ids_[cm]), &ba.obs_xy_[2 * o],
                                       ba.image_const_pose_[im] != 0));
    };
    for (size_t o = 0; o < O / 2; ++o) add_obs(o);
    for (size_t l = 0; l < L; ++l)
      rb[O + l].reset(recd.AddLidar(src.XYZ(ba.point_ids_[ba.lidar_point_[l]]), &ba.lidar_abcd_[4 * l], ba.lidar_w_[l]));
    for (size_t o = O / 2; o < O; ++o) add_obs(o);
    CHECK_EQ(recd.NumResiduals(), 2 * O + L);
    CHECK(recd.Finalize(0, /*cameras_variable=*/true));
    recd.PrepareForEvaluation(true, true);
    CHECK(recd.ok());
    cb.PrepareForEvaluation(true, true);                               // the mirror route at the same (moved) state
    for (size_t k = 0; k < blocks.size(); ++k) {
      Block& b = blocks[k];
      double ra[2] = {0, 0}, rbv[2] = {0, 0};
      std::vector<std::vector<double>> ja(b.params.size()), jb2(b.params.size());
      std::vector<double*> pa(b.params.size(), nullptr), pb(b.params.size(), nullptr);
      for (size_t i = 0; i < b.params.size(); ++i) {
        const size_t n = (size_t)b.f->num_residuals() * b.f->parameter_block_sizes()[i];
        ja[i].assign(n, -1.0); jb2[i].assign(n, -2.0);
        if (!b.constant[i]) { pa[i] = ja[i].data(); pb[i] = jb2[i].data(); }
      }
      CHECK(rb[k]->parameter_block_sizes() == b.f->parameter_block_sizes());
      CHECK(b.f->Evaluate(b.params.data(), ra, pa.data()));
      CHECK(rb[k]->Evaluate(b.params.data(), rbv, pb.data()));
      CHECK(ra[0] == rbv[0] && (b.f->num_residuals() == 1 || ra[1] == rbv[1]));
      for (size_t i = 0; i < b.params.size(); ++i)
        if (!b.constant[i]) CHECK(ja[i] == jb2[i]);
    }
    // in-place moves are picked up through the recorded addresses
    rec.points3D[9].xyz[0] += 0.01;
    recd.PrepareForEvaluation(false, true);
    cb.PrepareForEvaluation(false, true);
    double ra[2], rbv[2];
    for (size_t k = 0; k < blocks.size(); ++k) {
      CHECK(blocks[k].f->Evaluate(blocks[k].params.data(), ra, nullptr) && rb[k]->Evaluate(blocks[k].params.data(), rbv, nullptr));
      CHECK(ra[0] == rbv[0]);
    }
    HipBlockRecorder empty;
    CHECK(!empty.Finalize(0, false));                                  // no residuals: Solve returns false (:489-491)
  }
  // run-time switch
  setenv("COLMAP_PCD_HIP", "0", 1);
  CHECK(!HipBackendEnabled());
  setenv("COLMAP_PCD_HIP", "1", 1);
  CHECK(HipBackendEnabled());
  unsetenv("COLMAP_PCD_HIP");
  CHECK(HipBackendEnabled());
  return 0;
}

// ---- one cloud over several shards of this process (include/pcdhip.h "One cloud over several devices") ------------
// the exchange steps as a caller-supplied reduction: device -> host, reduce, host -> device (a production host would
// call RCCL here); every shard sits on device 0 on this box
static int HostMinU64(void*, uint64_t* const* buf, const int* devices, int n, uint64_t count) {
  std::vector<uint64_t> acc(count), tmp(count);
  for (int s = 0; s < n; ++s) {
    if (hipSetDevice(devices[s]) != hipSuccess) return 1;
    if (hipMemcpy(s ? tmp.data() : acc.data(), buf[s], count * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (s) for (uint64_t i = 0; i < count; ++i) acc[i] = std::min(acc[i], tmp[i]);
  }
  for (int s = 0; s < n; ++s) {
    if (hipSetDevice(devices[s]) != hipSuccess) return 1;
    if (hipMemcpy(buf[s], acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
  }
  return 0;
}
static int HostSumI32(void*, int32_t* const* buf, const int* devices, int n, uint64_t count) {
  std::vector<int32_t> acc(count), tmp(count);
  for (int s = 0; s < n; ++s) {
    if (hipSetDevice(devices[s]) != hipSuccess) return 1;
    if (hipMemcpy(s ? tmp.data() : acc.data(), buf[s], count * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (s) for (uint64_t i = 0; i < count; ++i) acc[i] += tmp[i];
  }
  for (int s = 0; s < n; ++s) {
    if (hipSetDevice(devices[s]) != hipSuccess) return 1;
    if (hipMemcpy(buf[s], acc.data(), count * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
  }
  return 0;
}

static int TestGpuShardedCloud() {
  // planar patches + exact duplicates far apart in file order (ties across shard cuts go to the lowest original index)
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> u(0.f, 1.f);
  std::normal_distribution<double> g(0.0, 0.25);
  const size_t n = 40000, Q = 6000;
  std::vector<float> xyz(3 * n), nrm(3 * n);
  for (size_t i = 0; i < n; ++i) {
    const int patch = (int)(i * 8 / n);
    float p[3] = {20.f * u(rng), 6.f * u(rng), 20.f * u(rng)};
    p[patch % 3] = 1.5f * patch;                       // axis-aligned planes
    float nv[3] = {0.f, 0.f, 0.f}; nv[patch % 3] = 1.f;
    for (int k = 0; k < 3; ++k) { xyz[3 * i + k] = p[k]; nrm[3 * i + k] = nv[k]; }
  }
  for (size_t k = 0; k < 300; ++k) {                   // duplicates: row n/2 + 7k copies row 11k
    for (int c = 0; c < 3; ++c) xyz[3 * (n / 2 + 7 * k) + c] = xyz[3 * (11 * k) + c];
  }
  std::vector<double> q(3 * Q);
  for (size_t i = 0; i < Q; ++i) {
    const size_t r = (size_t)(u(rng) * (n - 1));
    for (int k = 0; k < 3; ++k) q[3 * i + k] = (double)xyz[3 * r + k] + (i < 300 ? 0.0 : g(rng));
    if (i < 300) for (int k = 0; k < 3; ++k) q[3 * i + k] = (double)xyz[3 * (11 * i) + k];   // ON a duplicated point
  }
  pcd_cloud_options o;
  pcd_cloud_options_default(&o);
  o.raw_lidar_frame = 0;
  pcd_cloud* single = nullptr;
  CHECK_EQ((int)pcd_cloud_create(xyz.data(), nrm.data(), n, &o, &single), (int)PCD_OK);
  std::vector<uint32_t> i0(Q), i1(Q); std::vector<float> d0(Q), d1(Q); std::vector<uint8_t> f0(Q), f1(Q);
  CHECK_EQ((int)pcd_nn_query(single, q.data(), Q, i0.data(), d0.data(), f0.data()), (int)PCD_OK);
  for (size_t i = 0; i < 300; ++i) CHECK(i0[i] == 11 * i && d0[i] == 0.f);   // the tie goes to the lower row
  std::vector<double> ax(3 * Q), aa(4 * Q), ad(Q), ag(Q), bx(3 * Q), ba(4 * Q), bd(Q), bg(Q);
  std::vector<uint8_t> at(Q), bt(Q);
  const double range = 1.0;
  pcd_assoc_out oa{ax.data(), aa.data(), at.data(), ad.data(), ag.data(), nullptr, nullptr, nullptr};
  CHECK_EQ((int)pcd_associate(single, q.data(), Q, &range, 1, PCD_GATE_MAPPER_LOCAL, &oa), (int)PCD_OK);
  const pcd_shard_reduce host_red{HostMinU64, HostSumI32, nullptr};
  for (int nsh = 2; nsh <= 4; ++nsh) {
    for (int use_cb = 0; use_cb < 2; ++use_cb) {
      const int devs[4] = {0, 0, 0, 0};
      pcd_cloud_shards* sh = nullptr;
      CHECK_EQ((int)pcd_cloud_create_sharded(xyz.data(), nrm.data(), n, &o, devs, nsh, &sh), (int)PCD_OK);
      CHECK_EQ(pcd_cloud_shards_count(sh), nsh);
      CHECK_EQ((size_t)pcd_cloud_shards_size(sh), n);
      const pcd_shard_reduce* red = use_cb ? &host_red : nullptr;
      CHECK_EQ((int)pcd_nn_query_sharded(sh, q.data(), Q, red, i1.data(), d1.data(), f1.data()), (int)PCD_OK);
      CHECK(i1 == i0 && f1 == f0 && std::memcmp(d1.data(), d0.data(), Q * sizeof(float)) == 0);
      pcd_assoc_out ob{bx.data(), ba.data(), bt.data(), bd.data(), bg.data(), nullptr, nullptr, nullptr};
      CHECK_EQ((int)pcd_associate_sharded(sh, q.data(), Q, &range, 1, PCD_GATE_MAPPER_LOCAL, red, &ob), (int)PCD_OK);
      CHECK(bt == at && std::memcmp(bx.data(), ax.data(), 3 * Q * 8) == 0 && std::memcmp(ba.data(), aa.data(), 4 * Q * 8) == 0);
      CHECK(std::memcmp(bd.data(), ad.data(), Q * 8) == 0 && std::memcmp(bg.data(), ag.data(), Q * 8) == 0);
      pcd_cloud_shards_destroy(sh);
    }
  }
  pcd_cloud_destroy(single);
  return 0;
}

static int TestGpu() {
  if (pcd_device_count() < 1) { std::printf("FAIL: --gpu given but no gfx950 device\n"); return 1; }
  // cloud: plane y = 1 (visual frame) on a 5 cm lattice with normal (0,1,0), given in the raw LiDAR frame
  std::vector<float> xyz, nrm;
  for (int i = 0; i < 200; ++i)
    for (int j = 0; j < 200; ++j) {
      const float vx = 0.05f * i, vy = 1.0f, vz = 0.05f * j;           // visual
      xyz.insert(xyz.end(), {vz, -vx, -vy});                           // raw = (z', -x', -y')
      nrm.insert(nrm.end(), {0.f, 0.f, -1.f});                         // visual normal (0,1,0)
    }
  lidar::PointCloudProcess pcp;
  CHECK(pcp.InitializeFromRawCloud(xyz.data(), nrm.data(), xyz.size() / 3));
  CHECK_EQ(pcp.size(), 40000u);
  {  // the same cloud through PointCloudProcess(path).Initialize() (lidar/ply.cc:9-31) from a binary PLY
    const std::string ply = WritePly(true, false, xyz, nrm);
    lidar::PointCloudProcess from_file(ply);
    CHECK(from_file.Initialize());
    CHECK_EQ(from_file.size(), 40000u);
    std::array<double, 3> qq = {2.51, 1.3, 4.02};
    std::array<double, 6> la{}, lb{};
    CHECK(pcp.SearchNearestNeiborByKdtree(qq, la) && from_file.SearchNearestNeiborByKdtree(qq, lb));
    CHECK(la == lb);
    std::remove(ply.c_str());
    lidar::PointCloudProcess missing("/tmp/pcdhip_no_such_map.ply");
    CHECK(!missing.Initialize());
  }
  std::array<double, 3> q = {2.51, 1.3, 4.02};
  std::array<double, 6> l6{};
  CHECK(pcp.SearchNearestNeiborByKdtree(q, l6));
  CHECK(std::fabs(l6[0] - 2.5) < 1e-6 && std::fabs(l6[1] - 1.0) < 1e-6 && std::fabs(l6[2] - 4.0) < 1e-6);
  CHECK(l6[3] == 0.0 && l6[4] == 1.0 && l6[5] == 0.0);
  std::array<double, 3> bad = {NAN, 0, 0};
  std::array<double, 6> keep = {9, 9, 9, 9, 9, 9};
  CHECK(!pcp.SearchNearestNeiborByKdtree(bad, keep));
  CHECK(keep[0] == 9);                                                // out-param untouched on failure
  // batched MatchClosestLidarPoint: ground plane -> IcpGround, gate at 0.25 m rejects the far one
  std::vector<uint64_t> ids = {7, 8, 9};
  std::vector<double> pts = {2.51, 1.3, 4.02, 5.0, 1.2, 5.0, 3.0, 0.9, 3.0};
  std::unordered_map<uint64_t, LidarPoint> maps;
  CHECK(MatchClosestLidarPoints(pcp, ids, pts, {0.25}, PCD_GATE_MAPPER_LOCAL, &maps));
  CHECK_EQ(maps.size(), 2u);
  CHECK(maps.count(7) == 0 && maps.count(8) == 1 && maps.count(9) == 1);
  CHECK(maps[8].type == LidarPointType::IcpGround && maps[8].color[0] == 255 && maps[8].color[2] == 0);
  CHECK(std::fabs(maps[8].abcd[1] - 1.0) < 1e-12 && std::fabs(maps[8].abcd[3] + 1.0) < 1e-6);
  CHECK(std::fabs(maps[8].dist - 0.2) < 1e-6);
  // BA evaluator built from the mirrored assembly
  Reconstruction rec;
  GenerateReconstruction(3, 50, &rec);
  BundleAdjustmentConfig config;
  for (image_t i = 0; i < 3; ++i) config.AddImage(i);
  config.SetConstantPose(0);
  for (point3D_t p = 1; p <= 50; ++p) config.AddVariablePoint(p);
  LidarPoint lp; lp.type = LidarPointType::Icp; lp.abcd = {0, 0, 1, 0.25};
  config.AddLidarPoint(3, lp);
  BundleAdjusterHip ba(BundleAdjustmentOptions(), config);
  ba.SetUp(&rec, BundleAdjusterHip::OptimazePhrase::WholeMap);
  CHECK(ba.Create(0));
  std::vector<double> res(ba.NumResiduals());
  double cost = 0;
  pcd_ba_out out{};
  out.cost = &cost; out.residuals = res.data();
  CHECK_EQ((int)pcd_ba_evaluate(ba.handle(), &out), (int)PCD_OK);
  double s = 0;
  for (double r : res) s += r * r;
  CHECK(std::fabs(cost - 0.5 * s) <= 1e-9 * cost);
  CHECK(cost > 0 && cost < 300 * 8.0 + 1e4);   // +-2 px noise on 300 observations + one lidar term
  return TestGpuProjection() + TestGpuSiftMatcher() + TestGpuSiftBlockMatcher() + TestGpuCeresAdapterEndToEnd() +
         TestGpuShardedCloud();
}

int main(int argc, char** argv) {
  TestConfigNumObservations();
  TestTwoView();
  TestTwoViewConstantCamera();
  TestConstantPoints();
  TestVariableImage();
  TestCameraParameterGroups();
  TestPartiallyContainedTracks();
  TestForceToOptimizePoint();
  TestLidarBlocks();
  TestPlyReader();
  TestPoseReader();
  TestPoseWriter();
  TestMatchVariablePoint();
  TestCeresBlockShapes();
  TestExhaustiveBlocks();
  if (argc > 1 && std::strcmp(argv[1], "--gpu") == 0) g_fail += TestGpu();
  std::printf(g_fail ? "%d FAILED\n" : "ALL OK\n", g_fail);
  return g_fail ? 1 : 0;
}
