/*
 * pcdhip.h -- C ABI of libpcdhip.so, the MI355X (gfx950) implementation of
 * colmap-pcd's image-to-LiDAR registration hot path.
 *
 * The reference (Wangshihu12/colmap-pcd) has no plugin/FFI API for this
 * path: it is reached through C++ call sites.  Every entry point below names
 * the reference interface it replaces (paths relative to the reference's
 * src/).  C++ adapters with the reference's own signatures live in
 * colmap-pcd_amd/shim/; INTEGRATION.md shows the call-site patch.
 *
 * Conventions
 *   - every function returns a pcd_status (0 = OK) and never throws;
 *   - host pointers unless the name ends in _device; the caller owns all
 *     buffers it passes, the library owns device memory behind the handles;
 *   - a handle is bound to one HIP device and may be used from one thread at
 *     a time; different handles are independent;
 *   - `stream` arguments are hipStream_t passed as void* (NULL = default
 *     stream); _device calls are asynchronous on that stream;
 *   - _device calls are EAGER launches: they grow per-handle scratch, build
 *     per-cloud tables on first use and may synchronise the stream, so they
 *     cannot be recorded into a HIP graph.  A stream that is capturing
 *     (hipStreamBeginCapture, torch.cuda.graph) is refused with
 *     PCD_ERR_UNSUPPORTED before anything is allocated or launched;
 *   - there is NO CPU fallback: without a usable gfx950 device every entry
 *     point that computes returns PCD_ERR_NO_DEVICE.
 */
#ifndef PCDHIP_H_
#define PCDHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCDHIP_VERSION_MAJOR 0
#define PCDHIP_VERSION_MINOR 1

typedef enum {
  PCD_OK = 0,
  PCD_ERR_INVALID = 1,     /* bad argument (null pointer, size mismatch, unknown enum) */
  PCD_ERR_NO_DEVICE = 2,   /* no HIP device / wrong architecture                       */
  PCD_ERR_HIP = 3,         /* a HIP runtime call failed; see pcd_last_error()          */
  PCD_ERR_OOM = 4,
  PCD_ERR_UNSUPPORTED = 5
} pcd_status;

const char* pcd_last_error(void);            /* thread-local message of the last failure */
int pcd_version(void);                       /* major*100 + minor                         */
int pcd_device_count(void);                  /* number of usable gfx950 devices (0 on a CPU box) */

/* ------------------------------------------------------------------------
 * LiDAR cloud index                         replaces lidar/ply.cc:9-57
 *   PointCloudProcess::Initialize -> PointCloudDirectionTrans -> Kdtree::BuildMap
 *   (lidar/kdtree.cc:5-8, pcl::KdTreeFLANN::setInputCloud)
 * --------------------------------------------------------------------- */
typedef struct pcd_cloud pcd_cloud;

typedef enum {
  PCD_LAYOUT_XYZ_NRM = 0,  /* xyz[n][3] floats + nrm[n][3] floats (two arrays)             */
  PCD_LAYOUT_AOS32 = 1     /* lidarpt::Point, lidar/pt_type.h:14-30: x y z pad nx ny nz pad */
} pcd_layout;

typedef struct {
  int32_t device;           /* HIP device ordinal                                           */
  int32_t layout;           /* pcd_layout                                                   */
  int32_t raw_lidar_frame;  /* 1: input rows are in the PLY/LiDAR frame: apply ply.cc:38-54 */
                            /*    (x,y,z)->(-y,-z,x) to position and normal and drop rows   */
                            /*    with a NaN; 0: rows are already transformed and filtered  */
  float cell_size;          /* grid cell edge in metres, 0 = choose from point density      */
  uint32_t index_base;      /* global index of local row i = index_base + i*index_stride:   */
  uint32_t index_stride;    /*    lets N ranks each hold every N-th row of one cloud        */
                            /*    (0 is read as 1).  Requires raw_lidar_frame == 0 if != 1. */
  int32_t reserved[8];
} pcd_cloud_options;

void pcd_cloud_options_default(pcd_cloud_options* o);

/* xyz: n rows (layout XYZ_NRM: 3 floats/row; AOS32: 8 floats/row, nrm ignored). */
pcd_status pcd_cloud_create(const float* xyz, const float* nrm, uint64_t n,
                            const pcd_cloud_options* opts, pcd_cloud** out);
void pcd_cloud_destroy(pcd_cloud* c);

/* rows kept after the NaN filter (index space of every idx returned below) */
uint64_t pcd_cloud_size(const pcd_cloud* c);

typedef struct {
  float cell_size;
  float origin[3];
  int32_t dims[3];           /* fine grid cells per axis                                     */
  int32_t block_dims[3];     /* 4x4x4-cell blocks per axis                                   */
  uint64_t num_indexed;      /* finite points in the grid (Inf rows keep an index, never win) */
  uint64_t occupied_cells;
  double build_ms;
  float bbox_lo[3], bbox_hi[3];   /* tight bounds of the finite rows (meaningful when num_indexed > 0): what    */
                                  /* pcd_nn_refine_device tests queries against, and what a caller uses to send */
                                  /* each query to its home shard                                               */
} pcd_cloud_info;
pcd_status pcd_cloud_get_info(const pcd_cloud* c, pcd_cloud_info* info);

/* copy the transformed / filtered cloud back (tests; GUI display path ply.cc:59-84 wants it) */
pcd_status pcd_cloud_download(const pcd_cloud* c, float* xyz /*[size][3]*/, float* nrm /*[size][3]*/);

/* ------------------------------------------------------------------------
 * One cloud over several devices of this process.
 * What IncrementalMapper::LoadPointcloud / BundleAdjustmentController::LoadPointcloud
 * (sfm/incremental_mapper.cc:194-206, controllers/bundle_adjustment.cc:206-212)
 * would build on a multi-GPU host: the cloud cut into ndev spatially compact
 * shards, one per device; indices everywhere are the post-filter row indices
 * of the WHOLE cloud, and every result equals the single-device one bit for
 * bit (ties -> lowest index, also across shards).
 * Verification status: every test so far has run on hosts with ONE GPU, all
 * shards on device 0.  The code paths that only exist between distinct
 * devices (peer copies of the built-in reduction, per-device streams) are
 * covered by tests/test_sharded_gpu.py::test_shards_on_distinct_devices,
 * which skips itself below two devices -- they have not executed yet.
 * --------------------------------------------------------------------- */
typedef struct pcd_cloud_shards pcd_cloud_shards;
pcd_status pcd_cloud_create_sharded(const float* xyz, const float* nrm, uint64_t n, const pcd_cloud_options* opts,
                                    const int* devices, int ndev, pcd_cloud_shards** out);  /* opts->device unused */
void pcd_cloud_shards_destroy(pcd_cloud_shards* s);
int pcd_cloud_shards_count(const pcd_cloud_shards* s);
uint64_t pcd_cloud_shards_size(const pcd_cloud_shards* s);          /* rows kept after the NaN filter, all shards */
pcd_cloud* pcd_cloud_shards_get(pcd_cloud_shards* s, int shard);    /* borrowed: shard `shard` as a pcd_cloud      */

/* ------------------------------------------------------------------------
 * Nearest neighbour                          replaces lidar/kdtree.cc:10-21
 *   Kdtree::GetClosestPoint  (k = 1 pcl::KdTreeFLANN::nearestKSearch,
 *   FLANN L2_Simple<float> on x,y,z; exact)
 * query = (float)q_xyz (lidar/ply.cc:92).  Ties on the float distance go to
 * the lowest index.  found = 0: empty cloud, non-finite query, or nothing
 * closer than FLT_MAX; idx is then 0xFFFFFFFF and sqdist FLT_MAX.
 * --------------------------------------------------------------------- */
typedef enum {
  PCD_NN_AUTO = 0,        /* PCD_NN_GRID for large batches, PCD_NN_FALLBACK_ONLY for small ones (one launch) */
  PCD_NN_BRUTEFORCE = 1,  /* tiled all-pairs kernel: reference for the others */
  PCD_NN_FALLBACK_ONLY = 2,/* per-wavefront hierarchical search for every query */
  PCD_NN_GRID = 3         /* grid kernels (query sort + brick LDS tiles + exact fallback) whatever the batch size */
} pcd_nn_algo;

pcd_status pcd_nn_query(pcd_cloud* c, const double* q_xyz /*[Q][3]*/, uint64_t Q,
                        uint32_t* idx, float* sqdist, uint8_t* found);
pcd_status pcd_nn_query_algo(pcd_cloud* c, const double* q_xyz, uint64_t Q, int algo,
                             uint32_t* idx, float* sqdist, uint8_t* found);

/* Device-resident form.  keys[i] = (uint64)float_bits(sqdist) << 32 | global_idx,
 * PCD_KEY_NONE when nothing was found.  sqdist is a non-negative float, so the
 * unsigned AND the signed 64-bit order of keys equal the (distance, index)
 * order, and PCD_KEY_NONE is the largest value in both: shards of one cloud
 * combine with an element-wise MIN (RCCL ncclMin on ncclUint64 or ncclInt64). */
#define PCD_KEY_NONE 0x7FFFFFFFFFFFFFFFull
pcd_status pcd_nn_query_device(pcd_cloud* c, const double* d_q_xyz, uint64_t Q, int algo,
                               uint64_t* d_keys, void* stream);

/* Second phase of a search over SHARDS of one cloud (spatially compact shards, one per GPU).
 * d_keys comes in holding the best key found so far for every query (from the query's home shard after a
 * cross-rank MIN; PCD_KEY_NONE = nothing yet) and leaves holding min(incoming, this shard's result).  A query is
 * searched here only if it can still improve: not d_skip[i] (may be NULL; e.g. queries whose home is this shard)
 * and the exact float lower bound of its distance to the shard's bounding box does not exceed its incoming
 * distance (an equal distance is searched: a lower index may hide here).  All other entries are left untouched.
 * The element-wise MIN over ranks of the outputs is the exact single-cloud result, ties included. */
pcd_status pcd_nn_refine_device(pcd_cloud* c, const double* d_q_xyz, uint64_t Q, const uint8_t* d_skip,
                                uint64_t* d_keys, void* stream);

/* ------------------------------------------------------------------------
 * Plane association                          replaces the three serial loops
 *   optim/bundle_adjustment.cc:358-410  BundleAdjustmentConfig::MatchClosestLidarPoint   (PCD_GATE_MAPPER_LOCAL)
 *   sfm/incremental_mapper.cc:1413-1469 IncrementalMapper::AdjustGlobalBundleByLidar     (PCD_GATE_MAPPER_GLOBAL)
 *   controllers/bundle_adjustment.cc:130-185 BundleAdjustmentController::Run             (PCD_GATE_CONTROLLER)
 * each of which does: lidar/ply.cc:90-107 SearchNearestNeiborByKdtree ->
 * lidar/lidar_point.cc:5-50 LidarPoint(l_pt, plane) / Normalize ->
 * classification on the raw normal -> range gate.
 * --------------------------------------------------------------------- */
typedef enum { PCD_GATE_MAPPER_LOCAL = 0, PCD_GATE_MAPPER_GLOBAL = 1, PCD_GATE_CONTROLLER = 2 } pcd_gate_mode;
typedef enum { PCD_LIDAR_NONE = 0, PCD_LIDAR_ICP = 1, PCD_LIDAR_ICP_GROUND = 2 } pcd_lidar_type;
/* OR-ed into gate_mode: bound the search by the gate.  All three call sites drop an association whose
 * point-to-point distance exceeds max_range (mapper) / 2 m (controller), so no neighbour beyond that distance can be
 * recorded; with this flag the search prunes there (exact inside the bound), the set of recorded associations and
 * every field of their rows are identical, and rows with type 0 carry zeros / 0xFFFFFFFF instead of the rejected
 * winner: in bounded mode nn_idx / nn_sqdist are DEFINED ONLY FOR ACCEPTED ASSOCIATIONS (type != 0) -- a query whose
 * gate rejects it reports 0xFFFFFFFF / FLT_MAX, not its true nearest neighbour.  Ignored when keys are passed in.
 * pcd_associate_staged always searches this way (it returns only the recorded associations; the flag is accepted
 * and redundant there); pcd_nn_query* is never bounded. */
#define PCD_GATE_BOUNDED_SEARCH 0x100

typedef struct {
  double* lidar_xyz;   /* [Q][3] LidarPoint::LidarXYZ()  (winner position as doubles)        */
  double* abcd;        /* [Q][4] LidarPoint::LidarABCD() after Normalize()                   */
  uint8_t* type;       /* [Q]    pcd_lidar_type; 0 = the reference records no LidarPoint     */
  double* dist;        /* [Q]    point-to-point distance (SetDist at bundle_adjustment.cc:403) */
  double* angle;       /* [Q]    ComputeAngle (SetAngle at :404)                              */
  double* dist2plane;  /* [Q]    ComputeDist, may be NULL                                     */
  uint32_t* nn_idx;    /* [Q]    winner index, may be NULL                                    */
  float* nn_sqdist;    /* [Q]    may be NULL                                                  */
} pcd_assoc_out;

/* max_range: Q entries (per-point schedule, sfm/incremental_mapper.cc:1159-1163) or
 * one entry broadcast when max_range_count == 1; ignored for PCD_GATE_CONTROLLER. */
pcd_status pcd_associate(pcd_cloud* c, const double* q_xyz, uint64_t Q, const double* max_range,
                         uint64_t max_range_count, int gate_mode, const pcd_assoc_out* out);

/* Staged host form for the call sites (the drop-in's PCIe path): inputs and results live in PINNED host memory
 * owned by the cloud handle, and only the associations the reference would have recorded come back, as one
 * 80-byte record each, in ascending query order:
 *   pcd_assoc_staging   pinned input buffers for Q queries (valid until the next staging call or destroy); the
 *                       caller gathers Point3D::XYZ() / the range schedule straight into them
 *   pcd_associate_staged  H2D (24 B / query) -> search + epilogue -> device-side compaction of type != 0 ->
 *                       D2H of num_hits records; *hits points into pinned memory valid until the next call
 * (pcd_associate with caller-owned pageable arrays of 89 B / query stays for convenience: it is PCIe- and
 * staging-copy-bound, about 10x the device time.) */
typedef struct {
  double lidar_xyz[3];   /* LidarPoint::LidarXYZ()                    */
  double abcd[4];        /* LidarPoint::LidarABCD() after Normalize() */
  double dist, angle;    /* SetDist / SetAngle                        */
  uint32_t query;        /* row of q_xyz this association belongs to  */
  uint8_t type;          /* PCD_LIDAR_ICP | PCD_LIDAR_ICP_GROUND      */
  uint8_t pad[3];
} pcd_assoc_hit;
pcd_status pcd_assoc_staging(pcd_cloud* c, uint64_t Q, double** q_xyz /*[Q][3]*/, double** max_range /*[Q]*/);
pcd_status pcd_associate_staged(pcd_cloud* c, uint64_t Q, uint64_t max_range_count, int gate_mode,
                                const pcd_assoc_hit** hits, uint64_t* num_hits);

/* Device form: every pointer in `out` and d_q_xyz / d_max_range are device
 * pointers.  d_keys_in == NULL: run the search; otherwise use these keys
 * (e.g. after a cross-rank MIN) and skip the search.  On a sharded cloud
 * (index_stride != 1 or index_base != 0) a key whose index this shard does not
 * own yields type 0 here: the owner associates it, or use the payload path. */
pcd_status pcd_associate_device(pcd_cloud* c, const double* d_q_xyz, uint64_t Q, const double* d_max_range,
                                uint64_t max_range_count, int gate_mode, const uint64_t* d_keys_in,
                                const pcd_assoc_out* d_out, void* stream);

/* Sharded clouds: after the MIN over ranks each rank fills winner (xyz, normal)
 * for the keys it owns and zeros elsewhere: d_payload [Q][6] floats as int32 bit
 * patterns, so that a SUM over ranks reassembles them bit-exactly.
 * pcd_associate_from_payload_device then runs the epilogue on any rank. */
pcd_status pcd_nn_winner_payload_device(pcd_cloud* c, const uint64_t* d_keys, uint64_t Q,
                                        int32_t* d_payload, void* stream);
pcd_status pcd_associate_from_payload_device(int device, const double* d_q_xyz, uint64_t Q,
                                             const double* d_max_range, uint64_t max_range_count,
                                             int gate_mode, const uint64_t* d_keys, const int32_t* d_payload,
                                             const pcd_assoc_out* d_out, void* stream);

/* Post-BA outlier filter of the associations       replaces base/reconstruction.cc:771-805
 *   Reconstruction::FilterLidarOutlier: erase[i] = 1 when ||lidar_xyz[i] - points_xyz[i]|| exceeds
 *   max_proj_dist_error (type PCD_LIDAR_PROJ) or max_icp_dist_error (Icp / IcpGround); type 0 rows are
 *   left alone.  Device pointers (the arrays normally still live in HBM after BA). */
#define PCD_LIDAR_PROJ 3   /* LidarPointType::Proj: depth-projection associations (pcd_proj_*) */
pcd_status pcd_filter_lidar_outlier_device(int device, const double* d_points_xyz, const double* d_lidar_xyz,
                                           const uint8_t* d_type, uint64_t n, double max_proj_dist_error,
                                           double max_icp_dist_error, uint8_t* d_erase, void* stream);

/* ------------------------------------------------------------------------
 * Sharded cloud (pcd_cloud_create_sharded above): search and association
 * --------------------------------------------------------------------- */
/* The exchange steps of the sharded search are reductions over the shards' device buffers: buf[s] lives on
 * devices[s]; on return EVERY buf[s] must hold the element-wise result.  A C++ host plugs RCCL here
 * (ncclGroupStart; ncclAllReduce(buf[s], buf[s], count, ncclUint64, ncclMin, comm[s], 0) per shard; ncclGroupEnd),
 * tests an in-process loop.  NULL callbacks (or a NULL pcd_shard_reduce): the library reduces with peer copies to the
 * first shard's device.  All devices are idle (synchronised) when a callback is entered; return 0 for success. */
typedef struct {
  int (*min_u64)(void* user, uint64_t* const* buf, const int* devices, int nshards, uint64_t count);
  int (*sum_i32)(void* user, int32_t* const* buf, const int* devices, int nshards, uint64_t count);
  void* user;
} pcd_shard_reduce;

/* pcd_nn_query / pcd_associate over the shards: home-shard search, MIN over shards, refinement of the queries a foreign
 * shard's bounding box cannot rule out, MIN again (+ the winners' rows from their owners, SUM over shards). */
pcd_status pcd_nn_query_sharded(pcd_cloud_shards* s, const double* q_xyz, uint64_t Q, const pcd_shard_reduce* red,
                                uint32_t* idx, float* sqdist, uint8_t* found);
pcd_status pcd_associate_sharded(pcd_cloud_shards* s, const double* q_xyz, uint64_t Q, const double* max_range,
                                 uint64_t max_range_count, int gate_mode, const pcd_shard_reduce* red,
                                 const pcd_assoc_out* out);

/* ------------------------------------------------------------------------
 * Depth-projection association (LidarPointType::Proj)   replaces lidar/pcd_projection.{h,cc} `PcdProj`
 *   pcd_proj_create          <- PcdProj::PcdProj + BuildSubMap          pcd_projection.h:52, .cc:223-255
 *   pcd_proj_set_new_images  <- both PcdProj::SetNewImage overloads     .cc:13-89, .cc:102-220
 *                               (SearchSubMap .cc:258-297, SearchImageMap .cc:499-559,
 *                                ImageMapProj .cc:305-468, DistortOpenCV .cc:561-594), for a batch of images.
 * The winner of a feature pixel is the LiDAR point with the smallest float camera-frame norm among the points
 * whose splat covers the pixel; equal norms resolve to the point the reference's single-thread walk meets first
 * (submap key order, then cloud row) -- the reference's OpenMP loop is racy there, this is deterministic.
 * The cloud handle must outlive the projector and must hold the whole cloud (no shard).
 * --------------------------------------------------------------------- */
typedef struct pcd_proj pcd_proj;
typedef struct {                 /* lidar/pcd_projection.h:31-47, numeric members */
  double depth_image_scale;      /* 0.2 */
  int32_t max_proj_scale;        /* 10  */
  int32_t min_proj_scale;        /* 2   */
  double min_proj_dist;          /* 2   */
  float submap_length;           /* x, 1.0 */
  float submap_width;            /* z, 1.0 */
  float submap_height;           /* y, 1.0 */
  float choose_meter;            /* 40  */
  double min_lidar_proj_dist;    /* no default in the reference; controllers/incremental_mapper.cc:345 */
} pcd_proj_options;
typedef struct {
  double qvec[4], tvec[3];       /* Image::Qvec (w,x,y,z; used un-normalised as the reference does), Tvec */
  double params[8];              /* fx fy cx cy k1 k2 p1 p2: the reference reads an OPENCV camera
                                    unconditionally (.cc:21, .cc:561-571); zero-fill what a model lacks */
  uint64_t width, height;        /* Camera::Width / Height (full resolution) */
  uint64_t feat_begin, feat_end; /* this image's rows of feat_xy */
} pcd_proj_image;
void pcd_proj_default_options(pcd_proj_options* o);
pcd_status pcd_proj_create(pcd_cloud* cloud, const pcd_proj_options* options, pcd_proj** out);
void pcd_proj_destroy(pcd_proj* p);
uint64_t pcd_proj_num_submaps(const pcd_proj* p);
uint64_t pcd_proj_last_pairs(const pcd_proj* p);   /* (image, submap) pairs the last call projected */
/* The splat half-width beyond min_proj_dist is a_x*depth+b_x (and y): four function-local `static`s in the
 * reference (.cc:391-397), initialised from the FIRST camera the process projects with and never again.
 * Here they latch per projector on the first image of the first call; set=1 overrides them (coeffs4 = a_x, b_x,
 * a_y, b_y), set=0 reads them back.  *latched (optional) reports whether they are fixed yet. */
pcd_status pcd_proj_scale_coeffs(pcd_proj* p, int set, double* coeffs4, int* latched);
/* Host buffers.  feat_xy: [n_feat][2] full-resolution pixel coordinates (for overload #1 pass the Point2D's that
 * have a 3D point; for #2 every pt_xy).  Any output may be NULL:
 *   found[n_feat]        1 when a LiDAR point was matched (pt_xy.second of overload #2)
 *   lidar_index[n_feat]  cloud row of the winner (0xFFFFFFFF when none), dist[n_feat] its float norm
 *   lidar6[n_feat][6]    overload #1 value: x y z nx ny nz as doubles (zeros when none)
 *   cam_xyz[n_feat][3]   overload #2 value: pixel ray intersected with the winner's plane (zeros when none) */
pcd_status pcd_proj_set_new_images(pcd_proj* p, uint64_t n_images, const pcd_proj_image* images, uint64_t n_feat,
                                   const double* feat_xy, uint8_t* found, uint32_t* lidar_index, float* dist,
                                   double* lidar6, double* cam_xyz);

/* search-radius schedule, sfm/incremental_mapper.cc:1159-1163, 1423-1427 */
pcd_status pcd_search_range_schedule(const int32_t* global_opt_num, uint64_t n, double kd_max,
                                     double kd_min, double drop_speed, double* out);

/* ------------------------------------------------------------------------
 * Bundle-adjustment residual / Jacobian evaluation
 *   replaces what ceres::Solve calls per iteration on the problem built by
 *   optim/bundle_adjustment.cc:694-1131: for every residual block
 *   ceres::CostFunction::Evaluate(parameters, residuals, jacobians) of
 *     base/cost_functions.h:49-141   BundleAdjustmentCostFunction<Model>            (variable pose)
 *     base/cost_functions.h:256-370  BundleAdjustmentConstantPoseCostFunction<Model>
 *     base/cost_functions.h:150-241  BundleAdjustmentLidarCostFunction
 *   with base/camera_models.h WorldToImage of the 11 models, followed by the
 *   loss correction (optim/bundle_adjustment.cc:53-68) and the quaternion /
 *   subset manifolds (base/cost_functions.h:610-627).
 * --------------------------------------------------------------------- */
typedef struct pcd_ba pcd_ba;

typedef enum { PCD_LOSS_TRIVIAL = 0, PCD_LOSS_SOFT_L1 = 1, PCD_LOSS_CAUCHY = 2 } pcd_loss_type;

/* model ids = CameraModel::kModelId, base/camera_models.h:187-347 */
typedef enum {
  PCD_CAM_SIMPLE_PINHOLE = 0, PCD_CAM_PINHOLE = 1, PCD_CAM_SIMPLE_RADIAL = 2, PCD_CAM_RADIAL = 3,
  PCD_CAM_OPENCV = 4, PCD_CAM_OPENCV_FISHEYE = 5, PCD_CAM_FULL_OPENCV = 6, PCD_CAM_FOV = 7,
  PCD_CAM_SIMPLE_RADIAL_FISHEYE = 8, PCD_CAM_RADIAL_FISHEYE = 9, PCD_CAM_THIN_PRISM_FISHEYE = 10
} pcd_camera_model;
int pcd_camera_num_params(int model_id);   /* -1 for an unknown id */
/* Parameter groups of a model (base/camera_models.h FocalLengthIdxs / PrincipalPointIdxs / ExtraParamsIdxs):
 * group[k] = 0 focal length, 1 principal point, 2 extra (distortion) parameter, for k < num_params.
 * BundleAdjuster::ParameterizeCameras (optim/bundle_adjustment.cc:1047-1100) holds a group constant unless
 * refine_focal_length / refine_principal_point / refine_extra_params is set. */
pcd_status pcd_camera_param_groups(int model_id, uint8_t* group /*[num_params]*/);

/* Flat problem description (all host pointers, copied at create).
 * It is what BundleAdjuster::SetUp*ByLidar assembles block by block:
 *   cameras  <- Camera::ParamsData()            optim/bundle_adjustment.cc:828
 *   images   <- Image::Qvec()/Tvec()            :825-826   (qw qx qy qz tx ty tz)
 *   points   <- Point3D::XYZ()                  :894
 *   obs      <- one per AddResidualBlock of a reprojection functor :875,:893,:982
 *   lidar    <- one per AddLidarToProblem       :993-1040 (abcd NaN rows must be dropped by the caller
 *               exactly as :1005-1009 does; weight chosen by type :1013-1028)
 * image_const_pose[i] = !refine_extrinsics || HasConstantPose(i) (:831) or the
 * image is outside the config (:967-983): such blocks use the constant-pose functor.
 * image_const_tvec[i] bit k = tvec[k] held constant (SetSubsetManifold :912-915).
 * point_const[p] = ParameterizePoints (:1107-1131).
 * camera_refine = what ParameterizeCameras (:1047-1100) decides: one byte per entry of cam_params, 1 = the
 * parameter is optimised, 0 = held constant (SetParameterBlockConstant for a whole camera, SubsetManifold for
 * part of one).  NULL = all intrinsics constant (this fork's default, ba_refine_* = false,
 * controllers/incremental_mapper.h:156-158): the camera outputs of pcd_ba_out are then zero. */
typedef struct {
  int32_t device;
  int32_t num_cameras;
  const int32_t* cam_model;        /* [C]                        */
  const int32_t* cam_param_offset; /* [C] start in cam_params    */
  const double* cam_params;        /* packed                     */
  uint64_t cam_params_len;
  int32_t num_images;
  const double* poses;             /* [I][7]                     */
  const int32_t* image_camera;     /* [I]                        */
  const uint8_t* image_const_pose; /* [I] or NULL (all variable) */
  const uint8_t* image_const_tvec; /* [I] or NULL                */
  int32_t num_points;
  const double* points;            /* [P][3]                     */
  const uint8_t* point_const;      /* [P] or NULL                */
  uint64_t num_obs;
  const int32_t* obs_image;        /* [O]                        */
  const int32_t* obs_point;        /* [O]                        */
  const double* obs_xy;            /* [O][2]                     */
  uint64_t num_lidar;
  const int32_t* lidar_point;      /* [L]                        */
  const double* lidar_abcd;        /* [L][4]                     */
  const double* lidar_weight;      /* [L]                        */
  int32_t loss_type;               /* pcd_loss_type              */
  double loss_scale;
  const uint8_t* camera_refine;    /* [cam_params_len] or NULL   */
  int32_t reserved[8];
} pcd_ba_desc;

pcd_status pcd_ba_create(const pcd_ba_desc* desc, pcd_ba** out);
void pcd_ba_destroy(pcd_ba* ba);

/* parameter update between iterations (host -> device); NULL keeps the old values */
pcd_status pcd_ba_set_parameters(pcd_ba* ba, const double* poses /*[I][7]*/, const double* points /*[P][3]*/);
/* refined intrinsics: the camera parameter blocks change between iterations too (same layout as desc.cam_params) */
pcd_status pcd_ba_set_camera_parameters(pcd_ba* ba, const double* cam_params /*[cam_params_len]*/);

/* Outputs of one evaluation; every pointer may be NULL (not computed / not copied).
 * Raw blocks are exactly what each CostFunction::Evaluate hands to Ceres
 * (row-major num_residuals x block_size, ambient quaternion size 4):
 *   residuals [2*O + L]   obs blocks first, then lidar blocks
 *   jac_q [O][2][4]  jac_t [O][2][3]  jac_X [O][2][3]   (zero rows for constant-pose blocks)
 *   jac_lidar [L][3]
 *   jac_cam [O][2][PCD_CAM_JAC_STRIDE]   the camera-parameter block (2 x K row-major in the first K columns of
 *             each row, K = pcd_camera_num_params(model), remaining columns zero): what autodiff returns when
 *             refine_focal_length / refine_principal_point / refine_extra_params (optim/bundle_adjustment.h:76-81)
 *             leave the camera block variable; Ceres applies its SubsetManifold for the constant subset itself.
 * Normal-equation blocks use the loss-corrected residuals/Jacobians projected on
 * the manifolds (pose tangent = 3 quaternion-tangent + 3 tvec):
 *   H_img [I][6][6] g_img [I][6]   H_pt [P][3][3] g_pt [P][3]   W [O][6][3] = Jp^T JX */
#define PCD_CAM_JAC_STRIDE 12   /* widest camera model (FULL_OPENCV, THIN_PRISM_FISHEYE) */
typedef struct {
  double* cost;        /* [1]  1/2 sum rho(||r_block||^2).  Fixed-order sums, no atomics: bitwise reproducible
                          run to run.  Requested without H_pt / g_pt it is summed per observation (the cheap
                          LM trial-step pass), with them per track: the two agree up to summation order. */
  double* residuals;
  double* jac_q;
  double* jac_t;
  double* jac_X;
  double* jac_lidar;
  double* H_img;
  double* g_img;
  double* H_pt;
  double* g_pt;
  double* W;
  double* jac_cam;     /* [O][2][PCD_CAM_JAC_STRIDE] d r / d camera params, see above */
  /* Camera blocks of the normal equations (refined intrinsics, desc.camera_refine): loss-corrected, columns of
   * constant parameters zero, S = PCD_CAM_JAC_STRIDE rows / columns per camera (entries >= K zero):
   *   H_cam [C][S][S] = sum Jc^T Jc    g_cam [C][S] = sum Jc^T r      over the observations of the camera's images
   *   E_cam [I][S][6] = sum Jc^T Jp    camera x pose-tangent coupling of each image (an image has one camera)
   *   W_cam [O][S][3] = Jc^T JX        camera x point coupling of each observation
   * Together with H_img, g_img, H_pt, g_pt, W this is the full J^T J / J^T r of the problem in block form. */
  double* H_cam;
  double* g_cam;
  double* E_cam;
  double* W_cam;
} pcd_ba_out;

pcd_status pcd_ba_evaluate(pcd_ba* ba, const pcd_ba_out* out);                 /* host outputs   */

/* Post-BA filters reduced PER TRACK on the device, from the parameters resident after BA:
 *   Reconstruction::FilterPoints3DWithLargeReprojectionError   base/reconstruction.cc:1662-1712
 *   Reconstruction::FilterObservationsWithNegativeDepth        base/reconstruction.cc:837-855
 *   Reconstruction::ComputeMeanReprojectionError               base/reconstruction.cc:906-921
 * A track = all observations of a point in this problem (AddPointToProblem adds the observations from images outside
 * the config too, optim/bundle_adjustment.cc:927-985, so for the points of a BA config that is the whole track).
 * Per point: track shorter than 2 -> deleted; an element whose squared error (normalised quaternion, DBL_MAX behind
 * the camera, base/projection.cc:104-117) exceeds max_reproj_error^2 is erased; if at most one element would survive
 * the point is deleted with all its elements; otherwise Point3D::SetError(sum of sqrt(error) of the kept elements /
 * their number).  Sums run in ascending observation index.  Every pointer may be NULL.  The erase / bookkeeping on
 * the Reconstruction itself stays with the caller. */
typedef struct {
  uint8_t* obs_erase;            /* [O] 1: DeleteObservation (or its point is deleted)                       */
  uint8_t* obs_negative_depth;   /* [O] 1: !HasPointPositiveDepth (depth < eps)                              */
  uint8_t* point_delete;         /* [P] 1: DeletePoint3D                                                     */
  double* point_error;           /* [P] Point3D::Error() after the filter; -1 = HasError() false (deleted)   */
  double* summary;               /* [4] num_filtered (the function's return value), mean reprojection error  */
                                 /*     over the points with an error (0 if none), number of such points,     */
                                 /*     number of observations with negative depth                            */
} pcd_ba_filter_out;
pcd_status pcd_ba_filter_tracks_device(pcd_ba* ba, double max_reproj_error, const pcd_ba_filter_out* d_out, void* stream);
pcd_status pcd_ba_filter_tracks(pcd_ba* ba, double max_reproj_error, const pcd_ba_filter_out* out);   /* host outputs */

/* The Ceres route (optim/bundle_adjustment.cc:858-893, :967-983, :1031-1037: every residual block's Evaluate copies
 * its rows): the raw blocks of ALL residual blocks land in PINNED host buffers owned by the handle -- one device pass,
 * then one asynchronous copy per array at the pinned PCIe rate instead of pcd_ba_evaluate's synchronous copies into
 * the caller's pageable memory.  Constant-pose blocks (the reference's BundleAdjustmentConstantPoseCostFunction has
 * no pose parameter blocks) carry no pose Jacobians: jac_q / jac_t hold one row per VARIABLE-pose observation, packed
 * on the device, and pose_row[o] is observation o's row (0xFFFFFFFF: none).  The pointers stay valid until the next
 * pcd_ba_evaluate_blocks / pcd_ba_destroy on the handle. */
typedef struct {
  const double* residuals;   /* [2*O + L]                                                       */
  const double* jac_q;       /* [num_pose_rows][2][4]   NULL when want_jacobians == 0           */
  const double* jac_t;       /* [num_pose_rows][2][3]                                           */
  const double* jac_X;       /* [O][2][3]                                                       */
  const double* jac_lidar;   /* [L][3]                                                          */
  const double* jac_cam;     /* [O][2][PCD_CAM_JAC_STRIDE] or NULL (want_jac_cam == 0)          */
  const uint32_t* pose_row;  /* [O] host array (fixed at pcd_ba_create)                         */
  uint64_t num_pose_rows;
  uint64_t bytes_d2h;        /* bytes that crossed PCIe device -> host in this call             */
} pcd_ba_blocks;
pcd_status pcd_ba_evaluate_blocks(pcd_ba* ba, int want_jacobians, int want_jac_cam, pcd_ba_blocks* out);
pcd_status pcd_ba_evaluate_device(pcd_ba* ba, const pcd_ba_out* d_out, void* stream);  /* device outputs */
/* Inputs of the post-BA filters, per observation (either pointer may be NULL):
 *   sq_err[o] = CalculateSquaredReprojectionError (base/projection.cc:104-117), DBL_MAX when the point is not
 *               in front of the camera -- what FilterPoints3DWithLargeReprojectionError
 *               (base/reconstruction.cc:1662-1700) and ComputeMeanReprojectionError (:906) consume;
 *   depth[o]  = z of the point in the camera frame -- FilterObservationsWithNegativeDepth (:837-855) deletes
 *               observations with depth < DBL_EPSILON. */
pcd_status pcd_ba_observation_errors(pcd_ba* ba, double* sq_err /*[O]*/, double* depth /*[O]*/);
pcd_status pcd_ba_observation_errors_device(pcd_ba* ba, double* d_sq_err, double* d_depth, void* stream);
/* device-side parameter pointers for zero-copy updates: [I][7] and [P][3] doubles */
pcd_status pcd_ba_device_parameters(pcd_ba* ba, double** d_poses, double** d_points);

/* ------------------------------------------------------------------------
 * Exact SIFT descriptor matching (stretch row a19)
 *   replaces feature/sift.cc:1041-1054 MatchSiftFeaturesCPUBruteForce =
 *     feature/sift.cc:171-204 ComputeSiftDistanceMatrix (int32 dot of uint8 x 128) +
 *     feature/sift.cc:55-144  FindBestMatches[OneWay]BruteForce (ratio / distance tests, cross check)
 *   -- the exact result the reference's GPU path approximates with lib/SiftGPU SiftMatchGPU::GetSiftMatch
 *   (lib/SiftGPU/SiftGPU.h:268-352).  Defaults of SiftMatchingOptions (feature/sift.h:121-137):
 *   max_ratio 0.8, max_distance 0.7, cross_check 1.
 * desc: [n][128] uint8 row-major.  matches: [min(n1, ...)<= n1][2] = (point2D_idx1, point2D_idx2) in
 * ascending idx1, exactly the order of the reference's FeatureMatches vector.
 * --------------------------------------------------------------------- */
pcd_status pcd_sift_match(int device, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, float max_ratio,
                          float max_distance, int cross_check, uint32_t* matches /*[n1][2]*/, int32_t* num_matches);
/* device form: m12 [n1] / m21 [n2] receive the one-way results (-1 = none) as well */
pcd_status pcd_sift_match_device(int device, const uint8_t* d_desc1, int n1, const uint8_t* d_desc2, int n2,
                                 float max_ratio, float max_distance, int cross_check, int32_t* d_m12,
                                 int32_t* d_m21, uint32_t* d_matches, int32_t* d_num_matches, void* stream);

/* Many image pairs in one call: replaces feature/matching.cc:798 SiftFeatureMatcher::Match(image_pairs), which
 * ExhaustiveFeatureMatcher::Run (:902-960) calls once per block of up to block_size^2 pairs and whose workers
 * (:358-380 CPU, :403-440 GPU) run MatchSiftFeaturesCPU / MatchSiftFeaturesGPU pair by pair.
 * All descriptors live in one arena of 128-byte rows: image i owns rows [first_row[i], first_row[i+1]).
 * `pairs` is n_pairs x {image1, image2}.  Per pair the result is exactly pcd_sift_match(image1, image2).
 * Device form: first_row / pairs / match_offset are HOST arrays (read before the call returns), d_arena,
 * d_matches and d_counts device memory.  Pair p's list goes to d_matches + 2 * match_offset[p] and must have
 * room for n(image1) matches; d_counts[p] = its length.  Three launches per sub-batch of pairs (scores,
 * finalize, cross check + compaction) on `stream`; the call does not wait for them.
 * Host form: uploads the arena, returns the lists back to back in `matches` (list p =
 * matches[2 * list_offset[p] .. 2 * list_offset[p + 1])); PCD_ERR_INVALID if they exceed matches_capacity
 * (counted in matches; list_offset is still filled so the caller can size the buffer and call again). */
pcd_status pcd_sift_match_batch_device(int device, const uint8_t* d_arena, const uint64_t* first_row /*[n_images+1]*/,
                                       int n_images, const uint32_t* pairs /*[n_pairs][2]*/, int n_pairs,
                                       float max_ratio, float max_distance, int cross_check, uint32_t* d_matches,
                                       const uint64_t* match_offset /*[n_pairs]*/, int32_t* d_counts /*[n_pairs]*/,
                                       void* stream);
pcd_status pcd_sift_match_batch(int device, const uint8_t* arena, const uint64_t* first_row /*[n_images+1]*/, int n_images,
                                const uint32_t* pairs /*[n_pairs][2]*/, int n_pairs, float max_ratio, float max_distance,
                                int cross_check, uint32_t* matches, uint64_t matches_capacity,
                                uint64_t* list_offset /*[n_pairs+1]*/);

/* Matcher object with the shape of lib/SiftGPU's SiftMatchGPU (lib/SiftGPU/SiftGPU.h:268-352) as the reference
 * drives it in feature/sift.cc:1227-1266 MatchSiftFeaturesGPU: two descriptor slots stay on the device, so
 * matching image i against many j uploads i once.  set_descriptors clips num to max_sift like
 * SiftMatchGPU::SetDescriptors; match = GetSiftMatch(max_match, match_buffer, distmax, ratiomax,
 * mutual_best_match) and returns the exact brute-force result (see above). */
typedef struct pcd_sift_matcher pcd_sift_matcher;
pcd_status pcd_sift_matcher_create(int device, int max_sift, pcd_sift_matcher** out);
void pcd_sift_matcher_destroy(pcd_sift_matcher* m);
pcd_status pcd_sift_matcher_set_max_sift(pcd_sift_matcher* m, int max_sift);
pcd_status pcd_sift_matcher_set_descriptors(pcd_sift_matcher* m, int index /*0|1*/, int num, const uint8_t* desc);
pcd_status pcd_sift_matcher_match(pcd_sift_matcher* m, int max_match, uint32_t* matches /*[max_match][2]*/,
                                  float distmax, float ratiomax, int mutual_best_match, int32_t* num_matches);

/* ------------------------------------------------------------------------
 * Profiling hooks used by bench.py (HIP events on the launch stream)
 * --------------------------------------------------------------------- */
typedef struct {
  char name[48];
  uint64_t launches;
  double total_ms;
} pcd_kernel_time;
pcd_status pcd_profile_enable(int on);
pcd_status pcd_profile_only(const char* scope);   /* time just this scope (NULL / "": all): two HIP events per timed
                                                      scope cost a few microseconds of stream time each */
pcd_status pcd_profile_reset(void);
/* fills up to cap entries, returns the number of distinct kernels in *count (syncs the device) */
pcd_status pcd_profile_get(pcd_kernel_time* entries, int cap, int* count);

/* Not declared here on purpose: the tuning hooks bench.py / tools use (pcd_nn_set_tuning, pcd_nn_set_search,
 * pcd_nn_set_brick_shift, pcd_nn_set_bookkeeping) set PROCESS-GLOBAL state -- they are for experiments on one handle at
 * a time, not part of the ABI, and the thread-safety statement at the top of this header does not cover them. */

/* statistics of the last pcd_nn_query*(…, PCD_NN_AUTO) call on this handle */
typedef struct {
  uint64_t queries;
  uint64_t brick_groups;        /* wavefront work items of the brick kernel                 */
  uint64_t staged_points;       /* sum over groups of points staged through LDS             */
  uint64_t fallback_queries;    /* queries finished by the exact hierarchical kernel        */
  uint64_t fallback_points;     /* points scanned by that kernel                            */
  uint64_t pair_evals;          /* distance evaluations, both kernels                       */
} pcd_nn_stats;
pcd_status pcd_nn_last_stats(pcd_cloud* c, pcd_nn_stats* s);   /* syncs */

#ifdef __cplusplus
}
#endif
#endif /* PCDHIP_H_ */
