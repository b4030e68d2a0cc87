"""Multi-GPU decomposition of the hot path (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on the MI355X node, "gloo" in the CPU tests).  The reference has no distributed code
(SURVEY.md 2.1); the decomposition follows from the structure of the path:

  NN, cloud sharded   rank r indexes rows r, r+N, r+2N, ... of the cloud (global index = base + i*stride,
                      pcd_cloud_options.index_base/stride).  The per-query result is the minimum of the
                      packed key  float_bits(sqdist) << 32 | global_index  -- associative and commutative,
                      so ONE all-reduce(MIN) over the int64 keys (8 B / query) combines the shards
                      bit-exactly, ties included.  The winner's (xyz, normal) is then supplied by its owner:
                      every rank writes the 6 float bit patterns of the keys it owns and zeros elsewhere,
                      one all-reduce(SUM) on int32 reassembles them exactly (x + 0 = x on integers).
  NN, query sharded   cloud replicated (160 MB of 288 GB), queries split: no data-path collective.
  BA                  sharded by track: a rank owns a subset of the 3D points with ALL their observations and
                      LiDAR terms, so point blocks are complete locally; the per-image 6x6 + 6 blocks and the
                      cost are partial sums -> one all-reduce(SUM, f64) of I*42 + 1 doubles.
"""
import numpy as np
import torch
import torch.distributed as dist

KEY_NONE = 0x7FFFFFFFFFFFFFFF


def shard_rows(n, rank, world):
    """interleaved row shard of a cloud: (slice, index_base, index_stride)"""
    return slice(rank, n, world), rank, world


def shard_range(n, rank, world):
    """contiguous [lo, hi) split of n work items (queries)"""
    per = (n + world - 1) // world
    return min(rank * per, n), min((rank + 1) * per, n)


def shard_tracks(scene, rank, world):
    """BA scene -> the sub-scene of the tracks (points) owned by `rank` (points p with p % world == rank),
    with all their observations and LiDAR terms; cameras and images are replicated."""
    P = scene["points"].shape[0]
    own = (np.arange(P) % world) == rank
    new_id = np.cumsum(own) - 1
    osel = own[scene["obs_point"]]
    lsel = own[scene["lidar_point"]] if len(scene.get("lidar_point", [])) else np.zeros(0, bool)
    sub = dict(scene)
    sub["points"] = scene["points"][own]
    sub["obs_image"] = scene["obs_image"][osel]
    sub["obs_point"] = new_id[scene["obs_point"][osel]].astype(np.int32)
    sub["obs_xy"] = scene["obs_xy"][osel]
    if len(lsel):
        sub["lidar_point"] = new_id[scene["lidar_point"][lsel]].astype(np.int32)
        sub["lidar_abcd"] = scene["lidar_abcd"][lsel]
        sub["lidar_weight"] = scene["lidar_weight"][lsel]
    if scene.get("point_const") is not None:
        sub["point_const"] = scene["point_const"][own]
    return sub, np.nonzero(own)[0]


def combine_keys(keys):
    """element-wise MIN of the packed int64 keys over all ranks (RCCL ncclMin); in place"""
    assert keys.dtype == torch.int64
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MIN)
    return keys


def combine_payload(payload):
    """SUM of the int32 bit patterns of the winners' (xyz, normal); exactly one rank is non-zero per query"""
    assert payload.dtype == torch.int32
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(payload, op=dist.ReduceOp.SUM)
    return payload


def combine_blocks(img_blocks, cost):
    """SUM (f64) of the per-image normal-equation blocks [I*36 + I*6] and of the cost"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(img_blocks, op=dist.ReduceOp.SUM)
        dist.all_reduce(cost, op=dist.ReduceOp.SUM)
    return img_blocks, cost


def pack_keys(idx, sqdist, found, index_base=0, index_stride=1):
    """host helper (tests): (idx, float32 sqdist, found) -> packed int64 keys with global indices"""
    gi = index_base + idx.astype(np.uint64) * np.uint64(index_stride)
    k = (sqdist.view(np.uint32).astype(np.uint64) << np.uint64(32)) | gi
    k = np.where(found.astype(bool), k, np.uint64(KEY_NONE))
    return k.astype(np.int64)


def unpack_keys(keys):
    k = np.asarray(keys).astype(np.uint64)
    found = k != np.uint64(KEY_NONE)
    idx = np.where(found, k & np.uint64(0xFFFFFFFF), np.uint64(0xFFFFFFFF)).astype(np.uint32)
    sq = (k >> np.uint64(32)).astype(np.uint32).view(np.float32)
    return idx, sq, found.astype(np.uint8)
