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
  NN, compact shards  north_star's variant done so that it scales: the cloud is put in a spatial order once
                      (compact_order: coarse cells, z-major) and cut into N equal-count, contiguous row ranges
                      (index_base = first row, stride 1), so a shard is a slab of space with its own dense grid.
                      Two phases, two all-reduce(MIN) of 8 B / query:
                        1. every query is searched in its HOME shard only (nearest bounding box) -- Q/N queries
                           per rank against N/N-th of the cloud;
                        2. pcd_nn_refine_device: a rank searches the foreign queries whose current distance does
                           not rule its bounding box out (exact float bound), i.e. queries near a cut and the
                           far outliers; everything else is skipped.
                      The MIN of the keys is the exact single-cloud result, ties included.
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


def compact_order(xyz, cell=1.0):
    """permutation that puts the cloud in a spatially compact order: coarse cells of `cell` metres, z-major then y
    then x (stable inside a cell).  Row ranges of the permuted cloud are slabs of space.  Host-side, once per map."""
    x = np.asarray(xyz, np.float64)
    fin = np.isfinite(x).all(axis=1)
    lo = x[fin].min(axis=0) if fin.any() else np.zeros(3)
    c = np.floor((np.where(fin[:, None], x, lo) - lo) / cell).astype(np.int64)
    nx, ny = int(c[:, 0].max()) + 1, int(c[:, 1].max()) + 1
    return np.argsort((c[:, 2] * ny + c[:, 1]) * nx + c[:, 0], kind="stable")


def shard_cuts(n, world):
    """equal-count contiguous row ranges: cuts[r] .. cuts[r+1]"""
    return [(n * r) // world for r in range(world + 1)]


def home_shards(q, bbox_lo, bbox_hi):
    """home shard of every query = the shard whose bounding box is nearest (ties: lowest rank).  Any assignment is
    correct -- phase 2 searches every shard a query could still improve in -- this one minimises phase-2 work."""
    q = np.asarray(q, np.float64)
    lo = np.asarray(bbox_lo, np.float64)[None]        # [1][S][3]
    hi = np.asarray(bbox_hi, np.float64)[None]
    d = q[:, None, :] - np.clip(q[:, None, :], lo, hi)
    d2 = np.where(np.isfinite(d).all(axis=2), (d * d).sum(axis=2), np.inf)
    return np.argmin(d2, axis=1).astype(np.int32)


def two_phase_search(search_home, refine, keys, home_idx, reduce_min):
    """The two-phase sharded search, independent of what runs the per-shard kernel (HIP or, in the CPU tests, the oracle).
      search_home()          -> keys of this rank's home queries (torch int64, len(home_idx))
      refine(keys)           -> in place: min(keys, this shard's result) for the foreign queries it cannot rule out
      keys                   torch int64 [Q], overwritten;  home_idx: torch long indices of this rank's home queries
      reduce_min(keys)       element-wise MIN over ranks, in place"""
    keys.fill_(KEY_NONE)
    if home_idx.numel():
        keys[home_idx] = search_home()
    reduce_min(keys)
    refine(keys)
    reduce_min(keys)
    return keys


def bench_cloud_sharded(xyz, nrm, q_all, mr_all, rank, world, local_rank, dev, stream, steps, sync):
    """bench.py's cloud-sharded leg: spatially compact shards, two-phase search, winner payload, association."""
    import time
    import pcdhip
    order = compact_order(xyz)
    cuts = shard_cuts(len(order), world)
    rows = order[cuts[rank]:cuts[rank + 1]]
    shard = pcdhip.Cloud(xyz[rows], nrm[rows], device=local_rank, raw_lidar_frame=False, index_base=cuts[rank],
                         index_stride=1)
    info = shard.info()
    box = torch.tensor([info["bbox_lo"] + info["bbox_hi"]], dtype=torch.float64, device=dev)   # device tensors: RCCL
    boxes = [torch.zeros_like(box) for _ in range(world)]
    dist.all_gather(boxes, box)
    boxes = torch.cat(boxes).cpu().numpy()
    Q = q_all.shape[0]
    home = home_shards(q_all, boxes[:, :3], boxes[:, 3:])
    mine = np.nonzero(home == rank)[0]
    dq = torch.from_numpy(np.ascontiguousarray(q_all)).to(dev)
    dmr = torch.from_numpy(np.ascontiguousarray(mr_all)).to(dev)
    home_idx = torch.from_numpy(mine).to(dev)
    skip = torch.from_numpy((home == rank).astype(np.uint8)).to(dev)
    keys = torch.empty(Q, dtype=torch.int64, device=dev)
    kh = torch.empty(max(len(mine), 1), dtype=torch.int64, device=dev)
    payload = torch.empty(Q, 6, dtype=torch.int32, device=dev)
    f64 = lambda *s: torch.empty(*s, dtype=torch.float64, device=dev)
    aout = dict(lidar_xyz=f64(Q, 3), abcd=f64(Q, 4), type=torch.empty(Q, dtype=torch.uint8, device=dev), dist=f64(Q),
                angle=f64(Q))

    def search_home():
        qh = dq[home_idx].contiguous()
        shard.nn_device(qh, len(mine), kh, pcdhip.NN_AUTO, stream)
        return kh[:len(mine)]

    def step():
        two_phase_search(search_home, lambda k: shard.nn_refine_device(dq, Q, k, skip, stream), keys, home_idx,
                         lambda k: dist.all_reduce(k, op=dist.ReduceOp.MIN))
        shard.winner_payload_device(keys, Q, payload, stream)
        dist.all_reduce(payload)                                     # bit patterns, one owner each
        pcdhip.associate_from_payload_device(local_rank, dq, Q, dmr, Q, pcdhip.GATE_MAPPER_LOCAL, keys, payload, aout,
                                             stream)
    for _ in range(2):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    tc = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(tc, op=dist.ReduceOp.MAX)
    shard.close()
    t = float(tc.item()) / steps
    return dict(ms_per_step=t * 1e3, queries_per_sec=Q / t, scaling="strong", home_queries_this_rank=int(len(mine)),
                note="one cloud in N spatially compact shards (equal-count cuts of a coarse-cell order), all queries "
                     "known to every rank; two-phase search with two all-reduce(MIN) of the packed keys + one "
                     "all-reduce(SUM) of the winner payload")


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
