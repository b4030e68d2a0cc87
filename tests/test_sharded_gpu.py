"""One cloud over several shards of ONE process through the C ABI (pcd_cloud_create_sharded / pcd_nn_query_sharded /
pcd_associate_sharded): every result must equal the single-cloud one bit for bit -- index,
float distance bits, association fields -- including exact ties that straddle a shard cut (the winner is the lowest
ORIGINAL index, which may live in another shard than the query's home) and the raw LiDAR frame with NaN rows.
All shards sit on device 0 here (the box has one GPU); the reductions run through the library's own peer-copy path
and through a caller-supplied callback (what a C++ host would back with RCCL)."""
import ctypes as C

import numpy as np
import pytest
import torch

from pcdhip import synth

pytestmark = pytest.mark.gpu


def _exact(a, b, what):
    for x, y, n in zip(a, b, ("idx", "sqdist", "found")):
        xv = x.view(np.uint32) if x.dtype == np.float32 else x
        yv = y.view(np.uint32) if y.dtype == np.float32 else y
        bad = np.nonzero(xv != yv)[0]
        assert bad.size == 0, f"{what}: {n} differs at {bad[:5]}: {x[bad[:5]]} vs {y[bad[:5]]}"


def _cloud_with_straddling_ties(n=60000, seed=3):
    xyz, nrm = synth.cloud_planes(n, seed=seed, patches=10)
    rng = np.random.default_rng(seed)
    # exact duplicates far apart in the file order: the spatial sort puts them next to each other, the cuts may fall
    # between them, and the tie must still go to the lowest original index
    src = rng.integers(0, n // 2, 400)
    dst = rng.integers(n // 2, n, 400)
    xyz[dst] = xyz[src]
    return xyz, nrm, src


@pytest.mark.parametrize("nshards", [1, 2, 3, 4])
def test_sharded_equals_single_cloud(gpu, oracle, nshards):
    xyz, nrm, src = _cloud_with_straddling_ties()
    q = synth.queries(xyz, 8000, seed=5)
    q[:400] = xyz[src].astype(np.float64)                 # queries ON duplicated points: distance 0 twice
    q[400:420] = np.nan
    q[420:440] += 300.0                                   # far outside every shard
    single = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    ref = single.nn(q)
    _exact(ref, oracle.nn_bruteforce(xyz, q), "single cloud vs oracle")
    sh = gpu.ShardedCloud(xyz, nrm, [0] * nshards, raw_lidar_frame=False)
    assert len(sh) == len(single)
    _exact(sh.nn(q), ref, f"{nshards} shards, library reduction")
    # association: every field, both gates
    mr = synth.max_range_schedule(len(q), seed=2)
    for mode in (gpu.GATE_MAPPER_LOCAL, gpu.GATE_CONTROLLER):
        a = single.associate(q, mr, mode)
        b = sh.associate(q, mr, mode)
        for k in ("type", "nn_idx", "nn_sqdist", "lidar_xyz", "abcd", "dist", "angle", "dist2plane"):
            assert np.array_equal(np.ascontiguousarray(a[k]).view(np.uint8), np.ascontiguousarray(b[k]).view(np.uint8)), (mode, k)
    single.close(); sh.close()


def test_caller_supplied_reduction_and_raw_frame(gpu, oracle):
    """the exchange steps as callbacks (a C++ host: RCCL); raw LiDAR frame with NaN rows: indices are post-filter"""
    xyz_v, nrm_v = synth.cloud_uniform(30000, seed=21, box=np.array([12.0, 4.0, 12.0]))
    raw_xyz, raw_nrm = synth.visual_to_raw(xyz_v, nrm_v)
    rng = np.random.default_rng(4)
    raw_xyz[rng.integers(0, 30000, 200), rng.integers(0, 3, 200)] = np.nan
    exp_xyz, exp_nrm = oracle.direction_trans(raw_xyz, raw_nrm)
    q = synth.queries(exp_xyz, 5000, seed=2)
    calls = {"min": 0, "sum": 0}

    def view(ptr, count, dtype):
        # all shards are on device 0: wrap the raw device pointers as torch tensors through the CUDA array interface
        class _A:   # noqa: N801
            pass
        a = _A()
        a.__cuda_array_interface__ = {"shape": (count,), "typestr": dtype, "data": (ptr, False), "version": 2}
        return torch.as_tensor(a, device="cuda:0")

    def min_u64(user, bufs, devices, n, count):
        calls["min"] += 1
        ts = [view(bufs[s], count, "<i8") for s in range(n)]     # keys are < 2^63: signed order = unsigned order
        m = ts[0].clone()
        for t in ts[1:]:
            m = torch.minimum(m, t)
        for t in ts:
            t.copy_(m)
        torch.cuda.synchronize()
        return 0

    def sum_i32(user, bufs, devices, n, count):
        calls["sum"] += 1
        ts = [view(bufs[s], count, "<i4") for s in range(n)]
        m = ts[0].clone()
        for t in ts[1:]:
            m += t
        for t in ts:
            t.copy_(m)
        torch.cuda.synchronize()
        return 0

    red = gpu.ShardReduce(gpu.ShardReduce.MINFN(min_u64), gpu.ShardReduce.SUMFN(sum_i32), None)
    sh = gpu.ShardedCloud(raw_xyz, raw_nrm, [0, 0, 0], raw_lidar_frame=True)
    assert len(sh) == exp_xyz.shape[0] < 30000
    _exact(sh.nn(q, red), oracle.nn_bruteforce(exp_xyz, q), "3 shards, callback reduction, raw frame")
    assert calls["min"] == 2 and calls["sum"] == 0
    single = gpu.Cloud(raw_xyz, raw_nrm, raw_lidar_frame=True)
    a, b = single.associate(q, 1.2, gpu.GATE_MAPPER_LOCAL), sh.associate(q, 1.2, gpu.GATE_MAPPER_LOCAL, red)
    assert calls["sum"] == 1
    for k in a:
        if k in b:
            assert np.array_equal(np.asarray(a[k]).view(np.uint8), np.asarray(b[k]).view(np.uint8)), k
    single.close(); sh.close()


@pytest.mark.parametrize("callback", [False, True], ids=["library peer-copy reduction", "caller-supplied reduction"])
def test_shards_on_distinct_devices(gpu, oracle, callback):
    """The cross-device paths of csrc/shards.hip -- hipMemcpyPeer between DISTINCT devices in the library's reduction, the
    per-device null streams and hipSetDevice switching of the search, callbacks that receive buffers living on different
    devices -- need a host with more than one GPU.  The boxes this repository's GPU tests have run on so far have one:
    only the same-device configuration above has ever executed (include/pcdhip.h says so)."""
    ndev = gpu.device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} gfx950 device visible: the distinct-device configuration needs at least 2")
    devices = list(range(min(ndev, 4)))
    xyz, nrm, src = _cloud_with_straddling_ties()
    q = synth.queries(xyz, 8000, seed=5)
    q[:400] = xyz[src].astype(np.float64)
    q[400:420] = np.nan
    ref_cloud = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    ref = ref_cloud.nn(q)
    red = None
    if callback:
        def view(ptr, count, dtype, dev):
            class _A:   # noqa: N801
                pass
            a = _A()
            a.__cuda_array_interface__ = {"shape": (count,), "typestr": dtype, "data": (ptr, False), "version": 2}
            return torch.as_tensor(a, device=f"cuda:{dev}")

        def reduce_with(op):
            def fn(user, bufs, devs, n, count, dtype):
                ts = [view(bufs[s], count, dtype, devs[s]) for s in range(n)]
                for t, d in zip(ts, [devs[s] for s in range(n)]):
                    torch.cuda.synchronize(d)
                m = ts[0].clone()
                for t in ts[1:]:
                    m = op(m, t.to(m.device))
                for t in ts:
                    t.copy_(m.to(t.device))
                for s in range(n):
                    torch.cuda.synchronize(devs[s])
                return 0
            return fn
        mn, sm = reduce_with(torch.minimum), reduce_with(torch.add)
        red = gpu.ShardReduce(gpu.ShardReduce.MINFN(lambda u, b, d, n, c: mn(u, b, d, n, c, "<i8")),
                              gpu.ShardReduce.SUMFN(lambda u, b, d, n, c: sm(u, b, d, n, c, "<i4")), None)
    sh = gpu.ShardedCloud(xyz, nrm, devices, raw_lidar_frame=False)
    _exact(sh.nn(q, red) if red else sh.nn(q), ref, f"shards on devices {devices}")
    mr = synth.max_range_schedule(len(q), seed=2)
    a = ref_cloud.associate(q, mr, gpu.GATE_MAPPER_LOCAL)
    b = sh.associate(q, mr, gpu.GATE_MAPPER_LOCAL, red) if red else sh.associate(q, mr, gpu.GATE_MAPPER_LOCAL)
    for k in ("type", "nn_idx", "nn_sqdist", "lidar_xyz", "abcd", "dist", "angle", "dist2plane"):
        assert np.array_equal(np.ascontiguousarray(a[k]).view(np.uint8), np.ascontiguousarray(b[k]).view(np.uint8)), k
    ref_cloud.close(); sh.close()
