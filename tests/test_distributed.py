"""world_size-2 gloo tests of the N > 1 decomposition in pcdhip/dist.py.

What is under test is the decomposition itself: cloud shards (interleaved, and spatially compact with the
two-phase search) + all-reduce(MIN) on packed keys + all-reduce(SUM) on the winner payload bit patterns reproduce
the single-cloud result bit-exactly (ties across shards included), and track shards + all-reduce(SUM) of the
camera blocks reproduce the full normal equations.  The per-rank kernel is the CPU oracle in the CPU suite
(test_two_rank_decomposition_gloo) and the HIP library on the GPU box (tests/test_distributed_gpu.py runs the same
worker with both ranks on device 0)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_refine(po, shard_xyz, base, q, keys, skip):
    """what pcd_nn_refine_device does, with the oracle as the searcher: keys (numpy int64) in/out"""
    from pcdhip import dist as pd
    fin = np.isfinite(shard_xyz).all(axis=1)
    lo, hi = shard_xyz[fin].min(axis=0), shard_xyz[fin].max(axis=0)
    qf = q.astype(np.float32)
    p = np.minimum(np.maximum(qf, lo), hi)
    d = qf - p
    lb = d[:, 0] * d[:, 0]
    lb = lb + d[:, 1] * d[:, 1]
    lb = lb + d[:, 2] * d[:, 2]                              # float32, FLANN's order, as the kernel
    _, best, found = pd.unpack_keys(keys)
    best = np.where(found.astype(bool), best, np.float32(3.4028234663852886e38))
    act = np.isfinite(qf).all(axis=1) & (skip == 0) & (lb <= best)
    if act.any():
        li, lsq, lf = po.nn_bruteforce(shard_xyz, q[act])
        keys[act] = np.minimum(keys[act], pd.pack_keys(li, lsq, lf, base, 1))
    return int(act.sum())


def _two_phase(rank, world, use_hip, po, pd, synth):
    """spatially compact shards + two-phase search == single-cloud brute force, bit for bit, cross-shard ties included"""
    xyz, nrm = synth.cloud_planes(30000, seed=5, patches=10)
    order = pd.compact_order(xyz)
    xyz, nrm = xyz[order], nrm[order]                    # the global index space is the compact order
    n = xyz.shape[0]
    cuts = pd.shard_cuts(n, world)
    c = cuts[1]
    xyz[c - 6:c + 6] = xyz[c]                            # 12 copies of one point straddling the cut: ties across shards
    nrm[c - 6:c + 6] = nrm[c]
    qs = synth.queries(xyz, 3000, seed=6)                # incl. 5 % far outliers: phase 2 on both ranks
    qs[:50] = xyz[c].astype(np.float64) + np.random.default_rng(1).normal(0, 1e-3, (50, 3))
    qs[50] = xyz[c].astype(np.float64)
    qs[51] = [np.nan, 0, 0]
    Q = len(qs)
    lo_r, hi_r = cuts[rank], cuts[rank + 1]
    sx, sn = xyz[lo_r:hi_r], nrm[lo_r:hi_r]
    box = torch.tensor([list(sx.min(axis=0)) + list(sx.max(axis=0))], dtype=torch.float64)
    boxes = [torch.zeros_like(box) for _ in range(world)]
    dist.all_gather(boxes, box)
    boxes = torch.cat(boxes).numpy()
    home = pd.home_shards(qs, boxes[:, :3], boxes[:, 3:])
    mine = np.nonzero(home == rank)[0]
    skip = (home == rank).astype(np.uint8)
    cpu_min = lambda k: dist.all_reduce(k, op=dist.ReduceOp.MIN)
    if use_hip:
        import pcdhip
        shard = pcdhip.Cloud(sx, sn, device=0, raw_lidar_frame=False, index_base=lo_r, index_stride=1)
        dq = torch.from_numpy(qs).cuda()
        dskip = torch.from_numpy(skip).cuda()
        keys = torch.empty(Q, dtype=torch.int64, device="cuda")
        hidx = torch.from_numpy(mine).cuda()

        def search_home():
            kh = torch.empty(len(mine), dtype=torch.int64, device="cuda")
            shard.nn_device(dq[hidx].contiguous(), len(mine), kh, pcdhip.NN_GRID if rank == 0 else pcdhip.NN_AUTO)
            return kh

        def reduce_min(k):
            h = k.cpu()
            cpu_min(h)
            k.copy_(h)
        pd.two_phase_search(search_home, lambda k: shard.nn_refine_device(dq, Q, k, dskip), keys, hidx, reduce_min)
        torch.cuda.synchronize()
        got = keys.cpu().numpy()
        shard.close()
    else:
        keys = torch.empty(Q, dtype=torch.int64)
        hidx = torch.from_numpy(mine)

        def search_home():
            li, lsq, lf = po.nn_bruteforce(sx, qs[mine])
            return torch.from_numpy(pd.pack_keys(li, lsq, lf, lo_r, 1))
        refined = []

        def refine(k):
            kn = k.numpy()
            refined.append(_oracle_refine(po, sx, lo_r, qs, kn, skip))
        pd.two_phase_search(search_home, refine, keys, hidx, cpu_min)
        got = keys.numpy()
        assert 0 < refined[0] < Q - len(mine), "phase 2 must prune most foreign queries but not all"
    gi, gsq, gf = pd.unpack_keys(got)
    ei, esq, ef = po.nn_bruteforce(xyz, qs)
    bad = np.nonzero((gf != ef) | (gi != ei) | ((gsq.view(np.uint32) != esq.view(np.uint32)) & (ef != 0)))[0]
    assert bad.size == 0, (rank, bad[:8], gi[bad[:8]], ei[bad[:8]], gsq[bad[:8]], esq[bad[:8]], home[bad[:8]])
    assert gi[50] == c - 6 and gsq[50] == 0           # the tie goes to the lowest global index, which lives in shard 0
    assert not gf[51]


def _worker(rank, world, port, q, use_hip=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as po
        from pcdhip import synth
        from pcdhip import dist as pd
        # ---------------- NN: cloud sharded --------------------------------------------------
        xyz, nrm = synth.cloud_planes(20000, seed=3, patches=8)
        xyz[5000:5200] = xyz[100:300]                     # duplicates split across shards -> ties
        qs = synth.queries(xyz, 1500, seed=4)
        qs[:100] = xyz[5000:5100].astype(np.float64)
        sl, base, stride = pd.shard_rows(xyz.shape[0], rank, world)
        if use_hip:
            import pcdhip
            sh = pcdhip.Cloud(xyz[sl], nrm[sl], device=0, raw_lidar_frame=False, index_base=base, index_stride=stride)
            dk = torch.empty(len(qs), dtype=torch.int64, device="cuda")
            sh.nn_device(torch.from_numpy(qs).cuda(), len(qs), dk)
            torch.cuda.synchronize()
            keys = dk.cpu()
            sh.close()
        else:
            li, lsq, lf = po.nn_bruteforce(xyz[sl], qs)
            keys = torch.from_numpy(pd.pack_keys(li, lsq, lf, base, stride))
        pd.combine_keys(keys)
        gi, gsq, gf = pd.unpack_keys(keys.numpy())
        ei, esq, ef = po.nn_bruteforce(xyz, qs)
        assert np.array_equal(gi, ei) and np.array_equal(gsq.view(np.uint32), esq.view(np.uint32)) and np.array_equal(gf, ef)
        assert (gi[:100] == np.arange(100, 200)).all()     # lowest global index wins the tie
        # winner payload: owner fills bit patterns, others zero
        own = (gi % world) == rank
        payload = np.zeros((len(qs), 6), np.int32)
        rows = gi[own] // world
        payload[own, :3] = xyz[sl][rows].view(np.int32)
        payload[own, 3:] = nrm[sl][rows].view(np.int32)
        pt = torch.from_numpy(payload)
        pd.combine_payload(pt)
        got = pt.numpy()
        assert np.array_equal(got[:, :3].view(np.float32), xyz[ei]) and np.array_equal(got[:, 3:].view(np.float32), nrm[ei])
        # ---------------- NN: spatially compact shards, two-phase search ------------------------
        _two_phase(rank, world, use_hip, po, pd, synth)
        # ---------------- NN: query sharded (no collective) -----------------------------------
        lo, hi = pd.shard_range(len(qs), rank, world)
        pi, _, _ = po.nn_bruteforce(xyz, qs[lo:hi])
        assert np.array_equal(pi, ei[lo:hi])
        # ---------------- BA: track sharded ---------------------------------------------------
        scene = synth.ba_scene(8, 600, seed=9, const_pose_frac=0.25)
        sub, owned = pd.shard_tracks(scene, rank, world)
        if use_hip:
            import pcdhip
            hb = pcdhip.BA(**sub, loss_type=1, loss_scale=1.5, device=0)
            o = hb.evaluate(("cost", "H_img", "g_img", "H_pt", "g_pt"))
            cost, Himg, gimg, Hpt, gpt = o["cost"][0], o["H_img"], o["g_img"], o["H_pt"], o["g_pt"]
            hb.close()
        else:
            cost, Himg, gimg, Hpt, gpt, _ = po.BA(**sub, loss_type=1, loss_scale=1.5).normal_equations()
        blocks = torch.from_numpy(np.concatenate([Himg.ravel(), gimg.ravel()]))
        c = torch.tensor([cost], dtype=torch.float64)
        pd.combine_blocks(blocks, c)
        fc, fH, fg, fHp, fgp, _ = po.BA(**scene, loss_type=1, loss_scale=1.5).normal_equations()
        I = scene["poses"].shape[0]
        np.testing.assert_allclose(blocks.numpy()[: I * 36].reshape(I, 6, 6), fH, rtol=1e-11, atol=1e-6)
        np.testing.assert_allclose(blocks.numpy()[I * 36:].reshape(I, 6), fg, rtol=1e-11, atol=1e-6)
        assert abs(c.item() - fc) <= 1e-12 * fc
        tol = dict(rtol=1e-11, atol=1e-6) if use_hip else dict(rtol=0, atol=0)
        np.testing.assert_allclose(Hpt, fHp[owned], **tol)     # point blocks are complete locally
        np.testing.assert_allclose(gpt, fgp[owned], **tol)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def run_two_ranks(use_hip):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, use_hip)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_two_rank_decomposition_gloo():
    run_two_ranks(use_hip=False)


def test_key_order_is_signed_and_unsigned():
    """the packed key order must equal (distance, index) order under int64 MIN; NONE is the maximum"""
    sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
    from pcdhip import dist as pd
    d = np.array([0.0, 1e-30, 1.5, 1.5, 3.4e38], np.float32)
    i = np.array([7, 3, 9, 2, 0], np.uint32)
    k = pd.pack_keys(i, d, np.ones(5, np.uint8))
    assert (np.diff(np.sort(k)) >= 0).all() and list(np.argsort(k, kind="stable")) == [0, 1, 3, 2, 4]
    assert (k < pd.KEY_NONE).all() and (k >= 0).all()
    none = pd.pack_keys(i[:1], d[:1], np.zeros(1, np.uint8))
    assert none[0] == pd.KEY_NONE and min(int(none[0]), int(k[4])) == int(k[4])
