"""world_size-2 gloo tests (CPU) of the N > 1 decomposition in pcdhip/dist.py.

The per-rank kernel is stood in for by the CPU oracle on that rank's shard; what is under test is the
decomposition itself: interleaved cloud shards + all-reduce(MIN) on packed keys + all-reduce(SUM) on the
winner payload bit patterns reproduce the single-cloud result bit-exactly (ties included), and track
shards + all-reduce(SUM) of the camera blocks reproduce the full normal equations."""
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


def _worker(rank, world, port, q):
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
        # ---------------- NN: query sharded (no collective) -----------------------------------
        lo, hi = pd.shard_range(len(qs), rank, world)
        pi, _, _ = po.nn_bruteforce(xyz, qs[lo:hi])
        assert np.array_equal(pi, ei[lo:hi])
        # ---------------- BA: track sharded ---------------------------------------------------
        scene = synth.ba_scene(8, 600, seed=9, const_pose_frac=0.25)
        sub, owned = pd.shard_tracks(scene, rank, world)
        cost, Himg, gimg, Hpt, gpt, _ = po.BA(**sub, loss_type=1, loss_scale=1.5).normal_equations()
        blocks = torch.from_numpy(np.concatenate([Himg.ravel(), gimg.ravel()]))
        c = torch.tensor([cost], dtype=torch.float64)
        pd.combine_blocks(blocks, c)
        fc, fH, fg, fHp, fgp, _ = po.BA(**scene, loss_type=1, loss_scale=1.5).normal_equations()
        I = scene["poses"].shape[0]
        np.testing.assert_allclose(blocks.numpy()[: I * 36].reshape(I, 6, 6), fH, rtol=1e-11, atol=1e-6)
        np.testing.assert_allclose(blocks.numpy()[I * 36:].reshape(I, 6), fg, rtol=1e-11, atol=1e-6)
        assert abs(c.item() - fc) <= 1e-12 * fc
        np.testing.assert_allclose(Hpt, fHp[owned], rtol=0, atol=0)     # point blocks are complete locally
        np.testing.assert_allclose(gpt, fgp[owned], rtol=0, atol=0)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_decomposition_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


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
