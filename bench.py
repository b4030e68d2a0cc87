#!/usr/bin/env python3
"""bench.py -- colmap-pcd registration hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input:
  1. exact NN association of Q = 1 M 3D feature points against the N = 10 M-point LiDAR cloud
     (replaces the serial KD-tree loops, lidar/kdtree.cc:10-21 via controllers/bundle_adjustment.cc:130-185),
  2. the fused plane-association epilogue (lidar/lidar_point.cc, optim/bundle_adjustment.cc:358-410),
  3. one bundle-adjustment iteration's worth of evaluation on a 1000-camera / 1 M-point / ~5 M-observation
     OPENCV scene: residuals + Jacobians + per-image / per-point J^T J, J^T r blocks + cost, then one
     residual-only (cost) pass (what ceres::Solve asks of the cost functions per LM iteration).
Inputs are resident in HBM before the timed region.  value = feature points taken through the whole step
per second, summed over all ranks; nn_queries_per_sec and ba_iter_ms are reported beside it.

N > 1 (one process per GPU, torch.distributed / RCCL): weak scaling -- every rank holds the whole cloud
(160 MB of 288 GB) and its own 1 M queries and its own 1 M-point track shard of the BA scene; the BA
camera blocks are combined with one RCCL all-reduce (sum, f64) per evaluation.  The cloud-sharded NN
variant (interleaved shards, RCCL all-reduce MIN on the packed per-query keys + SUM of the winner
payload) is timed after the main region and reported under "cloud_sharded".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import pcdhip  # noqa: E402
from pcdhip import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
REC_BYTES = 16                   # staged cloud record {x,y,z,index}
PER_QUERY_BYTES = 20             # 12 B query in + 8 B key out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cloud", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--cams", type=int, default=1000)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-queries", type=int, default=1_000_000)   # ~2.5 s of one core
    ap.add_argument("--cpu-sample-points", type=int, default=600_000)      # ~10 s of one core
    ap.add_argument("--no-cloud-sharded", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of the N>1 path)")
    return ap.parse_args()


def cpu_baseline(xyz, nrm, q, mr, scene, nq, npts):
    """oracle ("port") timed on this host, single thread like the reference's serial loops."""
    from oracle import pyoracle as po
    t0 = time.time()
    kd = po.KDTree(xyz)                      # FLANN-style single tree, leaf 15
    t_build = time.time() - t0
    qs = q[:nq]
    t0 = time.time()
    idx, sq, found = kd.query(qs)
    out6, ok = po.search_nearest_neibor(xyz, nrm, idx, found)
    po.associate(qs, out6, ok, mr[:nq], po.GATE_MAPPER_LOCAL)
    t_nn = time.time() - t0
    # BA: the tracks of the first npts points
    sel = scene["obs_point"] < npts
    lsel = scene["lidar_point"] < npts
    sub = dict(scene)
    sub["points"] = scene["points"][:npts]
    sub["obs_image"], sub["obs_point"], sub["obs_xy"] = scene["obs_image"][sel], scene["obs_point"][sel], scene["obs_xy"][sel]
    sub["lidar_point"], sub["lidar_abcd"], sub["lidar_weight"] = scene["lidar_point"][lsel], scene["lidar_abcd"][lsel], scene["lidar_weight"][lsel]
    ob = po.BA(**sub)
    t0 = time.time()
    ob.normal_equations()                    # residual + Jacobian (Jet autodiff) + blocks
    ob.evaluate_raw()                        # stands in for the residual-only pass (upper bound: also fills J)
    t_ba = time.time() - t0
    per_point = t_nn / nq + t_ba / npts
    return dict(value=1.0 / per_point, unit="queries/s", cores=1, kind="port",
                sample=f"kd-tree over the full {xyz.shape[0]}-pt cloud (build {t_build:.1f}s, not counted), {nq} queries "
                       f"NN+association {t_nn:.2f}s; BA Jet evaluation of {npts} tracks / {int(sel.sum())} obs {t_ba:.2f}s",
                nn_queries_per_sec=nq / t_nn, ba_iter_ms_full_scene=t_ba / npts * scene["points"].shape[0] * 1e3)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal on a 1-GPU box: PCD_BENCH_FORCE_DEVICE=0 puts every rank on that device (use --backend gloo)
    if os.environ.get("PCD_BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["PCD_BENCH_FORCE_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    # ------------------------------------------------------------ inputs (seeded) ---
    xyz, nrm = synth.cloud_planes(a.cloud)
    q = synth.queries(xyz, a.queries, seed=99 + rank)            # every rank its own batch (weak scaling)
    mr = synth.max_range_schedule(a.queries, seed=5 + rank)
    scene = synth.ba_scene(a.cams, a.points, seed=11 + rank)
    Q = a.queries
    cloud = pcdhip.Cloud(xyz, nrm, device=local_rank, raw_lidar_frame=False)
    ba = pcdhip.BA(**scene, device=local_rank)
    I, P, O, L = ba.I, ba.P, ba.O, ba.L

    dq = torch.from_numpy(q).to(dev)
    dmr = torch.from_numpy(mr).to(dev)
    keys = torch.empty(Q, dtype=torch.int64, device=dev)
    aout = dict(lidar_xyz=torch.empty(Q, 3, dtype=torch.float64, device=dev),
                abcd=torch.empty(Q, 4, dtype=torch.float64, device=dev),
                type=torch.empty(Q, dtype=torch.uint8, device=dev),
                dist=torch.empty(Q, dtype=torch.float64, device=dev),
                angle=torch.empty(Q, dtype=torch.float64, device=dev))
    img_blocks = torch.empty(I * 42, dtype=torch.float64, device=dev)   # [I][36] H then [I][6] g, one all-reduce
    H_img, g_img = img_blocks[: I * 36], img_blocks[I * 36:]
    H_pt = torch.empty(P, 9, dtype=torch.float64, device=dev)
    g_pt = torch.empty(P, 3, dtype=torch.float64, device=dev)
    cost = torch.empty(2, dtype=torch.float64, device=dev)
    full = dict(cost=cost[0:1], H_img=H_img, g_img=g_img, H_pt=H_pt, g_pt=g_pt)
    resid_only = dict(cost=cost[1:2])

    def step():
        cloud.nn_device(dq, Q, keys, pcdhip.NN_AUTO, stream)
        cloud.associate_device(dq, Q, dmr, Q, pcdhip.GATE_MAPPER_LOCAL, aout, keys, stream)
        ba.evaluate_device(full, stream)
        if world > 1:
            dist.all_reduce(img_blocks)           # RCCL sum over xGMI: camera J^T J / J^T r blocks
        ba.evaluate_device(resid_only, stream)
        if world > 1:
            dist.all_reduce(cost)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # algorithmic bytes of the brick kernel: one stats pass outside the timed region (deterministic input)
    pcdhip.set_nn_tuning(0, -1, 1)
    cloud.nn_device(dq, Q, keys, pcdhip.NN_AUTO, stream)
    torch.cuda.synchronize()
    st = cloud.last_stats()
    pcdhip.set_nn_tuning(0, -1, 0)

    for _ in range(a.warmup):
        step()
    sync()
    pcdhip.profile_enable(True)       # HIP events around every kernel, on the launch stream
    pcdhip.profile_reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    prof = pcdhip.profile_get()
    pcdhip.profile_enable(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / a.steps * 1e3

    # ------------------------------------------------- cloud-sharded NN (north_star's collective) ---
    cs = None
    if world > 1 and not a.no_cloud_sharded:
        shard = pcdhip.Cloud(xyz[rank::world], nrm[rank::world], device=local_rank, raw_lidar_frame=False,
                             index_base=rank, index_stride=world)
        q0 = torch.from_numpy(synth.queries(xyz, Q, seed=99)).to(dev)    # same queries on every rank
        payload = torch.empty(Q, 6, dtype=torch.int32, device=dev)

        def cs_step():
            shard.nn_device(q0, Q, keys, pcdhip.NN_AUTO, stream)
            dist.all_reduce(keys, op=dist.ReduceOp.MIN)                  # packed (distance, index) keys
            shard.winner_payload_device(keys, Q, payload, stream)
            dist.all_reduce(payload)                                     # bit patterns, one owner each
            pcdhip.associate_from_payload_device(local_rank, q0, Q, dmr, Q, pcdhip.GATE_MAPPER_LOCAL, keys, payload,
                                                 aout, stream)
        for _ in range(2):
            cs_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            cs_step()
        sync()
        tc = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        cs = dict(ms_per_step=float(tc.item()) / a.steps * 1e3, queries_per_sec=Q / (float(tc.item()) / a.steps),
                  scaling="strong", note="one 10M cloud in N interleaved shards, same 1M queries on every rank")
        shard.close()

    if rank == 0:
        per = {k: ms / n for k, (n, ms) in prof.items()}                       # average per launch
        nn_ms = sum(ms for k, (n, ms) in prof.items() if k.startswith("nn_") or k == "associate") / a.steps
        ba_ms = sum(ms for k, (n, ms) in prof.items() if k.startswith("ba_")) / a.steps   # per step (both passes)
        brick_ms = per.get("nn_brick", float("nan"))
        staged = st["staged_points"] - 0
        alg_bytes = staged * REC_BYTES + PER_QUERY_BYTES * Q
        achieved = alg_bytes / (brick_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            traffic = json.load(open(tp)).get("nn_brick_bytes_per_launch")
        out = {
            "metric": "NN queries/sec + BA-iter ms, 10M-pt cloud / 1M 3D feats",
            "value": world * Q / (dt / a.steps),
            "unit": "queries/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (NN distances) / f64 (association, BA)", "data": "synthetic",
            "config": {"workload": "M: 10M-pt 'planes' cloud / 1M queries per GPU + BA scene "
                                   f"{a.cams} cams / {P} pts / {O} obs / {L} lidar terms (OPENCV)",
                       "cloud_points": a.cloud, "queries_per_gpu": Q, "parallelism":
                       "single GPU" if world == 1 else f"cloud replicated, queries + tracks sharded x{world}, "
                       "RCCL all-reduce of camera blocks"},
            "nn_queries_per_sec": world * Q / (nn_ms * 1e-3),
            "ba_iter_ms": ba_ms,
            "kernel_ms": {k: round(v, 4) for k, v in sorted(per.items())},
            "roofline": {"bound": "hbm", "kernel": "k_nn_brick", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes": alg_bytes, "launch_ms": brick_ms,
                         "staged_points": staged, "brick_groups": st["brick_groups"],
                         "fallback_queries": st["fallback_queries"],
                         "compulsory_bytes": 12 * a.cloud + PER_QUERY_BYTES * Q},
        }
        if cs:
            out["cloud_sharded"] = cs
        if not a.no_cpu_baseline and world == 1:   # reported at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(xyz, nrm, q, mr, scene, min(a.cpu_sample_queries, Q),
                                               min(a.cpu_sample_points, P))
        print(json.dumps(out))
    cloud.close()
    ba.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
