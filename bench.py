#!/usr/bin/env python3
"""bench.py -- colmap-pcd registration hot path on MI355X.

One "step" = one pass of the hot path over the metric workload M of BASELINE.json / SURVEY.md section 8d
(10 M-point LiDAR cloud, 1 M 3D feature points, one BA scene of 1000 cameras / 1 M tracks / ~5 M observations /
0.9 M LiDAR terms, OPENCV):
  1. exact NN association of the feature points against the cloud
     (replaces the serial KD-tree loops, lidar/kdtree.cc:10-21 via controllers/bundle_adjustment.cc:130-185),
  2. the fused plane-association epilogue (lidar/lidar_point.cc, optim/bundle_adjustment.cc:358-410),
  3. one bundle-adjustment iteration's worth of evaluation (section 8d: one evaluate WITH Jacobians + one
     residual-only evaluate): normal-equation mode -- cost, per-image 6x6 + 6 blocks, per-point 3x3 + 3 blocks
     and the 6x3 pose-point coupling block W of every observation (everything a Schur complement needs) --
     then the cost-only pass of the LM trial step.
Inputs are resident in HBM before the timed region.  value = feature points taken through the whole step per
second (whole job); nn_queries_per_sec and ba_iter_ms are reported beside it.  The Ceres route (raw mode:
residuals + ambient Jacobian blocks of every residual block, then a residual-only pass) is timed after the main
region and reported as ba_raw_iter_ms.

N > 1 (one process per GPU, torch.distributed / RCCL): STRONG scaling of the same workload -- the cloud is
replicated (160 MB of 288 GB), the 1 M queries are split into N spatial slabs and the BA scene into N track
shards (a rank owns points p = rank mod N with all their observations and LiDAR terms, so point blocks and W are
complete locally); the per-image blocks are partial sums: ONE RCCL all-reduce (sum, f64) of I*42 + 2 doubles per
step (blocks + the costs of both passes), issued asynchronously behind the BA passes and overlapped with the rank's
NN + association.  No collective on the NN path.  The cloud-sharded NN variant of
north_star (spatially compact shards, all-reduce MIN on the packed per-query keys) is timed after the main
region and reported under "cloud_sharded".
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks of the run (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cloud", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--cams", type=int, default=1000)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-queries", type=int, default=1_000_000)   # ~2.5 s of one core
    ap.add_argument("--cpu-sample-points", type=int, default=200_000)      # ~3 s of one core (Jets)
    ap.add_argument("--no-cloud-sharded", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the raw-mode / e2e / config-A legs")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of the N>1 path)")
    return ap.parse_args()


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, BEFORE this process has made any GPU
    call (nothing GPU-related is even imported yet), as plain child processes -- one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment -- and relay rank 0's JSON line.  (A process that has touched the GPU
    must never be replaced by exec on this pool; children are started with subprocess, this parent only waits.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Poll the ranks: a rank that dies (import error, OOM, a failed assert) leaves the others in init_process_group or
    # a collective until the RCCL timeout -- on the first non-zero exit, or at the deadline, the rest are terminated.
    # This parent has made no GPU call, so ending children is all the recovery there is to do.
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("PCD_BENCH_DEADLINE_S", "3000"))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or time.time() > deadline:
            failed = f"rank {bad[0]} exited with {rcs[bad[0]]}" if bad else "deadline reached"
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10
            while time.time() < t_kill and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    # rank 0's stdout may carry lines of the collective library (gloo announces its peers there): relay the JSON line
    lines = [ln for ln in (out0[0] if out0 else b"").decode().splitlines() if ln.startswith("{")]
    sys.stdout.write((lines[-1] + "\n") if lines else "")
    sys.stdout.flush()
    if failed or any(rcs):
        sys.stderr.write(f"bench.py: {failed or 'a rank failed'}; rank exit codes {rcs}\n")
        sys.exit(1)
    sys.exit(0)


ARGS = parse() if __name__ == "__main__" else None
if ARGS is not None:
    _ws = os.environ.get("WORLD_SIZE")
    if ARGS.gpus is None:
        ARGS.gpus = int(_ws) if _ws else 1
    if _ws is None and ARGS.gpus > 1:
        launch_ranks(ARGS)          # does not return
    if _ws is not None and int(_ws) != ARGS.gpus:
        sys.stderr.write(f"bench.py: --gpus {ARGS.gpus} but the launcher set WORLD_SIZE={_ws}\n")
        sys.exit(2)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import pcdhip  # noqa: E402
from pcdhip import dist as pdist  # noqa: E402
from pcdhip import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
POINT_BYTES = 12                 # xyz fp32 per staged cloud point
REC_BYTES = 16                   # what the layout actually streams: {x,y,z,index} records
PER_QUERY_BYTES = 20             # 12 B query in + 8 B key out


def source_hash():
    """hash of the NN kernel sources: profiles/traffic.json is only quoted when it was measured on these"""
    h = hashlib.sha256()
    for f in ("nn.hip", "brick_kernel.h", "brick_clip_kernel.h", "grid.h", "cloud.hip", "ba.hip", "ba_math.h", "ba_cam_jac.h"):
        with open(os.path.join(ROOT, "colmap-pcd_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(xyz, nrm, q, mr, scene, nq, npts):
    """The oracle ("port") timed on this host.  The reference's NN loops are serial and its Ceres evaluation uses
    all cores once the problem has >= 50 000 residuals (optim/bundle_adjustment.cc:515-530): both threadings are
    timed, for NN and for the Jet evaluation; `value` is the all-cores figure, the others are listed beside it."""
    from oracle import pyoracle as po
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                       # cgroup v2 CPU quota of the box ("max" = none)
        qv, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if qv != "max":
            quota = float(qv) / float(per)
            cores = max(1, min(cores, int(round(quota))))
    except (OSError, ValueError):
        pass
    cores = min(cores, 64)                     # threads actually used
    t0 = time.time()
    kd = po.KDTree(xyz)                      # FLANN-style single tree, leaf 15
    t_build = time.time() - t0
    qs = q[:nq]

    def nn_pass(threads):
        t0 = time.time()
        idx, sq, found = kd.query(qs) if threads == 1 else kd.query_mt(qs, threads)
        out6, ok = po.search_nearest_neibor(xyz, nrm, idx, found)
        po.associate(qs, out6, ok, mr[:nq], po.GATE_MAPPER_LOCAL)
        return time.time() - t0
    t_nn1 = nn_pass(1)
    t_nnc = nn_pass(cores)

    def sub_scene(n):
        sel = scene["obs_point"] < n
        lsel = scene["lidar_point"] < n
        sub = dict(scene)
        sub["points"] = scene["points"][:n]
        sub["obs_image"], sub["obs_point"], sub["obs_xy"] = scene["obs_image"][sel], scene["obs_point"][sel], scene["obs_xy"][sel]
        sub["lidar_point"], sub["lidar_abcd"], sub["lidar_weight"] = scene["lidar_point"][lsel], scene["lidar_abcd"][lsel], scene["lidar_weight"][lsel]
        return po.BA(**sub), int(sel.sum())
    P = scene["points"].shape[0]
    # one Ceres LM iteration's evaluation = every block's Evaluate with Jacobians (Jets) + once without
    ob1, nobs1 = sub_scene(npts)
    t0 = time.time(); ob1.evaluate_mt(1, True); ob1.evaluate_mt(1, False); t_ba1 = time.time() - t0
    nptc = min(P, npts * min(cores, 8))
    obc, nobsc = sub_scene(nptc)
    t0 = time.time(); obc.evaluate_mt(cores, True); obc.evaluate_mt(cores, False); t_bac = time.time() - t0
    per_point = lambda tnn, tba, n: tnn / nq + tba / n
    v_all = 1.0 / per_point(t_nnc, t_bac, nptc)
    # time the literal FLANN / PCL / Ceres calls if the box has them.  Probe for the headers (nothing is
    # installed or fetched): absent on every box seen so far, so the timed baseline is the oracle ("port")
    def have(*names):
        roots = ("/usr/include", "/usr/local/include", "/opt/rocm/include", "/usr/include/x86_64-linux-gnu")
        return any(os.path.exists(os.path.join(r, n)) for r in roots for n in names)
    ref_libs = dict(flann=have("flann/flann.hpp"), pcl=have("pcl/kdtree/kdtree_flann.h", "pcl-1.12/pcl/kdtree/kdtree_flann.h",
                                                            "pcl-1.10/pcl/kdtree/kdtree_flann.h"),
                    ceres=have("ceres/ceres.h"), eigen=have("eigen3/Eigen/Core", "Eigen/Core"))
    return dict(value=v_all, unit="queries/s", cores=cores, kind="port", reference_libs_on_box=ref_libs,
                host=f"{len(os.sched_getaffinity(0))} logical CPUs visible, cgroup quota {quota if quota else 'none'}, "
                     f"{cores} threads used",
                sample=f"kd-tree over the full {xyz.shape[0]}-pt cloud (build {t_build:.1f}s, not counted); {nq} queries "
                       f"NN+association: {t_nn1:.2f}s on 1 thread, {t_nnc:.2f}s on {cores}; BA Jet evaluation + "
                       f"residual-only pass: {npts} tracks / {nobs1} obs {t_ba1:.2f}s on 1 thread, "
                       f"{nptc} tracks / {nobsc} obs {t_bac:.2f}s on {cores}",
                nn_queries_per_sec=nq / t_nnc, ba_iter_ms_full_scene=t_bac / nptc * P * 1e3,
                single_thread=dict(cores=1, value=1.0 / per_point(t_nn1, t_ba1, npts), nn_queries_per_sec=nq / t_nn1,
                                   ba_iter_ms_full_scene=t_ba1 / npts * P * 1e3),
                reference_threading=dict(note="NN serial (as the reference's loops), BA on all cores (as its Ceres options)",
                                         value=1.0 / per_point(t_nn1, t_bac, nptc)))


def timed(fn, steps, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def main():
    a = ARGS
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
    # what the collective layer itself saw: ranks, backend, and the physical device behind every rank
    pr = torch.cuda.get_device_properties(local_rank)
    my_dev = dict(rank=rank, index=local_rank, name=pr.name,
                  id=str(getattr(pr, "uuid", None) or getattr(pr, "pci_bus_id", None) or local_rank))
    dist_info = dict(world_size=1, backend=None, devices=[my_dev], distinct_devices=1)
    if world > 1:
        devs = [None] * world
        dist.all_gather_object(devs, my_dev)
        dist_info = dict(world_size=dist.get_world_size(), backend=dist.get_backend(), devices=devs,
                         distinct_devices=len({d["id"] for d in devs}))
        assert dist_info["world_size"] == a.gpus, (dist_info, a.gpus)
        if a.backend == "nccl" and os.environ.get("PCD_BENCH_FORCE_DEVICE") is None:
            assert dist_info["distinct_devices"] == world, f"{world} ranks on {dist_info['distinct_devices']} devices: {devs}"

    # ------------------------------------------------------------ inputs (seeded) ---
    # ONE workload for every N: the same cloud, queries and scene on every rank; rank r works on its share.
    xyz, nrm = synth.cloud_planes(a.cloud)
    q_all = synth.queries(xyz, a.queries, seed=99)
    mr_all = synth.max_range_schedule(a.queries, seed=5)
    scene = synth.ba_scene(a.cams, a.points, seed=11, order="image")   # observations in AddImageToProblem order
    Qtot = a.queries
    if world > 1:
        # the split is SPATIAL: a rank gets a slab of the queries (contiguous range of a coarse-cell order), so the
        # bricks it works on are as densely populated as on one GPU -- a random 1/N of the queries would leave every
        # brick with 1/N of its queries and the per-rank staging work nearly unchanged (tools/scaling_probe.py)
        perm = pdist.compact_order(q_all)
        q_all, mr_all = q_all[perm], mr_all[perm]
    qlo, qhi = pdist.shard_range(Qtot, rank, world)
    q, mr = q_all[qlo:qhi], mr_all[qlo:qhi]
    Q = qhi - qlo
    my_scene = scene if world == 1 else pdist.shard_tracks(scene, rank, world)[0]
    cloud = pcdhip.Cloud(xyz, nrm, device=local_rank, raw_lidar_frame=False)
    torch.cuda.synchronize()
    t_create = time.perf_counter()
    ba = pcdhip.BA(**my_scene, device=local_rank)
    torch.cuda.synchronize()
    ba_create_ms = (time.perf_counter() - t_create) * 1e3   # upload + index building (host counting sorts, device gathers)
    I, P, O, L = ba.I, ba.P, ba.O, ba.L
    Ptot, Otot, Ltot = scene["points"].shape[0], len(scene["obs_image"]), len(scene["lidar_point"])

    dq = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
    dmr = torch.from_numpy(np.ascontiguousarray(mr)).to(dev)
    keys = torch.empty(max(Q, 1), dtype=torch.int64, device=dev)
    f64 = lambda *s: torch.empty(*s, dtype=torch.float64, device=dev)
    aout = dict(lidar_xyz=f64(max(Q, 1), 3), abcd=f64(max(Q, 1), 4),
                type=torch.empty(max(Q, 1), dtype=torch.uint8, device=dev), dist=f64(max(Q, 1)), angle=f64(max(Q, 1)))
    # [I][36] H, [I][6] g, cost of the Jacobian pass, cost of the residual-only pass: ONE all-reduce per step
    img_blocks = f64(I * 42 + 2)
    H_img, g_img = img_blocks[: I * 36], img_blocks[I * 36: I * 42]
    cost_j, cost_r = img_blocks[I * 42: I * 42 + 1], img_blocks[I * 42 + 1:]
    H_pt, g_pt, W = f64(P, 9), f64(P, 3), f64(max(O, 1), 18)
    full = dict(cost=cost_j, H_img=H_img, g_img=g_img, H_pt=H_pt, g_pt=g_pt, W=W)
    resid_only = dict(cost=cost_r)

    def nn_step():
        cloud.nn_device(dq, Q, keys, pcdhip.NN_AUTO, stream)
        cloud.associate_device(dq, Q, dmr, Q, pcdhip.GATE_MAPPER_LOCAL, aout, keys, stream)

    def ba_step():
        ba.evaluate_device(full, stream)
        ba.evaluate_device(resid_only, stream)
        if world > 1:
            dist.all_reduce(img_blocks)       # RCCL sum over xGMI: camera J^T J / J^T r blocks + the two costs

    def step():
        # The two halves of a step do not depend on each other.  N > 1: the BA passes go first, their one all-reduce
        # (I * 42 + 2 doubles) is issued asynchronously and travels over xGMI while this rank's NN + association run;
        # the step ends when both are done.  Every step still produces its reduced blocks and costs.
        ba.evaluate_device(full, stream)
        ba.evaluate_device(resid_only, stream)
        work = dist.all_reduce(img_blocks, async_op=True) if world > 1 else None
        nn_step()
        if work is not None:
            work.wait()                       # the compute stream waits for the collective; the host does not block

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
    # Timed region: HIP events (on the launch stream) around the roofline kernel only -- every timed scope costs two
    # event records, a few microseconds of stream time each; the per-kernel breakdown of all scopes is taken in a
    # second, untimed pass of the same K steps below.
    pcdhip.profile_only("nn_brick")
    pcdhip.profile_enable(True)
    pcdhip.profile_reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    prof_timed = pcdhip.profile_get()
    pcdhip.profile_only(None)
    pcdhip.profile_reset()
    for _ in range(a.steps):
        step()
    sync()
    prof = pcdhip.profile_get()
    pcdhip.profile_enable(False)
    if "nn_brick" in prof_timed:
        prof["nn_brick"] = prof_timed["nn_brick"]   # the roofline's launch duration comes from the timed region
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / a.steps * 1e3

    extras = {}
    if not a.no_extras:
        # ---- Ceres route: raw residual + Jacobian blocks, then a residual-only pass (residual vector) ----
        raw = dict(residuals=f64(2 * O + L), jac_q=f64(max(O, 1), 8), jac_t=f64(max(O, 1), 6), jac_X=f64(max(O, 1), 6),
                   jac_lidar=f64(max(L, 1), 3))
        res_only = dict(residuals=raw["residuals"])

        def raw_step():
            ba.evaluate_device(raw, stream)
            ba.evaluate_device(res_only, stream)
        for _ in range(2):
            raw_step()
        t_raw = timed(raw_step, a.steps, sync)
        t_nn = timed(nn_step, a.steps, sync)
        t_ba = timed(ba_step, a.steps, sync)
        extras.update(ba_raw_iter_ms=t_raw * 1e3, nn_wall_ms=t_nn * 1e3, ba_wall_ms=t_ba * 1e3, ba_create_ms=ba_create_ms)
        if world == 1:
            # N > 1 runs cut the queries in a spatially compact order (done on the host, outside the timed region):
            # the like-for-like single-GPU figure for a scaling ratio is this one, on the SAME pre-sorted queries
            perm = pdist.compact_order(q_all)
            dq_c = torch.from_numpy(np.ascontiguousarray(q_all[perm])).to(dev)
            dmr_c = torch.from_numpy(np.ascontiguousarray(mr_all[perm])).to(dev)

            def nn_step_c():
                cloud.nn_device(dq_c, Q, keys, pcdhip.NN_AUTO, stream)
                cloud.associate_device(dq_c, Q, dmr_c, Q, pcdhip.GATE_MAPPER_LOCAL, aout, keys, stream)
            for _ in range(2):
                nn_step_c()
            extras["nn_wall_ms_compact_order"] = timed(nn_step_c, a.steps, sync) * 1e3
            extras["scaling_note"] = ("N > 1 legs pre-sort the queries spatially on the host (untimed, once per query "
                                      "set); compare them with nn_wall_ms_compact_order + ba_wall_ms, not with "
                                      "ms_per_step, for a like-for-like strong-scaling ratio")
            del dq_c, dmr_c
        del raw, res_only
        if world == 1:
            # the same scene with point ids that follow the images (synth.ba_scene coherent=True): what an incremental
            # reconstruction's numbering looks like; the headline scene gives every point a random anchor image
            sc_c = synth.ba_scene(a.cams, a.points, seed=11, order="image", coherent=True)
            ba_c = pcdhip.BA(**sc_c, device=local_rank)
            Oc = ba_c.O
            Wc = f64(max(Oc, 1), 18)
            full_c = dict(cost=cost_j, H_img=H_img, g_img=g_img, H_pt=H_pt, g_pt=g_pt, W=Wc)

            def ba_c_step():
                ba_c.evaluate_device(full_c, stream)
                ba_c.evaluate_device(resid_only, stream)
            for _ in range(2):
                ba_c_step()
            pcdhip.profile_enable(True); pcdhip.profile_reset()
            t_c = timed(ba_c_step, a.steps, sync)
            pc = pcdhip.profile_get(); pcdhip.profile_enable(False)
            extras["ba_coherent_scene"] = dict(
                note="same cameras / tracks, point ids ascending with the anchor image (incremental-reconstruction numbering)",
                ba_iter_ms=t_c * 1e3, kernel_ms={k: round(ms / n, 4) for k, (n, ms) in sorted(pc.items()) if k.startswith("ba_")})
            ba_c.close()
            del Wc
        # ---- what the call sites need: the search bounded by each query's gate (PCD_GATE_BOUNDED_SEARCH) --------
        # same recorded associations, same field values (tests/test_assoc_gpu.py::test_gate_bounded_search); the
        # headline above keeps the unbounded exact search for every query
        def bounded_step():
            cloud.associate_device(dq, Q, dmr, Q, pcdhip.GATE_MAPPER_LOCAL | pcdhip.GATE_BOUNDED_SEARCH, aout, None, stream)
        for _ in range(2):
            bounded_step()
        t_bnd = timed(bounded_step, a.steps, sync)
        extras.update(nn_assoc_bounded_ms=t_bnd * 1e3, nn_assoc_bounded_queries_per_sec=Q / t_bnd * world)
        # ---- the drop-in's host path: pinned staging -> H2D -> search + epilogue -> compaction -> D2H of the hits ----
        # (what shim/lidar_hip.h MatchClosestLidarPointsFlat costs after the call site has gathered XYZ into the
        #  staging buffers and before it inserts into its hash maps; both exist in the reference's loops as well)
        sq, smr = cloud.staging(Q)
        sq[:] = q
        smr[:] = mr
        for _ in range(2):
            hits = cloud.associate_staged(Q, Q, pcdhip.GATE_MAPPER_LOCAL)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            hits = cloud.associate_staged(Q, Q, pcdhip.GATE_MAPPER_LOCAL)
        t_e2e = (time.perf_counter() - t0) / a.steps
        extras.update(e2e_ms=t_e2e * 1e3, e2e_hits=int(len(hits)),
                      e2e_def="pcd_associate_staged: pinned inputs -> pinned 80-B records of the accepted associations, "
                              "PCIe both ways included", e2e_over_device=t_e2e / t_nn)
        # ---- config A (Smith Hall 25-like): 2 M-point cloud, 20 k queries per call -> latency per call ----
        if rank == 0:
            xa, na = synth.cloud_planes(2_000_000)
            ca = pcdhip.Cloud(xa, na, device=local_rank, raw_lidar_frame=False)
            qa = torch.from_numpy(synth.queries(xa, 20_000, seed=3)).to(dev)
            mra = torch.from_numpy(synth.max_range_schedule(20_000, seed=3)).to(dev)
            ka = torch.empty(20_000, dtype=torch.int64, device=dev)
            oa = {k: v[:20_000] for k, v in aout.items()} if Q >= 20_000 else None
            if oa is not None:
                def a_step():   # as the shim calls it: search bounded by the gate, then the epilogue
                    ca.associate_device(qa, 20_000, mra, 20_000, pcdhip.GATE_MAPPER_LOCAL | pcdhip.GATE_BOUNDED_SEARCH,
                                        oa, None, stream)
                for _ in range(3):
                    a_step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    a_step()
                torch.cuda.synchronize()
                t_a = (time.perf_counter() - t0) / 50
                sqa, smra = ca.staging(20_000)
                sqa[:] = qa.cpu().numpy(); smra[:] = mra.cpu().numpy()
                ca.associate_staged(20_000, 20_000, pcdhip.GATE_MAPPER_LOCAL)
                t0 = time.perf_counter()
                for _ in range(50):
                    ca.associate_staged(20_000, 20_000, pcdhip.GATE_MAPPER_LOCAL)
                t_a2 = (time.perf_counter() - t0) / 50
                extras["config_A"] = dict(workload="2M-pt cloud / 20k queries per call (incremental mapper's local BA, "
                                                   "sfm/incremental_mapper.cc:1155-1165)",
                                          device_us_per_call=t_a * 1e6, e2e_us_per_call=t_a2 * 1e6,
                                          queries_per_sec=20_000 / t_a)
            ca.close()

    # ---- config 5 (SIFT, the one MFMA use): one block of the exhaustive matcher = 50 images x 8192 descriptors,
    # 1225 pairs through the batched entry (SiftFeatureMatcher::Match(image_pairs), feature/matching.cc:798) ----
    if not a.no_extras and rank == 0:
        n_img, n_desc = 50, 8192
        rng = np.random.default_rng(0)
        f = rng.random((n_desc, 128), dtype=np.float32) ** 2
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        base = np.clip(np.round(512 * f), 0, 255).astype(np.int32)
        arena = np.concatenate([np.clip(base[rng.permutation(n_desc)] + rng.integers(-5, 6, (n_desc, 128)), 0, 255).astype(np.uint8)
                                for _ in range(n_img)], axis=0)
        first = np.arange(n_img + 1, dtype=np.uint64) * np.uint64(n_desc)
        pairs = np.array([(i, j) for i in range(n_img) for j in range(i + 1, n_img)], np.uint32)
        off = np.arange(len(pairs), dtype=np.uint64) * np.uint64(n_desc)
        d_arena = torch.from_numpy(arena).to(dev)
        d_m = torch.empty(len(pairs) * n_desc, 2, dtype=torch.int32, device=dev)
        d_c = torch.empty(len(pairs), dtype=torch.int32, device=dev)
        for _ in range(2):
            pcdhip.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c, device=local_rank, stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            pcdhip.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c, device=local_rank, stream=stream)
        torch.cuda.synchronize()
        t_sift = (time.perf_counter() - t0) / 3
        useful = 2.0 * 128 * n_desc * n_desc * len(pairs) / t_sift / 1e12
        extras["sift_block"] = dict(workload=f"{n_img} images x {n_desc} descriptors, {len(pairs)} pairs (one exhaustive-matcher block)",
                                    ms=t_sift * 1e3, us_per_pair=t_sift / len(pairs) * 1e6, pairs_per_sec=len(pairs) / t_sift,
                                    useful_TOPs=useful, useful_frac_of_dense_i8_peak=useful / 5000.0,
                                    executed_TOPs=2 * useful, executed_frac_of_dense_i8_peak=2 * useful / 5000.0,
                                    executed_def="every score tile is multiplied twice, once per matching direction (csrc/sift.hip)",
                                    matches=int(d_c.sum().item()), config5_450_images_s=101025 * t_sift / len(pairs))
        del d_arena, d_m, d_c
        # config 5 itself: 450 images, all 101 025 pairs, block by block as ExhaustiveFeatureMatcher::Run
        # (feature/matching.cc:902-960) -- measured, not extrapolated (tests/test_sift_gpu.py checks the same sweep)
        n_img = 450
        pool = torch.from_numpy(np.concatenate([base, base[::-1]]).astype(np.int16)).to(dev)
        gen = torch.Generator(device=dev).manual_seed(7)
        d_arena = torch.empty((n_img * n_desc, 128), dtype=torch.uint8, device=dev)
        for i in range(n_img):
            pick = torch.randperm(2 * n_desc, device=dev, generator=gen)[:n_desc]
            noise = torch.randint(-5, 6, (n_desc, 128), device=dev, generator=gen, dtype=torch.int16)
            d_arena[i * n_desc:(i + 1) * n_desc] = (pool[pick] + noise).clamp_(0, 255).to(torch.uint8)
        first = np.arange(n_img + 1, dtype=np.uint64) * np.uint64(n_desc)
        d_m = torch.empty(2500 * n_desc, 2, dtype=torch.int32, device=dev)
        d_c = torch.empty(2500, dtype=torch.int32, device=dev)
        blocks = [b for b in pcdhip.exhaustive_blocks(n_img, 50) if len(b)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_pairs = 0
        for pr in blocks:
            pcdhip.sift_match_batch_device(d_arena, first, pr, d_m, np.arange(len(pr), dtype=np.uint64) * np.uint64(n_desc), d_c,
                                           device=local_rank, stream=stream)
            n_pairs += len(pr)
        torch.cuda.synchronize()
        t_c5 = time.perf_counter() - t0
        extras["sift_config5"] = dict(workload=f"{n_img} images x {n_desc} descriptors, exhaustive: {n_pairs} pairs in {len(blocks)} blocks of <= 50 x 50 images",
                                      seconds=t_c5, pairs_per_sec=n_pairs / t_c5,
                                      useful_TOPs=2.0 * 128 * n_desc * n_desc * n_pairs / t_c5 / 1e12)
        del d_arena, d_m, d_c, pool

    # ---- the Ceres route end to end (shim/ceres_adapter.h): PrepareForEvaluation with Jacobians + one sweep of every
    # block's Evaluate, PCIe included -- what a colmap user's ceres::Solve sees per evaluation (C++ harness, run as a
    # child process; config B = Smith Hall 450-like, M = the metric workload) ----
    if not a.no_extras and rank == 0:
        import subprocess
        exe = os.path.join(ROOT, "colmap-pcd_amd", "shim", "ceres_route_bench")
        if os.path.exists(exe):
            cr = {}
            for name, (ni, npt) in (("config_B", (450, 400_000)), ("workload_M", (a.cams, a.points))):
                try:
                    r = subprocess.run([exe, str(ni), str(npt), "0"], capture_output=True, timeout=600)
                    cr[name] = json.loads(r.stdout.decode().strip().splitlines()[-1])
                except Exception as e:   # noqa: BLE001
                    cr[name] = {"error": repr(e)}
            m = cr.get("workload_M", {})
            extras["ceres_route"] = cr
            extras["ceres_route_e2e_ms"] = m.get("ceres_route_e2e_ms")
            extras["ceres_route_def"] = ("HipEvaluation::PrepareForEvaluation(jacobians) [gather + H2D + raw kernels + D2H into the "
                                         "handle's pinned buffers] + one sweep of CostFunction::Evaluate over every residual "
                                         "block on the host threads listed; bytes_d2h_jacobians cross PCIe per evaluation")

    # ------------------------------------------------- cloud-sharded NN (north_star's collective) ---
    cs = None
    if world > 1 and not a.no_cloud_sharded and hasattr(pdist, "bench_cloud_sharded"):
        cs = pdist.bench_cloud_sharded(xyz, nrm, q_all, mr_all, rank, world, local_rank, dev, stream, a.steps, sync)

    if rank == 0:
        per = {k: ms / n for k, (n, ms) in prof.items()}                       # average per launch
        nn_ms = sum(ms for k, (n, ms) in prof.items() if k.startswith("nn_") or k == "associate") / a.steps
        ba_ms = sum(ms for k, (n, ms) in prof.items() if k.startswith("ba_")) / a.steps   # per step (both passes)
        brick_ms = per.get("nn_brick", float("nan"))
        staged = st["staged_points"]
        alg_bytes = staged * POINT_BYTES + PER_QUERY_BYTES * Q
        rec_achieved = (staged * REC_BYTES + PER_QUERY_BYTES * Q) / (brick_ms * 1e-3) / 1e9
        # PMC numbers (HBM-side bytes, VALU wave-instructions) come from tools/prof_bench.sh passes of this same command,
        # kept in profiles/traffic.json with the hash of the kernel sources they were measured on; they are quoted only
        # when that hash matches the sources in this tree (rocprofv3 cannot run inside the timed run).
        tj, tnote = {}, "profiles/traffic.json absent"
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            tj = json.load(open(tp))
            if tj.get("source_hash") == source_hash() and tj.get("workload") == [a.cloud, Qtot] and world == 1:
                tnote = ("PMC per launch (2*FETCH_SIZE + WRITE_SIZE; SQ_INSTS_VALU) from tools/prof_bench.sh on these "
                         "kernel sources, profiles/traffic.json")
            else:
                tj, tnote = {}, "profiles/traffic.json was measured on other kernel sources / another workload: not quoted"

        def pmc(kernel, field):
            for name, rec in tj.get("kernels", {}).items():
                if kernel in name and rec.get(field) is not None:
                    return rec[field]
            return None

        def entry(kernel, scope, bound, alg_bytes_, what, **extra):
            """one kernel against the HBM roofline on its SURVEY 8d bytes; `traffic` = PMC bytes per launch"""
            ms = per.get(scope)
            if ms is None:
                return None
            ach = alg_bytes_ / (ms * 1e-3) / 1e9
            tr = pmc(kernel, "bytes_per_launch")
            e = dict(kernel=kernel, bound=bound, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                     traffic=tr, launch_ms=ms, algorithmic_bytes=alg_bytes_, algorithmic_bytes_def=what)
            if tr is not None:
                e["hbm_frac"] = tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                e["traffic_over_algorithmic"] = tr / alg_bytes_
            e.update(extra)
            return e

        pairs = st["pair_evals"]
        valu = pmc("k_nn_brick", "valu_wave_instructions")
        brick = entry("k_nn_brick_clip", "nn_brick", "valu-issue", alg_bytes,
                      "12 B x staged points (counted by the kernel's statistics pass) + 20 B x queries",
                      staged_points=staged, brick_groups=st["brick_groups"], pair_evals=pairs,
                      pair_evals_per_query=pairs / max(Q, 1), pairs_per_sec=pairs / (brick_ms * 1e-3),
                      compare_valu_wave_instructions=pairs / 64 * 9, valu_wave_instructions=valu,
                      compare_share_of_valu=(pairs / 64 * 9 / valu) if valu else None,
                      # 8 full-rate + 1 half-rate VALU per pair on 1024 SIMDs at the measured 1.03 / 1.9 ns per
                      # wave-instruction (profiles/r02_ubench_valu_rate.txt, profiles/r04_ubench_cmp_block.txt)
                      compare_floor_ms=pairs / 64 * (8 * 1.03 + 1.9) * 1e-6 / 1024,
                      frac_16B_records=rec_achieved / HBM_PEAK_GBS, compulsory_bytes=12 * a.cloud + PER_QUERY_BYTES * Q,
                      note="VALU-issue-bound, not HBM-bound (DESIGN.md section 5): `frac` is SURVEY 8d's algorithmic "
                           "figure (staged points are re-read from L2 / MALL: it can exceed the fabric share), "
                           "`hbm_frac` the PMC share of HBM peak; the kernel's own measure is pairs_per_sec against "
                           "compare_floor_ms")
        fb_bytes = st["fallback_points"] * POINT_BYTES + PER_QUERY_BYTES * st["fallback_queries"]
        kernels = {
            "k_nn_brick_clip": brick,
            "k_nn_fallback": entry("k_nn_fallback", "nn_fallback", "latency (dependent walk steps)", fb_bytes,
                                   "12 B x leaf points scanned + 20 B x fallback queries (scope includes k_fb_compact)",
                                   fallback_queries=st["fallback_queries"], fallback_points=st["fallback_points"]),
            "k_ba_images": entry("k_ba_images", "ba_images_w", "hbm", 200 * O,
                                 "SURVEY 8d Jacobian pass: 200 B per observation (16 obs + 12 indices + point + 144 B of W)"),
            "k_ba_points": entry("k_ba_points", "ba_points", "latency (gathers at 4 waves/SIMD)", 84 * O + 96 * P + 100 * L,
                                 "84 B per observation (16 obs + 12 indices + 56 pose) + 96 B of point blocks per "
                                 "track + SURVEY 8d's 100 B per lidar term"),
            "k_ba_cost": entry("k_ba_cost", "ba_points_cost", "hbm", 44 * O + 76 * L,
                               "SURVEY 8d residual-only pass: 44 B per observation + 76 B per lidar term"),
        }
        sb = extras.get("sift_block")
        if sb:
            kernels["k_sift_scores_batch"] = dict(kernel="k_sift_scores_batch (+ finalize, compaction)", bound="mfma",
                                                  achieved=sb["useful_TOPs"], peak=5000.0, unit="TOP/s",
                                                  frac=sb["useful_frac_of_dense_i8_peak"], traffic=None,
                                                  launch_ms=sb["ms"], algorithmic_flops_def="2 x 128 x n1 x n2 per pair",
                                                  executed_frac=sb["executed_frac_of_dense_i8_peak"],
                                                  note="useful = one product per pair; the kernel executes two (one walk per "
                                                       "direction) at ~2.1 GHz under the 1.26 kW socket power it draws "
                                                       "(DESIGN.md section 4.4)")
        kernels = {k: v for k, v in kernels.items() if v}
        ba_alg = (200 + 44) * O + (100 + 76) * L
        traffic, hbm_frac = brick.get("traffic"), brick.get("hbm_frac")
        out = {
            "metric": "NN queries/sec + BA-iter ms, 10M-pt cloud / 1M 3D feats",
            "value": Qtot / (dt / a.steps),
            "unit": "queries/s",
            "n_gpus": world, "dist": dist_info, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (NN distances) / f64 (association, BA)", "data": "synthetic",
            "config": {"workload": f"M: {a.cloud}-pt 'planes' cloud / {Qtot} queries + BA scene "
                                   f"{a.cams} cams / {Ptot} pts / {Otot} obs / {Ltot} lidar terms (OPENCV), "
                                   "the same total at every N",
                       "cloud_points": a.cloud, "queries": Qtot, "queries_this_rank": Q, "parallelism":
                       "single GPU" if world == 1 else f"cloud replicated, queries + tracks split x{world}, "
                       "one async RCCL all-reduce of camera blocks + costs per step"},
            "nn_queries_per_sec": Q / (nn_ms * 1e-3) * world,
            "ba_iter_ms": ba_ms,
            "ba_iter_def": "normal-equation pass (cost, H_img, g_img, H_pt, g_pt, W) + cost-only pass, kernel time of rank 0",
            "kernel_ms": {k: round(v, 4) for k, v in sorted(per.items())},
            # top level: the dominant kernel (the keys the driver parses); every kernel of the step under "kernels"
            "roofline": dict(brick, traffic_note=tnote, kernels=kernels,
                             ba_iteration=dict(algorithmic_bytes=ba_alg, ms=ba_ms,
                                               frac=ba_alg / (ba_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ba_ms else None,
                                               definition="SURVEY 8d: (200 + 44) B per observation + (100 + 76) B per "
                                                          "lidar term over both passes of one iteration")),
        }
        out.update(extras)
        if cs:
            out["cloud_sharded"] = cs
            # both N > 1 decompositions side by side (north_star names the second): NN + association queries/s
            out["nn_modes"] = {"replicated_cloud_query_split": out.get("nn_queries_per_sec"),
                               "cloud_sharded_min_allreduce": cs.get("queries_per_sec")}
        if not a.no_cpu_baseline and world == 1:   # reported at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(xyz, nrm, q_all, mr_all, scene, min(a.cpu_sample_queries, Qtot),
                                               min(a.cpu_sample_points, Ptot))
        print(json.dumps(out))
    cloud.close()
    ba.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
