"""Two host threads, each with its own cloud handle and stream, hammer the NN / association entry points at the same time;
results must equal the single-threaded ones; then the batched SIFT entry points (shared per-device scratch, host +
device entry on a side stream) and the staged association path: python tools/thread_check.py"""
import os, sys, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colmap-pcd_amd"))
import numpy as np, torch, pcdhip
from pcdhip import synth

torch.zeros(1, device="cuda")
clouds, qs, refs = [], [], []
for t in range(2):
    xyz, nrm = synth.cloud_planes(300_000 + 50_000 * t, seed=10 + t)
    c = pcdhip.Cloud(xyz, nrm, raw_lidar_frame=False)
    q = synth.queries(xyz, 90_000 + 40_000 * t, seed=20 + t)
    clouds.append(c); qs.append(q); refs.append((c.nn(q), c.nn(q[:3000])))
errors = []


def work(t):
    try:
        st = torch.cuda.Stream()
        dq = torch.from_numpy(qs[t]).cuda()
        keys = torch.empty(len(qs[t]), dtype=torch.int64, device="cuda")
        for it in range(60):
            with torch.cuda.stream(st):
                clouds[t].nn_device(dq, len(qs[t]), keys, pcdhip.NN_AUTO, st.cuda_stream)
            st.synchronize()
            k = keys.cpu().numpy().view(np.uint64)
            idx = (k & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            if not np.array_equal(idx[refs[t][0][2] != 0], refs[t][0][0][refs[t][0][2] != 0]):
                errors.append((t, it, "grid path"))
            small = clouds[t].nn(qs[t][:3000])
            if not np.array_equal(small[0], refs[t][1][0]):
                errors.append((t, it, "one-launch path"))
    except Exception as e:   # noqa: BLE001
        errors.append((t, repr(e)))


th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
[x.start() for x in th]; [x.join() for x in th]
print("NN errors:", errors[:5])
assert not errors

# ---- batched SIFT: two host threads share the device's scratch; one uses the host entry point, the other the
# device entry point on a NON-BLOCKING side stream (the pair table / partial results must not be overwritten under
# a call that is still running: ADVICE r2) ----
rng = np.random.default_rng(5)
descs = [[rng.integers(0, 256, (int(rng.integers(900, 1400)), 128)).astype(np.uint8) for _ in range(5)] for _ in range(2)]
for d in descs:            # plant matches so the lists are not empty
    for i in range(1, 5):
        d[i][:300] = np.clip(d[0][:300].astype(np.int32) + rng.integers(-3, 4, (300, 128)), 0, 255).astype(np.uint8)
pairs = np.array([(a, b) for a in range(5) for b in range(a + 1, 5)], np.uint32)
want = [[pcdhip.sift_match(descs[t][a], descs[t][b]) for a, b in pairs] for t in range(2)]


def sift_host(t):
    try:
        for it in range(25):
            got = pcdhip.sift_match_batch(descs[t], pairs)
            for p in range(len(pairs)):
                if not np.array_equal(got[p], want[t][p]):
                    errors.append((t, it, p, "sift host batch")); return
    except Exception as e:   # noqa: BLE001
        errors.append((t, repr(e)))


def sift_dev(t):
    try:
        st = torch.cuda.Stream()
        arena, first = pcdhip._sift_arena(descs[t])
        d_arena = torch.from_numpy(arena).cuda()
        n1 = (first[1:] - first[:-1])[pairs[:, 0]]
        off = np.zeros(len(pairs) + 1, np.uint64); off[1:] = np.cumsum(n1)
        d_m = torch.zeros(int(off[-1]) * 2 + 2, dtype=torch.int32, device="cuda")
        d_c = torch.zeros(len(pairs), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for it in range(25):
            with torch.cuda.stream(st):
                pcdhip.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c, stream=st.cuda_stream)
                pcdhip.sift_match_batch_device(d_arena, first, pairs[::-1].copy(), d_m, off, d_c, stream=st.cuda_stream)
                pcdhip.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c, stream=st.cuda_stream)
            st.synchronize()
            m = d_m.cpu().numpy().view(np.uint32); c = d_c.cpu().numpy()
            for p in range(len(pairs)):
                g = m[2 * int(off[p]): 2 * (int(off[p]) + int(c[p]))].reshape(-1, 2)
                if not np.array_equal(g, want[t][p]):
                    errors.append((t, it, p, "sift device batch")); return
    except Exception as e:   # noqa: BLE001
        errors.append((t, repr(e)))


th = [threading.Thread(target=sift_host, args=(0,)), threading.Thread(target=sift_dev, args=(1,))]
[x.start() for x in th]; [x.join() for x in th]
print("SIFT errors:", errors[:5])
assert not errors

# ---- staged association (pinned staging owned by the handle): one handle per thread ----
ref_hits = []
for t in range(2):
    sq, smr = clouds[t].staging(len(qs[t]))
    sq[:] = qs[t]; smr[:] = 1.2
    ref_hits.append(clouds[t].associate_staged(len(qs[t]), len(qs[t]), pcdhip.GATE_MAPPER_LOCAL).copy())


def staged(t):
    try:
        for it in range(40):
            h = clouds[t].associate_staged(len(qs[t]), len(qs[t]), pcdhip.GATE_MAPPER_LOCAL)
            if h.tobytes() != ref_hits[t].tobytes():
                errors.append((t, it, "staged association")); return
    except Exception as e:   # noqa: BLE001
        errors.append((t, repr(e)))


th = [threading.Thread(target=staged, args=(t,)) for t in range(2)]
[x.start() for x in th]; [x.join() for x in th]
print("staged errors:", errors[:5])
assert not errors
print("OK")
