"""Two host threads, each with its own cloud handle and stream, hammer the NN / association entry points at the same time;
results must equal the single-threaded ones: python tools/thread_check.py"""
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
print("errors:", errors[:5])
assert not errors
print("OK")
