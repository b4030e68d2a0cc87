"""Clock / power while the 50-image SIFT block runs back to back (rocm-smi sampled from a thread): python tools/sift_power.py"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
import numpy as np, torch, pcdhip
n_img, n_desc = 50, 8192
rng = np.random.default_rng(0)
f = rng.random((n_desc, 128), dtype=np.float32) ** 2
f /= np.linalg.norm(f, axis=1, keepdims=True)
base = np.clip(np.round(512 * f), 0, 255).astype(np.int32)
arena = np.concatenate([np.clip(base[rng.permutation(n_desc)] + rng.integers(-5, 6, (n_desc, 128)), 0, 255).astype(np.uint8) for _ in range(n_img)], axis=0)
first = np.arange(n_img + 1, dtype=np.uint64) * np.uint64(n_desc)
pairs = np.array([(a, b) for a in range(n_img) for b in range(a + 1, n_img)], np.uint32)
P = len(pairs)
off = np.arange(P, dtype=np.uint64) * np.uint64(n_desc)
d_arena = torch.from_numpy(arena).cuda()
d_m = torch.empty(P * n_desc, 2, dtype=torch.int32, device="cuda")
d_c = torch.empty(P, dtype=torch.int32, device="cuda")
pcdhip.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c)
torch.cuda.synchronize()
stop = False
samples = []
def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
            samples.append([l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "power" in l])
        except Exception as e:
            samples.append([repr(e)])
        time.sleep(0.05)
th = threading.Thread(target=sample); th.start()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    pcdhip.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c)
e1.record()
torch.cuda.synchronize()
stop = True; th.join()
print("%d blocks: %.2f ms each" % (reps, e0.elapsed_time(e1) / reps))
for s in samples[:: max(1, len(samples) // 12)]:
    print(s)
