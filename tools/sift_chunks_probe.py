import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
import numpy as np, torch, pcdhip
n = 8192
rng = np.random.default_rng(0)
f = rng.random((n, 128), dtype=np.float32) ** 2
f /= np.linalg.norm(f, axis=1, keepdims=True)
d = np.clip(np.round(512 * f), 0, 255).astype(np.uint8)
t1 = torch.from_numpy(d).cuda(); t2 = torch.from_numpy(d[rng.permutation(n)].copy()).cuda()
m12 = torch.empty(n, dtype=torch.int32, device="cuda"); m21 = torch.empty(n, dtype=torch.int32, device="cuda")
mm = torch.empty(n, 2, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
for nc in (0, 2, 4, 8, 16, 32, 64):
    pcdhip.set_sift_tuning(nc, 0)
    for _ in range(3): pcdhip.sift_match_device(t1, n, t2, n, m12, m21, mm, cnt)
    torch.cuda.synchronize()
    pcdhip.profile_enable(True); pcdhip.profile_reset()
    for _ in range(30): pcdhip.sift_match_device(t1, n, t2, n, m12, m21, mm, cnt)
    torch.cuda.synchronize()
    p = pcdhip.profile_get(); pcdhip.profile_enable(False)
    print("nchunk", nc, {k: round(t / c * 1e3, 1) for k, (c, t) in p.items()}, int(cnt.item()))
for n_small in (1000, 2048, 4096):
    a, b = t1[:n_small], t2[:n_small]
    for nc in (0, 1, 2, 4, 8, 16):
        pcdhip.set_sift_tuning(nc, 0)
        for _ in range(3): pcdhip.sift_match_device(a, n_small, b, n_small, m12, m21, mm, cnt)
        torch.cuda.synchronize()
        pcdhip.profile_enable(True); pcdhip.profile_reset()
        for _ in range(30): pcdhip.sift_match_device(a, n_small, b, n_small, m12, m21, mm, cnt)
        torch.cuda.synchronize()
        p = pcdhip.profile_get(); pcdhip.profile_enable(False)
        print("n", n_small, "nchunk", nc, "scores us", round(p["sift_scores"][1] / p["sift_scores"][0] * 1e3, 1))
