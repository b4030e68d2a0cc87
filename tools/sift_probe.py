"""Throughput of the MFMA SIFT matcher on one image pair: python tools/sift_probe.py [n1] [n2] [reps]
(config 5 of BASELINE.json: 8192 x 128 uint8 descriptors per image, exhaustive pairs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import pcdhip  # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rng = np.random.default_rng(0)
f = rng.random((max(n1, n2), 128), dtype=np.float32) ** 2
f /= np.linalg.norm(f, axis=1, keepdims=True)
d = np.clip(np.round(512 * f), 0, 255).astype(np.uint8)
t1 = torch.from_numpy(d[:n1].copy()).cuda()
t2 = torch.from_numpy(d[rng.permutation(max(n1, n2))[:n2]].copy()).cuda()
m12 = torch.empty(n1, dtype=torch.int32, device="cuda")
m21 = torch.empty(n2, dtype=torch.int32, device="cuda")
mm = torch.empty(n1, 2, dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
for _ in range(3):
    pcdhip.sift_match_device(t1, n1, t2, n2, m12, m21, mm, cnt)
torch.cuda.synchronize()
pcdhip.profile_enable(True)
pcdhip.profile_reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    pcdhip.sift_match_device(t1, n1, t2, n2, m12, m21, mm, cnt)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
p = pcdhip.profile_get()
pcdhip.profile_enable(False)
for k, (n, t) in p.items():
    print("  %-16s %8.3f ms/launch" % (k, t / n))
sc = p["sift_scores"][1] / p["sift_scores"][0]
ops = 2.0 * 128 * n1 * n2              # useful = one S = D1.D2^T (round 4: every tile is multiplied twice, once per direction)
print("pair %d x %d: %.3f ms -> %.0f pairs/s; %d matches" % (n1, n2, ms, 1e3 / ms, int(cnt.item())))
print("k_sift_scores: %.3f ms, %.1f useful TOP/s int8 MFMA (executed = 2 x useful), dense i8 peak ~5000 TOP/s" % (sc, ops / sc / 1e9))

# ---- a block of the exhaustive matcher through the batched entry (SiftFeatureMatcher::Match(image_pairs),
# feature/matching.cc:798; ExhaustiveMatchingOptions::block_size = 50 images) ----
ONLY = os.environ.get("PROBE_ONLY_BLOCK", "0") == "1"   # profiling: just the 50-image block
for n_img, n_desc in ((50, n1),) if ONLY else ((16, n1), (50, n1), (50, 2048)):
    first = np.arange(n_img + 1, dtype=np.uint64) * np.uint64(n_desc)
    base = np.clip(np.round(512 * f[:n_desc]), 0, 255).astype(np.int32)
    arena = np.concatenate([np.clip(base[rng.permutation(n_desc)] + rng.integers(-5, 6, (n_desc, 128)), 0, 255).astype(np.uint8)
                            for _ in range(n_img)], axis=0)
    pairs = np.array([(a, b) for a in range(n_img) for b in range(a + 1, n_img)], np.uint32)
    P = len(pairs)
    off = np.arange(P, dtype=np.uint64) * np.uint64(n_desc)
    d_arena = torch.from_numpy(arena).cuda()
    d_m = torch.empty(P * n_desc, 2, dtype=torch.int32, device="cuda")
    d_c = torch.empty(P, dtype=torch.int32, device="cuda")
    pcdhip.profile_enable(False)
    pcdhip.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c)
    torch.cuda.synchronize()
    e0.record()
    pcdhip.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    useful = 2.0 * 128 * n_desc * n_desc * P
    print("batch: %d images x %d descriptors, %d pairs: %.2f ms = %.3f ms/pair -> %.0f pairs/s, %.1f useful TOP/s "
          "(one S = D1.D2^T per pair), %d matches"
          % (n_img, n_desc, P, ms, ms / P, P / ms * 1e3, useful / ms / 1e9, int(d_c.sum().item())), flush=True)
    # the same pairs one call each (what the single-pair entry costs)
    if P <= 200:
        torch.cuda.synchronize()
        e0.record()
        for a, b in pairs:
            pcdhip.sift_match_device(d_arena[a * n_desc:(a + 1) * n_desc], n_desc, d_arena[b * n_desc:(b + 1) * n_desc], n_desc,
                                     m12[:n_desc], m21[:n_desc], mm[:n_desc], cnt)
        e1.record()
        torch.cuda.synchronize()
        print("       pair by pair: %.3f ms/pair" % (e0.elapsed_time(e1) / P))

# ---- the same block on a LOW-MATCH set: every image has its own random descriptors (no image shares a descriptor pool
# with another one), so almost nothing passes the distance / ratio tests -- the walk's tests see the same exchangeable
# score statistics, and the time should not depend on how many matches there are
n_img, n_desc = 50, n1
arena = np.concatenate([np.clip(np.round(512 * (lambda g: g / np.linalg.norm(g, axis=1, keepdims=True))(
    rng.random((n_desc, 128), dtype=np.float32) ** 2)), 0, 255).astype(np.uint8) for _ in range(n_img)], axis=0)
first = np.arange(n_img + 1, dtype=np.uint64) * np.uint64(n_desc)
pairs = np.array([(a, b) for a in range(n_img) for b in range(a + 1, n_img)], np.uint32)
P = len(pairs)
off = np.arange(P, dtype=np.uint64) * np.uint64(n_desc)
d_arena = torch.from_numpy(arena).cuda()
d_m = torch.empty(P * n_desc, 2, dtype=torch.int32, device="cuda")
d_c = torch.empty(P, dtype=torch.int32, device="cuda")
pcdhip.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c)
torch.cuda.synchronize()
e0.record()
pcdhip.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print("low-match batch: %d images x %d independent descriptors, %d pairs: %.2f ms = %.3f ms/pair, %.1f useful TOP/s, %d matches"
      % (n_img, n_desc, P, ms, ms / P, 2.0 * 128 * n_desc * n_desc * P / ms / 1e9, int(d_c.sum().item())), flush=True)
