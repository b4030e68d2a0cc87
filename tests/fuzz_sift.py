"""(test infrastructure, run by hand; not collected by pytest)  Randomised parity sweep of the SIFT matcher:
python tests/fuzz_sift.py [seconds] [seed].  Random image sizes (0, 1, around the 128-row tile, up to 1500), SIFT-like
and adversarial descriptors (duplicates, all-zero rows, saturated rows), random max_ratio / max_distance / cross_check;
single-pair entry vs the oracle (feature/sift.cc:55-204), batched entry vs the single-pair entry."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd")); sys.path.insert(0, ROOT)
import numpy as np
import pcdhip
from oracle import pyoracle as oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
# FUZZ_SIFT_NCHUNK / FUZZ_SIFT_BATCH_PARTIALS: pcd_sift_set_tuning for the whole sweep (chunks per stripe walk, bytes of
# partial results per sub-batch)
pcdhip.set_sift_tuning(int(os.environ.get("FUZZ_SIFT_NCHUNK", "0")), int(os.environ.get("FUZZ_SIFT_BATCH_PARTIALS", "0")))
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget


def image(n, pool):
    if n == 0:
        return np.zeros((0, 128), np.uint8)
    kind = rng.choice(["sift", "sift", "random", "few"])
    if kind == "sift":
        d = np.clip(pool[rng.integers(0, pool.shape[0], n)] + rng.integers(-8, 9, (n, 128)), 0, 255).astype(np.uint8)
    elif kind == "random":
        d = rng.integers(0, 256, (n, 128), dtype=np.uint8)
    else:   # few distinct rows: ties everywhere
        base = rng.integers(0, 80, (4, 128), dtype=np.uint8)
        d = base[rng.integers(0, 4, n)]
    if n > 3 and rng.random() < 0.5:
        d[rng.integers(0, n)] = 0
        d[rng.integers(0, n)] = 255
        d[rng.integers(0, n)] = d[rng.integers(0, n)]
    return np.ascontiguousarray(d)


ncase = 0
while time.time() < t_end:
    f = rng.random((600, 128), dtype=np.float32) ** 2
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    pool = np.clip(np.round(512 * f), 0, 255).astype(np.int32)
    sizes = [int(rng.choice([0, 1, 2, 64, 127, 128, 129, 255, 300, 700, 1500])) for _ in range(int(rng.integers(2, 6)))]
    imgs = [image(n, pool) for n in sizes]
    opt = dict(max_ratio=float(rng.choice([0.6, 0.8, 0.95, 1.0])), max_distance=float(rng.choice([0.3, 0.7, 1.2, 3.2])),
               cross_check=bool(rng.integers(0, 2)))
    pairs = [(int(rng.integers(0, len(imgs))), int(rng.integers(0, len(imgs)))) for _ in range(int(rng.integers(1, 8)))]
    got = pcdhip.sift_match_batch(imgs, pairs, **opt)
    for (a, b), g in zip(pairs, got):
        exp = oracle.sift_match(imgs[a], imgs[b], **opt)[0]
        single = pcdhip.sift_match(imgs[a], imgs[b], **opt)
        if not (np.array_equal(single, exp) and np.array_equal(g, exp)):
            print("MISMATCH", sizes, (a, b), opt, len(g), len(single), len(exp), flush=True)
            sys.exit(1)
    ncase += 1
    if ncase % 20 == 0:
        print("cases %d, %.0f s left" % (ncase, t_end - time.time()), flush=True)
print("OK: %d cases, no mismatch" % ncase)
