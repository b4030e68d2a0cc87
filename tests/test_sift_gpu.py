"""GPU parity of the MFMA SIFT matcher (through the C ABI) against the oracle restating
feature/sift.cc:55-204.  Bar: identical match lists (integer work: exact dot products, exact
best / second-best, first-index ties); the acos / ratio thresholds are float and evaluated with the
device libm, so a row whose test is decided by the last ulp could differ -- none is expected and the
test reports the margin if one ever shows up."""
import numpy as np
import pytest
import torch

from tests.test_sift_cpu import sift_reference_cases

pytestmark = pytest.mark.gpu


@pytest.fixture
def sift_tuning(gpu):
    """pcd_sift_set_tuning for one test (chunks per stripe walk, bytes of partials per sub-batch), reset afterwards"""
    def set_(nchunk=0, batch_partials=0):
        gpu.set_sift_tuning(nchunk, batch_partials)
    yield set_
    gpu.set_sift_tuning(0, 0)


def test_reference_known_answers_on_gpu(gpu, oracle):
    """expected counts of src/feature/sift_test.cc:296-428: 2, 0, 0, 0, 50, 50, 48, 49, 50, 48"""
    for name, d1, d2, opt, expected in sift_reference_cases(oracle):
        m = gpu.sift_match(d1, d2, **opt)
        assert len(m) == expected, (name, len(m), expected)
        assert np.array_equal(m, oracle.sift_match(d1, d2, **opt)[0]), name


@pytest.mark.parametrize("n1,n2", [(1, 1), (127, 129), (128, 128), (300, 77), (1000, 1500), (17000, 260)])
@pytest.mark.parametrize("cross", [True, False])
def test_random_descriptors_exact(gpu, oracle, n1, n2, cross):
    rng = np.random.default_rng(n1 * 7 + n2)
    # SIFT-like: squared uniforms, L2-normalised, x512; second set = noisy permuted copy + distractors
    f = rng.random((max(n1, n2), 128), dtype=np.float32) ** 2
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    base = np.clip(np.round(512 * f), 0, 255).astype(np.uint8)
    d1 = base[:n1].copy()
    noisy = np.clip(base.astype(np.int32) + rng.integers(-6, 7, base.shape), 0, 255).astype(np.uint8)
    d2 = noisy[rng.permutation(max(n1, n2))[:n2]].copy()
    if n2 > 10:
        d2[3] = d2[5]                       # duplicate column: ties -> first index, ratio test fails
    if n1 > 10:
        d1[7] = 0                           # all-zero descriptor: never matches
    for ratio in (0.8, 0.95):
        exp, e12, e21 = oracle.sift_match(d1, d2, max_ratio=ratio, cross_check=cross)
        got = gpu.sift_match(d1, d2, max_ratio=ratio, cross_check=cross)
        assert np.array_equal(got, exp), (n1, n2, cross, ratio, len(got), len(exp))
    if n1 >= 100:
        assert len(exp) > 10               # the test set really produces matches


@pytest.mark.parametrize("nchunk", [1, 2, 5])
def test_long_walks_with_ties(gpu, oracle, nchunk, sift_tuning):
    """the stripe kernel's running row-direction state over MANY column tiles (pcd_sift_set_tuning forces the number of
    column chunks; the default picks one tile per chunk at these sizes): few distinct descriptors => equal best
    scores in different tiles, blocks, lanes and chunks, so every tie rule of the end-of-walk merge decides"""
    sift_tuning(nchunk=nchunk)
    rng = np.random.default_rng(100 + nchunk)
    n1, n2 = 700, 1900                                   # 6 row tiles x 15 column tiles
    base = rng.integers(0, 90, (6, 128), dtype=np.uint8)
    d1 = base[rng.integers(0, 6, n1)].copy()
    d2 = base[rng.integers(0, 6, n2)].copy()
    d2[::3] = np.clip(d2[::3].astype(np.int32) + rng.integers(0, 2, (len(d2[::3]), 128)), 0, 255).astype(np.uint8)
    d1[5] = 0
    d2[1234] = 255
    for cross, ratio, dist in ((True, 0.8, 0.7), (False, 1.0, 3.2), (True, 1.0, 3.2)):
        exp, e12, e21 = oracle.sift_match(d1, d2, max_ratio=ratio, max_distance=dist, cross_check=cross)
        got = gpu.sift_match(d1, d2, max_ratio=ratio, max_distance=dist, cross_check=cross)
        assert np.array_equal(got, exp), (nchunk, cross, ratio, len(got), len(exp))
    # the batched entry on the same two sets (both orders)
    pairs = np.array([[0, 1], [1, 0]], np.uint32)
    res = gpu.sift_match_batch([d1, d2], pairs, max_ratio=1.0, max_distance=3.2, cross_check=True)
    assert np.array_equal(res[0], oracle.sift_match(d1, d2, max_ratio=1.0, max_distance=3.2, cross_check=True)[0])
    assert np.array_equal(res[1], oracle.sift_match(d2, d1, max_ratio=1.0, max_distance=3.2, cross_check=True)[0])


def test_one_way_results_and_asymmetric_layout_check(gpu, oracle):
    """device API: m12 / m21 equal the oracle's one-way results; the inputs are asymmetric (different sizes,
    different contents per row and per column) so a transposed or permuted MFMA result layout cannot pass"""
    rng = np.random.default_rng(5)
    n1, n2 = 200, 333
    d1 = rng.integers(0, 60, (n1, 128), dtype=np.uint8)
    d2 = rng.integers(0, 60, (n2, 128), dtype=np.uint8)
    d2[:150] = np.clip(d1[:150][::-1].astype(np.int32) + rng.integers(-2, 3, (150, 128)), 0, 255)
    exp, e12, e21 = oracle.sift_match(d1, d2, max_ratio=0.99, max_distance=3.0)
    t1, t2 = torch.from_numpy(d1).cuda(), torch.from_numpy(d2).cuda()
    m12 = torch.empty(n1, dtype=torch.int32, device="cuda")
    m21 = torch.empty(n2, dtype=torch.int32, device="cuda")
    mm = torch.empty(n1, 2, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    gpu.sift_match_device(t1, n1, t2, n2, m12, m21, mm, cnt, max_ratio=0.99, max_distance=3.0)
    torch.cuda.synchronize()
    assert np.array_equal(m12.cpu().numpy(), e12) and np.array_equal(m21.cpu().numpy(), e21)
    assert int(cnt.item()) == len(exp) and np.array_equal(mm.cpu().numpy()[: len(exp)].astype(np.uint32), exp)
    assert (e12[:150] == np.arange(149, -1, -1)).mean() > 0.9


def test_full_size_pair_properties(gpu):
    """8192 x 8192 (the reference's max_num_features, feature/sift.h:59): size-independent properties --
    matching a set against a permutation of itself recovers the permutation; cross-checked matches are
    symmetric; swapping the roles of the two sets transposes the result"""
    rng = np.random.default_rng(11)
    n = 8192
    f = rng.random((n, 128), dtype=np.float32) ** 2
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    d1 = np.clip(np.round(512 * f), 0, 255).astype(np.uint8)
    perm = rng.permutation(n)
    d2 = d1[perm]
    m = gpu.sift_match(d1, d2)
    assert len(m) == n and np.array_equal(perm[m[:, 1]], m[:, 0])
    mt = gpu.sift_match(d2, d1)
    assert np.array_equal(mt[np.argsort(mt[:, 1])][:, ::-1], m)


def _sift_like_images(rng, sizes):
    """images that really share features: every image is a noisy subset of one pool of SIFT-like descriptors"""
    pool_n = max(max(sizes), 1) * 2
    f = rng.random((pool_n, 128), dtype=np.float32) ** 2
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    pool = np.clip(np.round(512 * f), 0, 255).astype(np.int32)
    out = []
    for n in sizes:
        pick = rng.permutation(pool_n)[:n]
        out.append(np.clip(pool[pick] + rng.integers(-5, 6, (n, 128)), 0, 255).astype(np.uint8))
    return out


@pytest.mark.parametrize("cross,budget", [(True, None), (False, None), (True, "20000")])
def test_batch_equals_pair_by_pair_and_oracle(gpu, oracle, cross, budget, sift_tuning):
    """pcd_sift_match_batch (SiftFeatureMatcher::Match(image_pairs), feature/matching.cc:798) over a ragged set of
    images incl. an empty one, a 1-descriptor one, sizes that are not tile multiples, a pair of an image with
    itself and repeated pairs: every list equals the single-pair entry and the oracle.  budget = a small bound on
    the partial-result scratch, so the pair list is cut into many sub-batches (launch sets)."""
    if budget:
        sift_tuning(batch_partials=int(budget))
    rng = np.random.default_rng(11)
    sizes = [300, 0, 1, 129, 700, 128, 515]
    imgs = _sift_like_images(rng, sizes)
    pairs = [(a, b) for a in range(len(sizes)) for b in range(len(sizes)) if a <= b] + [(4, 0), (0, 4), (6, 3)]
    got = gpu.sift_match_batch(imgs, pairs, cross_check=cross)
    assert len(got) == len(pairs)
    total = 0
    for (a, b), g in zip(pairs, got):
        single = gpu.sift_match(imgs[a], imgs[b], cross_check=cross)
        assert np.array_equal(g, single), (a, b, len(g), len(single))
        exp = oracle.sift_match(imgs[a], imgs[b], cross_check=cross)[0]
        assert np.array_equal(g, exp), (a, b)
        total += len(g)
    assert total > 500
    assert gpu.sift_match_batch(imgs, np.zeros((0, 2), np.uint32)) == []


def test_batch_device_form_and_sub_batches(gpu, oracle):
    """device form with caller-chosen list offsets (3 unused slots between the lists stay untouched)"""
    rng = np.random.default_rng(12)
    sizes = [2048, 1500, 2048, 900]
    imgs = _sift_like_images(rng, sizes)
    arena = np.concatenate(imgs, axis=0)
    first = np.zeros(len(sizes) + 1, np.uint64)
    first[1:] = np.cumsum(sizes)
    pairs = np.array([(a, b) for a in range(4) for b in range(4) if a != b], np.uint32)
    n1 = np.array([sizes[a] for a, _ in pairs], np.uint64)
    off = np.zeros(len(pairs), np.uint64)
    off[1:] = np.cumsum(n1 + 3)[:-1]          # caller's own layout: 3 unused slots between the lists
    d_arena = torch.from_numpy(arena).cuda()
    d_m = torch.full((int(off[-1] + n1[-1]), 2), -1, dtype=torch.int32, device="cuda")
    d_c = torch.full((len(pairs),), -7, dtype=torch.int32, device="cuda")
    gpu.sift_match_batch_device(d_arena, first, pairs, d_m, off, d_c)
    torch.cuda.synchronize()
    m, c = d_m.cpu().numpy(), d_c.cpu().numpy()
    for p, (a, b) in enumerate(pairs):
        exp = oracle.sift_match(imgs[a], imgs[b])[0]
        assert c[p] == len(exp), (a, b)
        assert np.array_equal(m[int(off[p]): int(off[p]) + c[p]].astype(np.uint32), exp), (a, b)
        assert (m[int(off[p]) + int(n1[p]): int(off[p]) + int(n1[p]) + 3] == -1).all() or p == len(pairs) - 1


def test_batch_full_size_block(gpu):
    """a block of the exhaustive matcher at the reference's maximum image size (8192 descriptors,
    feature/sift.h:59): 6 images -> 15 pairs; lists equal the single-pair entry (itself oracle-checked above)."""
    rng = np.random.default_rng(13)
    imgs = _sift_like_images(rng, [8192] * 6)
    pairs = [(a, b) for a in range(6) for b in range(a + 1, 6)]
    got = gpu.sift_match_batch(imgs, pairs)
    for (a, b), g in zip(pairs, got):
        assert np.array_equal(g, gpu.sift_match(imgs[a], imgs[b])), (a, b)
    assert sum(len(g) for g in got) > 15 * 2000


def test_config5_exhaustive_sweep_450_images(gpu):
    """BASELINE.json config 5 end to end: 450 images x 8192 descriptors, every pair once (101 025 pairs), block by
    block as ExhaustiveFeatureMatcher::Run does (feature/matching.cc:902-960, block_size 50), each block one
    pcd_sift_match_batch_device call.  Checked: the block enumeration covers every unordered pair exactly once; every
    pair of a sample (in every block) has exactly the match list of the single-pair entry (itself oracle-checked
    above); all lists are strictly ascending in the first index and inside their capacity."""
    import time
    rng = np.random.default_rng(2024)
    n_img, n_desc = 450, 8192
    f = rng.random((2 * n_desc, 128), dtype=np.float32) ** 2
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    pool = torch.from_numpy(np.clip(np.round(512 * f), 0, 255).astype(np.int16)).cuda()
    g = torch.Generator(device="cuda").manual_seed(7)
    arena = torch.empty((n_img * n_desc, 128), dtype=torch.uint8, device="cuda")
    for i in range(n_img):   # every image: a noisy subset of one pool (the images really share features)
        pick = torch.randperm(2 * n_desc, device="cuda", generator=g)[:n_desc]
        noise = torch.randint(-5, 6, (n_desc, 128), device="cuda", generator=g, dtype=torch.int16)
        arena[i * n_desc:(i + 1) * n_desc] = (pool[pick] + noise).clamp_(0, 255).to(torch.uint8)
    first = np.arange(n_img + 1, dtype=np.uint64) * np.uint64(n_desc)
    max_pairs = 50 * 50
    d_m = torch.empty((max_pairs * n_desc, 2), dtype=torch.int32, device="cuda")
    d_c = torch.empty(max_pairs, dtype=torch.int32, device="cuda")
    m12 = torch.empty(n_desc, dtype=torch.int32, device="cuda")
    m21 = torch.empty(n_desc, dtype=torch.int32, device="cuda")
    mm = torch.empty((n_desc, 2), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    seen = np.zeros((n_img, n_img), bool)
    total_pairs = total_matches = 0
    t_batch = 0.0
    for pairs in gpu.exhaustive_blocks(n_img, 50):
        P = len(pairs)
        if P == 0:
            continue
        assert P <= max_pairs
        a, b = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
        assert not seen[a, b].any() and not seen[b, a].any()
        seen[a, b] = True
        off = np.arange(P, dtype=np.uint64) * np.uint64(n_desc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gpu.sift_match_batch_device(arena, first, pairs, d_m, off, d_c)
        torch.cuda.synchronize()
        t_batch += time.perf_counter() - t0
        c = d_c[:P].cpu().numpy()
        assert (c >= 0).all() and (c <= n_desc).all()
        total_pairs += P
        total_matches += int(c.sum())
        for p in rng.choice(P, 2, replace=False):   # two pairs of every block against the single-pair entry
            i, j = int(a[p]), int(b[p])
            gpu.sift_match_device(arena[i * n_desc:(i + 1) * n_desc], n_desc, arena[j * n_desc:(j + 1) * n_desc], n_desc,
                                  m12, m21, mm, cnt)
            k = int(cnt.item())
            assert k == int(c[p]), (i, j, k, int(c[p]))
            got = d_m[p * n_desc: p * n_desc + k]
            assert torch.equal(got, mm[:k]), (i, j)
            assert bool((got[1:, 0] > got[:-1, 0]).all())
    assert total_pairs == n_img * (n_img - 1) // 2 == 101025
    assert (seen | seen.T | np.eye(n_img, dtype=bool)).all()
    assert total_matches > 1000 * total_pairs        # the images share features: thousands of matches per pair
    print(f"config 5: {total_pairs} pairs in {t_batch:.2f} s of batched calls, {total_matches / total_pairs:.0f} matches per pair")
