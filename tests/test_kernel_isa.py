"""Assumptions of the hand-scheduled NN kernels, checked on the ISA hipcc emits (no GPU needed): the next compiler bump
must not silently break bit-exactness (FLANN's arithmetic has no fused multiply-add), occupancy (128 VGPRs = 4 wavefronts
per SIMD), the vmcnt accounting of the LDS-DMA tile loops (a spill reload is a VMEM operation the hand-written
s_waitcnt vmcnt(N) would miscount) or the LDS budget (4 workgroups per CU)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@pytest.fixture(scope="module")
def nn_isa(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "nn.s"
    subprocess.check_call([HIPCC] + FLAGS + [os.path.join(ROOT, "colmap-pcd_amd", "csrc", "nn.hip"), "-o", str(out)])
    return out.read_text()


def _kernels(isa):
    """name -> (metadata dict, body text)"""
    meta = {}
    kernels = isa[isa.index("amdhsa.kernels:"):isa.index("amdhsa.target:")]
    for blk in kernels.split("\n  - ")[1:]:          # one YAML list entry per kernel (its own keys: 4 spaces deep)
        name = re.search(r"\n    \.name:\s+(\S+)", blk).group(1)
        get = lambda k: int(re.search(r"\n    \." + k + r":\s+(\d+)", blk).group(1))
        meta[name] = dict(vgpr=get("vgpr_count"), sgpr=get("sgpr_count"), lds=get("group_segment_fixed_size"),
                          scratch=get("private_segment_fixed_size"))
    body = {}
    for m in re.finditer(r"^(_ZN3pcd\w+):[^\n]*\n(.*?)\n\.Lfunc_end", isa, re.S | re.M):
        body[m.group(1)] = m.group(2)
    return meta, body


def _find(d, part):
    hits = [k for k in d if part in k]
    assert hits, (part, sorted(d)[:5])
    return hits


def test_no_fused_multiply_add_in_the_distance_kernels(nn_isa):
    meta, body = _kernels(nn_isa)
    for part in ("k_nn_brick", "k_nn_fallback", "k_nn_bruteforce"):
        for k in _find(body, part):
            fma = re.findall(r"\bv_(?:fma|mad|fmac|pk_fma|mac)\w*_f32", body[k])
            assert not fma, (k, fma[:3])


def test_brick_kernels_resources(nn_isa):
    meta, body = _kernels(nn_isa)
    clip = _find(meta, "k_nn_brick_clip")[0]
    old = _find(meta, "k_nn_brickILi8")[0]
    for k, lds in ((clip, 4 * 2 * 256 * 16 + 4 * 128 * 16), (old, 4 * 2 * 256 * 16)):   # tiles (+ the stage-A buffer)
        m = meta[k]
        assert m["vgpr"] <= 128, (k, m)              # 4 wavefronts per SIMD
        assert m["scratch"] == 0, (k, m)             # no spills: the tile loops count their VMEM operations by hand
        assert m["lds"] == lds, (k, m)               # 4 workgroups per CU
        assert "scratch_" not in body[k], k
    # the fallback walk keeps 8 wavefronts per SIMD
    for k in _find(meta, "k_nn_fallback"):
        assert meta[k]["vgpr"] <= 64 and meta[k]["scratch"] == 0, (k, meta[k])


def test_tile_loop_has_only_its_own_vmem_operations(nn_isa):
    """between a tile's LDS-DMA instructions and the counted wait that covers them the kernels may issue no other
    vector-memory instruction than LDS-DMA: `s_waitcnt vmcnt(4)` means "all but the 4 DMAs of the next tile" """
    meta, body = _kernels(nn_isa)
    for part in ("k_nn_brick_clip", "k_nn_brickILi8"):
        k = _find(body, part)[0]
        lines = body[k].split("\n")
        waits = [i for i, ln in enumerate(lines) if "s_waitcnt vmcnt(4)" in ln]
        assert waits, k
        for w in waits:
            # walk back over the 4 DMA instructions of the tile in flight; everything between them and the wait
            seen, j = 0, w - 1
            while j >= 0 and seen < 4:
                ln = lines[j].strip()
                if ln.startswith("global_load_lds_dwordx4"):
                    seen += 1
                elif re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", ln):
                    raise AssertionError(f"{k}: `{ln}` between a tile's DMAs and its counted wait (line {j})")
                elif re.match(r"\.LBB|s_cbranch|s_branch", ln):
                    break      # left the basic block: the tile was issued in the predecessor
                j -= 1


@pytest.fixture(scope="module")
def sift_isa(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "sift.s"
    subprocess.check_call([HIPCC] + FLAGS + [os.path.join(ROOT, "colmap-pcd_amd", "csrc", "sift.hip"), "-o", str(out)])
    return out.read_text()


def test_sift_walk_resources_and_fast_path(sift_isa):
    """the SIFT walk (csrc/sift.hip sift_rows): 4 wavefronts per SIMD with 8-wavefront workgroups (<= 128 VGPRs, two
    workgroups' LDS per CU), no spills (a reload in the block loop would put a vmcnt(0) in front of the tile's LDS-DMA),
    8 MFMAs and 8 LDS fragment reads per block, and the per-register test as compares into SGPR pairs -- not one
    v_cmp -> s_cbranch_vccz pair per register (3x the MFMAs' time)."""
    meta, body = _kernels(sift_isa)
    for part in ("k_sift_scores_batch", "k_sift_scores_stripe"):
        k = _find(meta, part)[0]
        m = meta[k]
        assert m["vgpr"] <= 128 and m["scratch"] == 0, (k, m)
        assert m["lds"] <= 80 * 1024, (k, m)
        b = body[k]
        assert "scratch_" not in b, k
        assert len(re.findall(r"\bv_mfma_i32_32x32x32_i8\b", b)) == 4 * 8, k          # 4 column blocks per tile
        assert len(re.findall(r"\bds_read_b128\b", b)) == 4 * 8, k                      # 4 fragments + 4 constant reads each
        assert len(re.findall(r"\bglobal_load_lds_dwordx4\b", b)) == 2 * 2, k           # prologue + loop, 2 per wavefront
        cmps = len(re.findall(r"\bv_cmp_gt_i32_e64 s\[", b)) + len(re.findall(r"\bv_cmp_gt_i32_e32 vcc", b))
        assert cmps >= 4 * 32, (k, cmps)                                                  # one compare per register
        assert len(re.findall(r"\bs_cbranch_vccz\b", b)) <= 8, k                         # (not one per register)
