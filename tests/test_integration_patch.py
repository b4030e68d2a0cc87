"""integration/colmap-pcd-hip.patch is a real unified diff against the reference tree: it must
apply cleanly to the files it touches and be exactly what integration/make_patch.py generates.  Compiling the patched
tree needs COLMAP's dependencies (Qt, PCL, FLANN, Ceres, Eigen, Boost ...), none of which is installed here; the
adapters the patch calls are compiled and tested in colmap-pcd_amd/shim/test_shim.cc instead.  Skipped where the
reference tree is absent (the GPU box)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "integration", "colmap-pcd-hip.patch")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")


def _touched_files():
    files = []
    for line in open(PATCH, encoding="utf-8"):
        if line.startswith("--- a/"):
            files.append(line[6:].strip())
    return files


def test_patch_applies_to_the_reference_tree(tmp_path):
    files = _touched_files()
    assert {"src/optim/bundle_adjustment.cc", "src/sfm/incremental_mapper.cc", "src/controllers/bundle_adjustment.cc",
            "src/util/option_manager.cc", "src/lidar/ply.cc", "src/lidar/hip_backend.h"} <= set(files)
    for f in files:
        src = os.path.join(REF, f)
        if os.path.exists(src):                      # hip_backend.h is a new file
            dst = tmp_path / f
            dst.parent.mkdir(parents=True, exist_ok=True)
            shutil.copy(src, dst)
    subprocess.check_call(["git", "init", "-q", "."], cwd=tmp_path)
    r = subprocess.run(["git", "apply", "--check", "--verbose", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    subprocess.check_call(["git", "apply", PATCH], cwd=tmp_path)
    ba = (tmp_path / "src/optim/bundle_adjustment.cc").read_text(encoding="utf-8")
    # the three association loops, LoadPointcloud's device index, Solve / Add*ToProblem, the option
    assert ba.count("hip_blocks_->AddReprojection(") == 5 and ba.count("hip_blocks_->AddLidar(") == 1
    assert "problem_options.evaluation_callback = hip_blocks_.get();" in ba
    assert "MatchClosestLidarPoints(reconstruction_, closest_ids, closest_ranges)" in \
        (tmp_path / "src/sfm/incremental_mapper.cc").read_text(encoding="utf-8")
    assert "PCD_GATE_MAPPER_GLOBAL" in (tmp_path / "src/sfm/incremental_mapper.cc").read_text(encoding="utf-8")
    assert "PCD_GATE_CONTROLLER" in (tmp_path / "src/controllers/bundle_adjustment.cc").read_text(encoding="utf-8")
    assert "InitializeFromRawCloud" in (tmp_path / "src/lidar/ply.cc").read_text(encoding="utf-8")
    assert '"Mapper.lidar_backend"' in (tmp_path / "src/util/option_manager.cc").read_text(encoding="utf-8")
    # every shim symbol the patched code calls exists in the shim headers of this repository
    shim = "".join(open(os.path.join(ROOT, "colmap-pcd_amd", "shim", h), encoding="utf-8").read()
                   for h in ("lidar_hip.h", "ceres_adapter.h"))
    for sym in ("MatchClosestLidarPointsFlat", "ToLidarPoint", "HipBlockRecorder", "AddReprojection", "AddLidar",
                "Finalize", "HipBackendEnabled", "InitializeFromRawCloud"):
        assert sym in shim, sym


def test_patch_is_what_the_generator_writes(tmp_path):
    gen = tmp_path / "make_patch.py"
    shutil.copy(os.path.join(ROOT, "integration", "make_patch.py"), gen)
    subprocess.check_call([sys.executable, str(gen), REF], cwd=tmp_path, stdout=subprocess.DEVNULL)
    assert (tmp_path / "colmap-pcd-hip.patch").read_bytes() == open(PATCH, "rb").read()
