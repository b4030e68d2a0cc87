"""reference-signature C++ adapters (colmap-pcd_amd/shim): structure logic on CPU, device paths on the GPU"""
import os
import subprocess

import pytest

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colmap-pcd_amd")


def _build():
    subprocess.check_call(["make", "-s", "-C", PKG, "shim/test_shim"])
    return os.path.join(PKG, "shim", "test_shim")


def test_shim_structure_cpu():
    r = subprocess.run([_build()], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_shim_gpu():
    r = subprocess.run([_build(), "--gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr
