import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "colmap-pcd_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU checker (oracle/liboracle.so); built on demand with gcc/g++."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def pcdhip():
    import pcdhip as m
    return m


@pytest.fixture(scope="session")
def gpu(pcdhip):
    """The HIP library on a real device.  GPU tests must fail, not skip, when it is missing."""
    pcdhip.lib()
    assert pcdhip.device_count() >= 1, "no gfx950 device visible: -m gpu tests need the MI355X box"
    return pcdhip
