"""The HIP path (through the C ABI) against the vectors the reference's own tests hold
(tests/golden/reference_kats.json): projection_test.cc:95-124,180-222, pose_test.cc:168-186,
camera_models_test.cc:133-217, reconstruction_test.cc:394-445, 510-533, 599-614.  Same checks as tests/test_reference_kats_cpu.py runs on the oracle."""
import numpy as np
import pytest

from tests import refkats
from tests.test_reference_kats_cpu import OracleBackend

pytestmark = pytest.mark.gpu


class GpuBackend:
    def __init__(self, gpu):
        self.gpu = gpu

    def _ba(self, model, cam, pose, X, obs):
        n = len(X)
        return self.gpu.BA([model], [cam], [pose], [0], X, np.zeros(n, np.int32), np.arange(n, dtype=np.int32), obs)

    def observation_errors(self, model, cam, qvec, tvec, X, obs):
        ba = self._ba(model, cam, list(qvec) + list(tvec), X, obs)
        out = ba.observation_errors()
        ba.close()
        return out

    def world_to_image(self, model, cam, uv):
        uv = np.asarray(uv, np.float64)
        n = len(uv)
        ba = self._ba(model, cam, [1, 0, 0, 0, 0, 0, 0], np.concatenate([uv, np.ones((n, 1))], axis=1), np.zeros((n, 2)))
        r = ba.evaluate(("residuals",))["residuals"].reshape(n, 2)
        ba.close()
        return r


def test_filter_kats(gpu):
    """reconstruction_test.cc:394-445, :510-533, :599-614 on the device's per-track filter reduce"""
    class B(GpuBackend):
        def filter_tracks(self, model, cam, poses, points, obs_image, obs_point, obs_xy, max_err):
            ba = self.gpu.BA([model], [cam], poses, [0] * len(poses), points, np.asarray(obs_image, np.int32),
                             np.asarray(obs_point, np.int32), np.asarray(obs_xy, np.float64).reshape(-1, 2))
            r = ba.filter_tracks(max_err)
            ba.close()
            return r
    refkats.check_filters(B(gpu))


def test_squared_reprojection_error(gpu):
    refkats.check_squared_reprojection_error(GpuBackend(gpu))


def test_depth(gpu):
    refkats.check_depth(GpuBackend(gpu))


def test_quaternion_rotate_point(gpu):
    refkats.check_quaternion_rotate_point(GpuBackend(gpu))


def test_camera_model_round_trips(gpu, oracle):
    refkats.check_camera_model_round_trips(GpuBackend(gpu), OracleBackend(oracle))
