"""The oracle against the vectors the reference's own tests hold for the path (tests/golden/reference_kats.json):
projection_test.cc:95-124,180-222, pose_test.cc:168-186, camera_models_test.cc:133-217."""
import numpy as np

from tests import refkats


class OracleBackend:
    def __init__(self, po):
        self.po = po

    def observation_errors(self, model, cam, qvec, tvec, X, obs):
        n = len(X)
        ob = self.po.BA([model], [cam], [list(qvec) + list(tvec)], [0], X, np.zeros(n, np.int32), np.arange(n, dtype=np.int32), obs)
        return ob.observation_errors()

    def world_to_image(self, model, cam, uv):
        # identity pose, point (u, v, 1), observation 0: the residual IS WorldToImage(u, v)
        uv = np.asarray(uv, np.float64)
        n = len(uv)
        X = np.concatenate([uv, np.ones((n, 1))], axis=1)
        ob = self.po.BA([model], [cam], [[1, 0, 0, 0, 0, 0, 0]], [0], X, np.zeros(n, np.int32), np.arange(n, dtype=np.int32),
                        np.zeros((n, 2)))
        return ob.residuals().reshape(n, 2)


def _filter_tracks_oracle(po, model, cam, poses, points, obs_image, obs_point, obs_xy, max_err):
    ob = po.BA([model], [cam], poses, [0] * len(poses), points, np.asarray(obs_image, np.int32), np.asarray(obs_point, np.int32),
               np.asarray(obs_xy, np.float64).reshape(-1, 2))
    sq, depth = ob.observation_errors()
    return po.filter_tracks(sq, depth, np.asarray(obs_point, np.int32), len(points), max_err)


def test_filter_kats(oracle):
    """reconstruction_test.cc:394-445, :510-533, :599-614 on the oracle's restatement of the per-track filters"""
    class B(OracleBackend):
        def filter_tracks(self, *a):
            return _filter_tracks_oracle(self.po, *a)
    refkats.check_filters(B(oracle))


def test_squared_reprojection_error(oracle):
    refkats.check_squared_reprojection_error(OracleBackend(oracle))


def test_depth(oracle):
    refkats.check_depth(OracleBackend(oracle))


def test_quaternion_rotate_point(oracle):
    refkats.check_quaternion_rotate_point(OracleBackend(oracle))


def test_camera_model_round_trips(oracle):
    b = OracleBackend(oracle)
    refkats.check_camera_model_round_trips(b, b)
