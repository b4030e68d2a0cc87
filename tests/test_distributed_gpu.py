"""The world_size-2 decomposition tests of tests/test_distributed.py with the HIP library as the per-rank kernel: two
processes (gloo for the collectives), both on device 0 -- interleaved shards, spatially compact shards with the
two-phase search (pcd_nn_query_device + pcd_nn_refine_device) incl. ties across shards, track-sharded BA blocks."""
import pytest

from tests.test_distributed import run_two_ranks

pytestmark = pytest.mark.gpu


def test_two_rank_decomposition_hip(gpu):
    run_two_ranks(use_hip=True)
