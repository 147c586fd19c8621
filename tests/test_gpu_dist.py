"""The N>1 path on the REAL HIP backend: two processes share cuda:0, rendezvous over gloo, and
run every sharded operator - row-sharded search with id offsets, device-side merge of the gathered candidate lists
(short and long lists), sharded K = N ranking / score rows, k-means on a row-sharded index (device sums all-reduced,
per-shard final assignment), tile-dealt dedup - against the single-process CPU oracle.  On an 8-GPU node the only
difference is the transport inside ``lotus_amd/_dist.py`` (RCCL on device tensors instead of host staging)."""
import pytest

import dist_cases

pytestmark = pytest.mark.gpu


def test_sharded_operators_on_hip_backend_two_ranks_one_gpu(tmp_path):
    res = dist_cases.run(tmp_path, "hip")
    dist_cases.check(res, exact=False)


def test_2d_split_on_the_hip_backend(tmp_path):
    """HipVS(shard=(2, 2)): four processes (2 query groups x 2 corpus shards) on cuda:0 - merge inside a corpus group,
    concatenation across the query groups, k-means and score rows through the corpus sub-group."""
    res = dist_cases.run_2d(tmp_path, "hip")
    dist_cases.check_2d(res, exact=False)
