"""Golden fixtures (tests/golden/*.npz, generator committed next to them): the oracle must reproduce them bit-exactly
on the CPU, and the HIP path must match them on the GPU."""
import glob
import os

import numpy as np
import pytest

import oracle
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
FLAT = sorted(glob.glob(os.path.join(HERE, "golden", "flat_*.npz")))


def _load(p):
    z = np.load(p)
    ids = z["ids"] if z["ids"].size else None
    return z["xb"], z["xq"], int(z["k"]), int(z["metric"]), ids, z["D"], z["I"]


@pytest.mark.parametrize("path", FLAT, ids=[os.path.basename(p) for p in FLAT])
@pytest.mark.parametrize("use_c", [True, False])
def test_oracle_reproduces_flat_golden(path, use_c):
    xb, xq, k, metric, ids, D, I = _load(path)
    D2, I2 = oracle.flat_search(xb.astype(np.float32), xq.astype(np.float32), k, metric, ids=ids, use_c=use_c)
    assert np.array_equal(I, I2)
    assert np.allclose(D, D2, rtol=0, atol=2e-6)  # BLAS builds may differ in the last bits


def test_oracle_reproduces_kmeans_and_dedup_golden():
    z = np.load(os.path.join(HERE, "golden", "kmeans_blobs.npz"))
    r = oracle.kmeans_faiss(z["x"].astype(np.float32), int(z["k"]), niter=int(z["niter"]),
                            max_points_per_centroid=int(z["mppc"]))
    assert np.array_equal(r.train_ids, z["train_ids"])
    assert (r.assign == z["assign"]).mean() >= 0.999
    assert np.allclose(r.obj, z["obj"], rtol=1e-5)
    z = np.load(os.path.join(HERE, "golden", "dedup_pairs.npz"))
    pi, pj, ps = oracle.range_self_join(z["x"].astype(np.float32), float(z["thr"]))
    assert np.array_equal(pi, z["pi"]) and np.array_equal(pj, z["pj"])
    assert np.array_equal(oracle.dedup_components(len(z["x"]), pi, pj), z["labels"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FLAT, ids=[os.path.basename(p) for p in FLAT])
def test_hip_matches_flat_golden(path, hip_backend, tmp_path):
    from lotus_amd import HipVS

    xb, xq, k, metric, ids, D, I = _load(path)
    vs = HipVS(metric=metric, backend=hip_backend)
    vs.index(None, xb, str(tmp_path / "g"))  # fp16 embeddings -> fp16 storage, bit-identical values on the device
    out = vs(xq, k, ids=None if ids is None else ids.tolist())
    err, hard, recall = synth.compare_topk(D, I, out.distances, out.indices, atol=1e-5)
    assert err <= 1e-5 and hard == 0 and recall == 1.0
    assert np.array_equal(out.indices < 0, I < 0)
    if "dups" in path:
        assert np.array_equal(out.indices, I)  # exact ties: id-ascending, exactly as the oracle
