"""Randomised GPU parity sweep: 48 seeded shapes across every routing decision of `lvs_flat_search_keys` (streaming kernel,
128- and 256-query geometries, one and several passes, fp16 / hi|lo operands, IP / L2, explicit row ids and offsets)
against the CPU oracle.  Same bars as test_gpu_parity.py."""
import numpy as np
import pytest

import oracle
import synth
from lotus_amd import _capi

pytestmark = pytest.mark.gpu

F16, SPLIT = _capi.PACK_F16, _capi.PACK_SPLIT
IP, L2 = _capi.METRIC_IP, _capi.METRIC_L2


def _cases():
    rng = np.random.default_rng(20260923)
    out = []
    for i in range(48):
        regime = i % 6
        nq = [int(rng.integers(1, 33)), int(rng.integers(33, 129)), int(rng.integers(129, 700)),
              int(rng.integers(1, 33)), int(rng.integers(200, 900)), int(rng.integers(33, 300))][regime]
        nb = int(rng.integers(300, 9000)) if regime != 0 else int(rng.integers(4096, 30000))
        d = int(rng.choice([8, 64, 100, 200, 384, 768]))
        k = [int(rng.integers(1, 16)), int(rng.integers(1, 16)), int(rng.integers(1, 16)),
             int(rng.integers(16, 57)), int(rng.integers(16, 57)), int(rng.integers(57, 130))][regime]
        mode = SPLIT if rng.random() < 0.3 else F16
        metric = L2 if rng.random() < 0.35 else IP
        with_ids = bool(rng.random() < 0.25)
        out.append((i, nq, nb, d, k, mode, metric, with_ids))
    return out


@pytest.mark.parametrize("seed,nq,nb,d,k,mode,metric,with_ids", _cases())
def test_random_shapes_match_the_oracle(hip_backend, seed, nq, nb, d, k, mode, metric, with_ids):
    be = hip_backend
    xb = synth.corpus(nb, d, seed=1000 + seed)
    xq, _ = synth.queries(xb, nq, seed=2000 + seed)
    if metric == L2:
        xb = xb * np.float32(1.3)
    stored = (lambda x: x.astype(np.float16).astype(np.float32)) if mode == F16 else (lambda x: x.astype(np.float32))
    cb = be.pack(xb.astype(np.float16) if mode == F16 else xb, mode)
    cq = be.pack(xq.astype(np.float16) if mode == F16 else xq, mode)
    kw, relabel = {}, None
    if with_ids:  # a shard whose rows report explicit (shuffled, offset) ids
        relabel = (np.random.default_rng(seed).permutation(nb) + 7).astype(np.uint32)
        kw["row_ids"] = be.to_device(relabel.view(np.int32))
    keys = be.search_keys(cb, cq, k, metric, **kw)
    D, I = be.keys_to_result(keys, metric)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    Dr, Ir = oracle.flat_search(stored(xb), stored(xq), k, metric)
    if relabel is not None:
        Ir = np.where(Ir >= 0, relabel.astype(np.int64)[np.maximum(Ir, 0)], -1)
    atol = 1e-5  # (r6) one bar for both metrics
    err, hard, recall = synth.compare_topk(Dr, Ir, D, I, atol=atol)
    assert (I >= 0).sum() == (Ir >= 0).sum()
    assert err <= atol, f"score error {err}"
    assert hard == 0, f"{hard} id mismatches outside near-ties"
    assert recall >= 0.9999, recall
