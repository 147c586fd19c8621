"""Step-by-step k-means parity harness shared by the GPU tests (real ``HipBackend``, configs[4]'s shape) and a CPU run over
the oracle-backed test double (small shape): the reference is ``oracle.kmeans_faiss`` with every iteration recorded
(``lotus/utils.py:61-62``)."""
import numpy as np

import oracle
from lotus_amd import _capi

F16, SPLIT, L2 = _capi.PACK_F16, _capi.PACK_SPLIT, _capi.METRIC_L2
_CFG5 = {}


def reference(x, K, nit):
    """oracle.kmeans_faiss over ``nit`` iterations with every iteration's centroids, assignment, cluster sizes and updated
    centroids kept.  ``x``: fp16 rows (identical values on both sides)."""
    trace = []
    ref = oracle.kmeans_faiss(x.astype(np.float32), K, niter=nit, final_assign=False, trace=trace)
    return dict(K=K, nit=nit, x=x, ref=ref, trace=trace, xt32=x[ref.train_ids].astype(np.float32))


def cfg5_reference():
    """Blob rows of SURVEY.md 8(d) (benchdata.blobs) at K = 1 024, d = 768, 300 000 rows (computed once per session)."""
    if not _CFG5:
        import benchdata

        x, _ = benchdata.blobs(benchdata.CFG_KMEANS, 300_000, 768, 1024)
        _CFG5.update(reference(x, 1024, 8))
    return _CFG5


def _assign_of(be, keys):
    _, I = be.keys_to_result(keys, L2)
    return I.reshape(-1).cpu().numpy()


def teacher_forced(be, c, max_flip_fraction=1e-4):
    """The ORACLE's centroids of iteration i -> ONE device step (certified one-pass assignment -> in-row-order sums ->
    objective -> division -> split replay).  Required per iteration:
      * EVERY disagreeing row is a near-tie: its distances to the two candidate centroids differ by <= 2e-5 (relative) in
        the oracle's own float32 arithmetic (oracle.flipped_rows) - the band inside which a different float32 summation
        order may pick either centroid;
      * assignment agreement >= 1 - 1e-4 over every row that is NOT choosing between the two halves of a
        cluster the previous iteration's split_clusters just cut in two: a fresh twin pair c (1 +- 1/1024) leaves every row
        of the old cluster nearly equidistant to both (measured on the blob rows: ~1 % of them inside the 2e-5 band), so
        rows whose two candidates are BOTH fresh twins may flip in any number - each of them still has to be a near-tie
        (the bullet above), and they are counted and printed;
      * cluster sizes equal except for the clusters those rows touch; objective within 1e-5;
      * the divided centroids of every untouched cluster bit-identical to the oracle's;
      * fed the oracle's assignment, the device's sums + division + split_clusters replay give the oracle's next centroids,
        cluster sizes and split count bit for bit (identical hassign => identical split decisions).
    -> number of flipped rows over all iterations."""
    import torch

    K, ref, trace, xt32 = c["K"], c["ref"], c["trace"], c["xt32"]
    packed = be.pack(c["x"], F16)
    train = be.gather(packed, be.to_device(ref.train_ids))
    nt = train.n
    x2 = train.norms.double().sum().reshape(1)
    total = 0
    per_iter = []
    for it, rec in enumerate(trace):
        c_dev = be.to_device(rec["centroids"])
        cpk, cstats = be.kmeans_pack_centroids(c_dev, SPLIT)
        keys = be.nearest(cpk, train, L2, exact_scores=False, corpus_stats=cstats)
        a_dev = _assign_of(be, keys)
        fl = oracle.flipped_rows(xt32, rec["centroids"], a_dev, rec["assign"])
        assert fl["all_near_ties"], (it, fl["rows"][~fl["near_tie"]][:5], fl["gaps"][~fl["near_tie"]][:5])
        # clusters the previous iteration's split_clusters touched (the re-seeded empty one and its donor): their centroid
        # after the update differs from the plain division
        twins = (np.flatnonzero(np.any(trace[it - 1]["next"] != trace[it - 1]["divided"], axis=1)) if it > 0
                 else np.zeros(0, np.int64))
        between_twins = np.isin(a_dev[fl["rows"]], twins) & np.isin(rec["assign"][fl["rows"]], twins)
        n_out = int((~between_twins).sum())
        assert n_out <= max_flip_fraction * nt, (it, n_out, len(fl["rows"]))
        assert int(between_twins.sum()) == 0 or int(ref.nsplit[it - 1]) > 0, it
        per_iter.append((it, int(len(fl["rows"])), n_out, float(fl["gaps"].max()) if len(fl["rows"]) else 0.0))
        total += len(fl["rows"])
        touched = np.union1d(a_dev[fl["rows"]], rec["assign"][fl["rows"]])
        clean = np.setdiff1d(np.arange(K), touched)
        # the device's own assignment: sums, sizes, objective, division
        sums, counts = be.kmeans_accumulate_keys(train, keys, K)
        assert np.array_equal(counts.cpu().numpy()[clean], rec["hassign"][clean]), it
        obj = torch.zeros(1, dtype=torch.float64, device=be.device)
        be.kmeans_objective(c_dev, sums, counts, x2, obj)
        assert abs(obj.item() / float(ref.obj[it]) - 1) <= 1e-5, (it, obj.item(), ref.obj[it])
        c_div = c_dev.clone()
        be.kmeans_update_centroids(sums, counts, c_div)
        assert np.array_equal(c_div.cpu().numpy()[clean], rec["divided"][clean]), it
        # the oracle's assignment through the device's sums, division and split replay
        sums_r, counts_r = be.kmeans_accumulate(train, be.to_device(rec["assign"]), K)
        c_next = c_dev.clone()
        ns = torch.zeros(1, dtype=torch.int32, device=be.device)
        be.kmeans_finish(sums_r, counts_r, c_next, nt, SPLIT, ns)
        assert int(ns.item()) == int(ref.nsplit[it]), it
        assert np.array_equal(counts_r.cpu().numpy(), rec["hassign_after"]), it
        assert np.array_equal(c_next.cpu().numpy(), rec["next"]), it
    print("teacher-forced flips per iteration (iteration, rows, of which NOT between fresh split twins, largest relative gap):",
          per_iter)
    return total


def divergence(be, c, tr, obj, nsplit):
    """A device run's per-iteration records ``tr`` (lotus_amd.cluster.kmeans(trace=...)) against the oracle's: identical
    assignments, centroids (bit for bit), split counts and objectives up to the first iteration in which any row is assigned
    differently - and in that iteration every differing row must be a near-tie.  -> dict for a report."""
    ref, trace, xt32 = c["ref"], c["trace"], c["xt32"]
    nit = min(len(tr), len(trace))
    rep = {"iterations_compared": nit, "first_divergence_iteration": None, "flipped_rows": 0, "all_flips_are_near_ties": True,
           "max_flip_rel_gap": 0.0, "centroids_bit_identical_until_divergence": True, "objective_max_rel_err": 0.0}
    for it in range(nit):
        same_c = bool(np.array_equal(np.asarray(tr[it]["centroids"]), trace[it]["centroids"]))
        rep["centroids_bit_identical_until_divergence"] &= same_c
        rep["objective_max_rel_err"] = max(rep["objective_max_rel_err"], abs(float(obj[it]) / float(ref.obj[it]) - 1))
        fl = oracle.flipped_rows(xt32, trace[it]["centroids"], np.asarray(tr[it]["assign"]), trace[it]["assign"])
        if len(fl["rows"]):
            rep.update(first_divergence_iteration=it, flipped_rows=int(len(fl["rows"])),
                       all_flips_are_near_ties=bool(fl["all_near_ties"]), max_flip_rel_gap=float(fl["gaps"].max()))
            break
        rep["split_counts_equal_until_divergence"] = bool(np.array_equal(nsplit[:it + 1], ref.nsplit[:it + 1]))
    return rep


def free_run(be, c):
    from lotus_amd.cluster import kmeans

    ref = c["ref"]
    tr = []
    r = kmeans(c["x"], c["K"], niter=c["nit"], backend=be, final_assign=False, trace=tr)
    assert np.array_equal(r.train_ids, ref.train_ids) and len(tr) == c["nit"]
    host = [{"centroids": t["centroids"].cpu().numpy(), "assign": _assign_of(be, t["keys"])} for t in tr]
    rep = divergence(be, c, host, r.obj, r.nsplit)
    assert rep["centroids_bit_identical_until_divergence"] and rep["objective_max_rel_err"] <= 1e-5, rep
    assert rep.get("split_counts_equal_until_divergence", True), rep
    it0 = rep["first_divergence_iteration"]
    after_split = it0 is not None and it0 > 0 and int(ref.nsplit[it0 - 1]) > 0  # fresh twins: see teacher_forced
    assert rep["all_flips_are_near_ties"] and rep["flipped_rows"] <= (2e-3 if after_split else 1e-4) * len(ref.train_ids), rep
    return rep
