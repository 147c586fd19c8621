"""Host-side MODELS of the three rules the exactness of the device path rests on - not the kernels, the RULES they implement -
run against brute force on random inputs.  First: the banded candidate lists behind the certified one-pass search (lvs_flat_search_keys_hi_banded +
lvs_certify_topk_banded) - not the kernel, the RULE it implements - run against brute force on random inputs:
whatever the order the rows arrive in, however they are cut into slabs with lists of their own, and whenever the slabs
publish their thresholds to each other, a query the certificate passes has its exact top k among the listed candidates."""
import numpy as np
import pytest


def banded_search(hi, slabs, k, k1, band, rng, publish=0.3):
    """hi: one-pass scores of all rows.  Every slab walks its rows in order with a list of k1 slots; a row is admitted while
    its score is >= max(last slot, k-th slot - band, shared threshold); slabs take turns row by row (as concurrent workgroups
    do) and now and then publish their threshold to the shared word.  -> merged list (row numbers, best first, <= k1)."""
    lists = [[] for _ in slabs]          # per slab: sorted [(score, -row)] best first
    pos = [0] * len(slabs)
    shared = -np.inf

    def own_threshold(lst):
        t = -np.inf
        if len(lst) >= k1:
            t = lst[k1 - 1][0]
        if len(lst) >= k:
            t = max(t, lst[k - 1][0] - band)
        return t

    live = [i for i, s in enumerate(slabs) if len(s)]
    while live:
        i = live[int(rng.integers(len(live)))]
        row = slabs[i][pos[i]]
        pos[i] += 1
        if pos[i] == len(slabs[i]):
            live.remove(i)
        s = hi[row]
        if s >= max(own_threshold(lists[i]), shared):
            lists[i].append((s, -row))
            lists[i].sort(reverse=True)
            del lists[i][k1:]
        if rng.random() < publish:
            shared = max(shared, own_threshold(lists[i]))
    merged = sorted((e for lst in lists for e in lst), reverse=True)[:k1]
    return [-r for _, r in merged]


def certified(hi, exact, cand, k, k1, bound, band_c):
    """lvs_certify_topk_banded: rows outside the list scored below max(last slot, k-th one-pass score - band_c * bound)."""
    if len(cand) < k:
        return True  # fewer than k rows exist
    s = np.sort(hi[cand])[::-1]
    x = s[k - 1] - band_c * bound
    if len(cand) >= k1:
        x = max(x, s[k1 - 1])
    kth_exact = np.sort(exact[cand])[::-1][k - 1]
    return kth_exact > x + bound


@pytest.mark.parametrize("seed", range(6))
def test_a_certified_query_has_its_exact_top_k_among_the_candidates(seed):
    rng = np.random.default_rng(seed)
    n_cert = n_open = 0
    for trial in range(120):
        n = int(rng.integers(20, 400))
        k = int(rng.integers(1, 8))
        k1 = k + int(rng.integers(1, 7))
        bound = float(rng.choice([1e-3, 1e-2, 5e-2]))
        spread = float(rng.choice([0.02, 0.2, 1.0]))            # from crowded bands to wide-open ones
        exact = rng.normal(0.0, spread, n)
        exact[rng.integers(0, n, 3)] = exact[0]                  # a few exact ties
        hi = exact + rng.uniform(-bound, bound, n)               # the one-pass score: within the bound of the exact one
        order = rng.permutation(n)
        nslab = int(rng.integers(1, 6))
        cutpts = np.sort(rng.integers(0, n + 1, nslab - 1))
        slabs = [list(p) for p in np.split(order, cutpts)]
        cand = banded_search(hi, slabs, k, k1, 2.05 * bound, rng, publish=float(rng.choice([0.0, 0.3, 1.0])))
        assert len(set(cand)) == len(cand) and len(cand) <= k1
        # inside the band the list is complete: every row within 2.05 bound of the k-th best one-pass score is listed
        if n >= k:
            kth = np.sort(hi)[::-1][k - 1]
            inside = set(np.flatnonzero(hi >= kth - 2.05 * bound).tolist())
            top_k1 = set(np.argsort(-hi, kind="stable")[:k1].tolist())
            assert (inside & top_k1) <= set(cand) or len(inside) > k1
        if certified(hi, exact, np.asarray(cand, dtype=np.int64), k, k1, bound, 2.02):
            n_cert += 1
            true_kth = np.sort(exact)[::-1][min(k, n) - 1]
            must = set(np.flatnonzero(exact > true_kth).tolist())       # every row strictly above the k-th exact score
            assert must <= set(cand), (seed, trial)
            got = np.sort(exact[cand])[::-1][:k]
            assert np.array_equal(got, np.sort(exact)[::-1][:k])         # the same k exact scores (ties: any of the equals)
        else:
            n_open += 1
    assert n_cert >= 30 and n_open >= 5   # the trials exercise both outcomes


def test_band_zero_is_the_plain_certificate_and_a_wide_band_the_plain_list():
    rng = np.random.default_rng(11)
    n, k, k1 = 300, 5, 9
    exact = rng.normal(0, 1, n)
    hi = exact + rng.uniform(-1e-3, 1e-3, n)
    slabs = [list(p) for p in np.array_split(rng.permutation(n), 4)]
    plain = list(np.argsort(-hi, kind="stable")[:k1])
    assert banded_search(hi, slabs, k, k1, 1e9, rng) == plain            # a band wider than any spread admits what a plain list admits
    cand = banded_search(hi, slabs, k, k1, 0.0, rng)                      # no band: only rows above the running k-th best
    assert cand[:k] == plain[:k]


# ---- the other two rules the exactness of the device path rests on, modelled the same way ---------------------------------
@pytest.mark.parametrize("seed", range(4))
def test_pooled_sample_thresholds_keep_the_merged_top_k_exact(seed):
    """lvs_flat_search_seed_scores -> all-gather -> lvs_flat_search_keys_seeded: every shard starts from the k-th
    largest of ALL shards' per-tile sample maxima.  Each of those values is the score of a real row of the searched set, so the
    threshold never exceeds the global k-th best: shards may return short lists, their merge is the brute-force top k."""
    rng = np.random.default_rng(100 + seed)
    for trial in range(60):
        world = int(rng.integers(1, 9))
        k = int(rng.integers(1, 12))
        tile = int(rng.choice([4, 16, 64]))
        sizes = rng.integers(0, 600, world)                      # uneven shards, some empty or shorter than a tile
        scores = [rng.normal(0, 1, int(s)) for s in sizes]
        if rng.random() < 0.3:                                   # duplicated rows across shards: ties AT the threshold
            for s in scores:
                if len(s) > 3:
                    s[:3] = 1.5
        tiles = int(rng.integers(1, 6))
        maxima = []
        for s in scores:                                         # a shard samples its first `tiles` whole tiles
            for t in range(tiles):
                blk = s[t * tile:(t + 1) * tile]
                maxima.append(blk.max() if len(blk) == tile else -np.inf)
        finite = sorted((m for m in maxima if np.isfinite(m)), reverse=True)
        thr = finite[k - 1] if len(finite) >= k else -np.inf     # seed_kth_kernel: no threshold without k sampled values
        parts = [np.sort(s[s >= thr])[::-1][:k] for s in scores]
        merged = np.sort(np.concatenate(parts))[::-1][:k]
        truth = np.sort(np.concatenate(scores))[::-1][:k]
        assert np.array_equal(merged, truth), (seed, trial)


@pytest.mark.parametrize("seed", range(4))
def test_two_candidate_certificate_returns_the_exact_nearest_row(seed):
    """lvs_nearest3 + lvs_nearest3_select + lvs_resolve_pairs: from the one-pass scores the kernel keeps best,
    second and third; best - second > 2 bound certifies the winner, else best - third > 2 bound leaves exactly two candidates
    (two exact dot products decide), else the row goes to the exact search.  Whatever the branch, the answer is the exact
    argmax (any of the equals on an exact tie)."""
    rng = np.random.default_rng(200 + seed)
    taken = {"certified": 0, "pair": 0, "open": 0}
    for trial in range(400):
        n = int(rng.integers(1, 40))
        bound = float(rng.choice([1e-3, 3e-2]))
        exact = rng.normal(0, float(rng.choice([0.02, 0.3])), n)
        if n > 2 and rng.random() < 0.3:
            exact[1] = exact[0] * (1 + 1e-9)                     # split twins
        hi = exact + rng.uniform(-bound, bound, n)
        order = np.argsort(-hi, kind="stable")
        best, second, third = (hi[order[i]] if i < n else -np.inf for i in range(3))
        if best - second > 2 * bound:
            winner, branch = order[0], "certified"
        elif best - third > 2 * bound:
            pair = order[:2]
            winner, branch = pair[np.argmax(exact[pair])], "pair"
        else:
            winner, branch = int(np.argmax(exact)), "open"       # the exact search over every row
        taken[branch] += 1
        assert exact[winner] == exact.max(), (seed, trial, branch)
    assert min(taken.values()) >= 10
