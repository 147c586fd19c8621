"""CPU model of the launch arithmetic of the grouped ``lvs_rq_kernel`` (lotus_amd/csrc/lvs_rq.hip: ``rq_max_ranges``,
``lvs_rq_fits``, ``lvs_rq_launch`` and the kernel's blockIdx -> (corpus range, query group) map), for the calls the literal
``sem_search`` loop and small joins make with 257 .. 4 096 queries (lotus/sem_ops/sem_search.py:120-138,
sem_sim_join.py:132-134).  The GPU parity rows of ``test_seeded_list_kernel`` run the code itself; this pins the rules the
comments state: one workgroup per (range, group), the siblings of a range on ONE XCD, never more workgroups than CUs."""
import itertools

import pytest

GROUPQ, MAXQ, CUS, XCDS = 256, 4096, 256, 8


def max_ranges(groups):
    return 256 if groups <= 1 else 8 * (32 // groups)


def fits(nq, nb):
    groups = -(-nq // GROUPQ)
    if not (96 < nq <= MAXQ and nb >= 32768):
        return False
    if groups == 1:
        return True
    return groups * max_ranges(groups) >= 224 and nb >= 32768 * groups


def launch(nq, nb):
    """-> (groups, ranges, blocks_per_wg, grid) as lvs_rq_launch computes them."""
    nblocks = -(-nb // 32)
    groups = max(1, -(-nq // GROUPQ))
    ranges = min(max_ranges(groups), (nblocks + 3) // 4)
    ranges = max(ranges, 1)
    bpw = -(-nblocks // ranges)
    ranges = -(-nblocks // bpw)
    grid = ranges if groups == 1 else 8 * groups * -(-ranges // 8)
    return groups, ranges, bpw, grid


def block_map(b, groups, ranges):
    """The kernel's prologue: workgroup b -> (range, group) or None (grid padding)."""
    if groups == 1:
        return b, 0
    s = b >> 3
    rng = (s // groups) * 8 + (b & 7)
    return (rng, s % groups) if rng < ranges else None


def test_group_counts_taken_are_the_ones_that_fill_the_chip():
    taken = [g for g in range(1, 17) if fits(g * GROUPQ, 32768 * 16)]
    assert taken == [1, 2, 3, 4, 5, 6, 7, 8, 10, 14, 15, 16]  # 9 and 11 - 13 groups would idle 40 - 80 CUs
    for g in taken[1:]:
        assert 224 <= g * max_ranges(g) <= CUS and max_ranges(g) % 8 == 0
    assert not fits(96, 10 ** 6) and fits(97, 10 ** 6) and fits(4096, 10 ** 6) and not fits(4097, 10 ** 6)
    assert not fits(512, 60_000) and fits(512, 65_536)  # every group's workgroups need a few blocks each


@pytest.mark.parametrize("nq,nb", [(300, 70_001), (512, 66_000), (700, 100_000), (1300, 200_000), (2560, 1_000_000),
                                   (4000, 530_000), (4096, 1_000_000), (3500, 500_000), (3840, 600_001), (257, 32768 * 2), (200, 40_000), (128, 33_000)])
def test_every_range_and_group_has_exactly_one_workgroup_and_siblings_share_an_xcd(nq, nb):
    assert fits(nq, nb)
    groups, ranges, bpw, grid = launch(nq, nb)
    assert grid <= CUS and ranges * bpw * 32 >= nb and (ranges - 1) * bpw * 32 < nb  # no empty range, whole corpus covered
    seen = {}
    for b in range(grid):
        m = block_map(b, groups, ranges)
        if m is not None:
            assert m not in seen
            seen[m] = b
    assert set(seen) == set(itertools.product(range(ranges), range(groups)))
    per_xcd = [0] * XCDS
    for (rng, _grp), b in seen.items():
        per_xcd[b % XCDS] += 1
        assert b % XCDS == seen[(rng, 0)] % XCDS          # a range's siblings: one XCD (one L2)
    assert max(per_xcd) <= CUS // XCDS                    # and never more workgroups on an XCD than it has CUs
    if groups > 1:
        for rng in range(ranges):                         # consecutive dispatch slots of that XCD
            slots = sorted(seen[(rng, g)] >> 3 for g in range(groups))
            assert slots == list(range(slots[0], slots[0] + groups))
    # the last group may be ragged; its queries are the call's last ones
    assert (groups - 1) * GROUPQ < nq <= groups * GROUPQ


# ---- round 6: chunked calls (lvs_rq_item / lvs_rq_ranges_for / lvs_rq_grid in lvs_tile.h; the chunk policy of lvs_capi.hip) -----
def ranges_for(groups):
    return 256 if groups <= 1 else (8 * (32 // groups) if groups <= 32 else 256 // groups)


def item(b, groups, nparts):
    """lvs_rq_item: workgroup b -> (range, group) or None."""
    if groups <= 1:
        return b, 0
    if groups <= 32:
        s = b >> 3
        rng = (s // groups) * 8 + (b & 7)
        return (rng, s % groups) if rng < nparts else None
    it = (b & 7) * 32 + (b >> 3)
    rng, grp = divmod(it, groups)
    return (rng, grp) if rng < nparts else None


def chunks(nq, chunk=32768):
    """The chunk policy: `chunk` queries while that many are left, then 8 192s, one 4 096, the rest."""
    out, left = [], nq
    while left > 0:
        cn = chunk if left >= chunk else (8192 if left >= 8192 else (4096 if left > 4096 else left))
        out.append(cn)
        left -= cn
    return out


@pytest.mark.parametrize("groups", [32, 64, 128, 256])
def test_beyond_32_groups_an_xcd_runs_32_items_of_one_range(groups):
    ranges = ranges_for(groups)
    assert groups * ranges == CUS  # one workgroup per CU, every CU busy
    seen = {}
    for b in range(CUS if groups > 32 else 8 * groups * -(-ranges // 8)):
        m = item(b, groups, ranges)
        assert m is not None and m not in seen
        seen[m] = b
    assert set(seen) == set(itertools.product(range(ranges), range(groups)))
    for x in range(XCDS):
        mine = [m for m, b in seen.items() if b % XCDS == x]
        assert len(mine) == 32
        if groups >= 32:
            assert len({rng for rng, _ in mine}) == 1  # the XCD's 32 workgroups stream ONE range through its L2
    # the earlier rule is unchanged up to 32 groups
    for g in range(2, 33):
        for b in range(8 * g * -(-ranges_for(g) // 8)):
            assert item(b, g, ranges_for(g)) == block_map(b, g, ranges_for(g))


@pytest.mark.parametrize("nq", [4097, 5000, 9000, 10_000, 20_000, 33_000, 40_000, 65_536, 100_000, 1_000_000])
def test_chunk_policy_covers_the_queries_with_launchable_group_counts(nq):
    cs = chunks(nq)
    assert sum(cs) == nq and all(c > 0 for c in cs)
    for c in cs[:-1]:
        assert c in (32768, 8192, 4096)
    for c in cs:
        groups = -(-c // GROUPQ)
        assert groups <= 32 or groups % 32 == 0  # what lvs_rq_launch / lvs_rj_launch accept
    assert cs == sorted(cs, reverse=True)
    assert chunks(100_000) == [32768, 32768, 32768, 1696]
