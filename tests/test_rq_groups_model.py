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
