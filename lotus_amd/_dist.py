"""The path's three exchange steps over ``torch.distributed``.

On a node of MI355X GPUs the process group is ``nccl`` (= RCCL over xGMI) and device tensors go straight into the
collective.  With any other backend (``gloo``: the CPU tests, or several ranks sharing ONE GPU in the 2-process GPU
test) device tensors are staged through host memory around the same collective, so the code path above these helpers -
shard offsets, device-side merge of the gathered candidate lists, device sums - is identical in both cases.

Only three collectives exist on the path:
  * ``all_gather_rows``  - per-shard candidate keys ``[Q,k]`` (8 B each) -> ``[world,Q,k]``  (search / sim-join),
                           per-shard cluster ids or score blocks (final k-means assignment, K = N callers);
  * ``all_reduce_sum_``  - ``[K,d]`` centroid sums + ``[K]`` counts + objective                (k-means iteration);
  * ``all_gather_object``- variable-length pair lists, host side                                (dedup).
"""
from __future__ import annotations


def context(shard: bool, pg=None):
    """-> (dist module or None, rank, world) for a sharded operator; (None, 0, 1) when not distributed."""
    if not shard:
        return None, 0, 1
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return None, 0, 1
    world = dist.get_world_size(pg)
    if world == 1:
        return None, 0, 1
    return dist, dist.get_rank(pg), world


def _staging_device(t, pg):
    """None when ``t`` can go into the collective as it is; otherwise the device to stage it on (RCCL takes device
    tensors only, gloo host tensors only)."""
    import torch
    import torch.distributed as dist

    backend = str(dist.get_backend(pg)).lower()
    has_nccl, has_gloo = "nccl" in backend, "gloo" in backend
    if t.is_cuda:
        return None if has_nccl else torch.device("cpu")
    return None if (has_gloo or not has_nccl) else torch.device("cuda", torch.cuda.current_device())


def all_gather_rows(t, pg=None):
    """``t`` (same shape on every rank) -> tensor ``[world, *t.shape]`` on ``t``'s device."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(pg)
    t = t.contiguous()
    stage = _staging_device(t, pg)
    src = t if stage is None else t.to(stage)
    out = torch.empty((world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
    if src.is_cuda:
        dist.all_gather_into_tensor(out, src, group=pg)  # one RCCL all-gather over xGMI
    else:
        dist.all_gather([out[r] for r in range(world)], src, group=pg)
    return out if stage is None else out.to(t.device)


def all_reduce_sum_(tensors, pg=None):
    """In-place sum over ranks of every tensor in ``tensors`` (device or host)."""
    import torch.distributed as dist

    for t in tensors:
        stage = _staging_device(t, pg)
        if stage is None:
            dist.all_reduce(t, group=pg)
        else:
            tmp = t.to(stage)
            dist.all_reduce(tmp, group=pg)
            t.copy_(tmp)
    return tensors


def barrier(pg=None) -> None:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier(group=pg)
