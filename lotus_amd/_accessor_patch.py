"""Opt-in monkey-patch of the reference's pandas accessors (``lotus/sem_ops/sem_sim_join.py:84``,
``sem_search.py:91``, ``sem_dedup.py:32``) so that ``df.sem_sim_join(...)`` etc. run the loop-free
implementations of :mod:`lotus_amd.ops`.  Only active when the configured vector store is a ``HipVS``; any other
store falls through to the original accessor.

The cascade callers reach the hot path THROUGH these accessors - ``sem_filter``'s embedding proxy calls
``df.sem_search(col, instruction, K=len(df), return_scores=True)`` (``sem_filter.py:491-497``), ``sem_join``'s helper
calls ``l1_df.sem_sim_join(l2_df, K=len(l2), keep_index=True)`` (``sem_join.py:343-373``), ``sem_topk``'s "quick-sem"
sorts with ``sem_search(K=len(df))`` (``sem_topk.py:786-788``) - so patching the three accessors is what wires them to
``HipVS.scores()`` (K = all live rows: one score row, no top-k) and to the full-ranking path."""
from __future__ import annotations

_saved: dict = {}


def _is_hip(vs) -> bool:
    from .vs import HipVS

    return isinstance(vs, HipVS)


def install() -> None:
    if _saved:
        return
    import lotus
    from lotus.sem_ops.sem_dedup import SemDedupByDataframe
    from lotus.sem_ops.sem_search import SemSearchDataframe
    from lotus.sem_ops.sem_sim_join import SemSimJoinDataframe

    from . import ops

    _saved[SemSimJoinDataframe] = SemSimJoinDataframe.__call__
    _saved[SemSearchDataframe] = SemSearchDataframe.__call__
    _saved[SemDedupByDataframe] = SemDedupByDataframe.__call__

    def sim_join(self, other, left_on, right_on, K, lsuffix="", rsuffix="", score_suffix="", keep_index=False):
        if not _is_hip(lotus.settings.vs):
            return _saved[SemSimJoinDataframe](self, other, left_on, right_on, K, lsuffix=lsuffix, rsuffix=rsuffix,
                                               score_suffix=score_suffix, keep_index=keep_index)
        return ops.sem_sim_join(self._obj, other, left_on, right_on, K, lsuffix=lsuffix, rsuffix=rsuffix,
                                score_suffix=score_suffix, keep_index=keep_index)

    def search(self, col_name, query, K=None, n_rerank=None, return_scores=False, suffix="_sim_score"):
        if not _is_hip(lotus.settings.vs) or K is None or n_rerank is not None:
            return _saved[SemSearchDataframe](self, col_name, query, K=K, n_rerank=n_rerank,
                                              return_scores=return_scores, suffix=suffix)
        return ops.sem_search(self._obj, col_name, query, K, return_scores=return_scores, suffix=suffix)

    def dedup(self, col_name, threshold):
        if not _is_hip(lotus.settings.vs):
            return _saved[SemDedupByDataframe](self, col_name, threshold)
        return ops.sem_dedup(self._obj, col_name, threshold)

    # the reference wraps these three in @operator_cache (lotus/cache.py:33): keep that behaviour
    try:
        from lotus.cache import operator_cache
    except Exception:  # pragma: no cover - older checkouts
        def operator_cache(f):
            return f
    SemSimJoinDataframe.__call__ = operator_cache(sim_join)
    SemSearchDataframe.__call__ = operator_cache(search)
    SemDedupByDataframe.__call__ = operator_cache(dedup)


def uninstall() -> None:
    for cls, fn in _saved.items():
        cls.__call__ = fn
    _saved.clear()
