"""Plugin-surface types.  When LOTUS is importable the real ABCs are used (``sem_sim_join`` checks
``isinstance(vs, VS)``, ``lotus/sem_ops/sem_sim_join.py:101-106``); otherwise minimal local stand-ins with the
same abstract methods (``lotus/vector_store/vs.py:10-58``, ``lotus/models/rm.py:10-85``, ``lotus/types.py:232-235``)
keep the package usable on a box without LOTUS (e.g. the GPU test box)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any

HAVE_LOTUS = False
try:  # pragma: no cover - depends on the environment
    from lotus.models.rm import RM  # type: ignore
    from lotus.types import RMOutput  # type: ignore
    from lotus.vector_store.vs import VS  # type: ignore

    HAVE_LOTUS = True
except Exception:  # LOTUS (or one of its heavy dependencies) is not importable here

    @dataclass
    class RMOutput:  # type: ignore[no-redef]
        distances: Any
        indices: Any

    class VS(ABC):  # type: ignore[no-redef]
        """Vector-store plugin contract (4 abstract methods + ``index_dir``)."""

        def __init__(self) -> None:
            self.index_dir: str | None = None

        @abstractmethod
        def index(self, docs, embeddings, index_dir: str, **kwargs): ...

        @abstractmethod
        def load_index(self, index_dir: str): ...

        @abstractmethod
        def __call__(self, query_vectors, K: int, ids=None, **kwargs) -> RMOutput: ...

        @abstractmethod
        def get_vectors_from_index(self, index_dir: str, ids): ...

    class RM(ABC):  # type: ignore[no-redef]
        """Retriever (embedder) contract; only ``convert_query_to_query_vector``'s ndarray pass-through matters
        to the hot path (``lotus/models/rm.py:77-78``)."""

        @abstractmethod
        def _embed(self, docs): ...

        def __call__(self, docs):
            return self._embed(docs)

        def convert_query_to_query_vector(self, queries):
            import numpy as np

            if isinstance(queries, np.ndarray):
                return queries
            if isinstance(queries, str):
                queries = [queries]
            elif hasattr(queries, "tolist"):
                queries = queries.tolist()
            return self._embed(queries)
