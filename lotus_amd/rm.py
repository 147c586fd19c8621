"""Embedding producer hand-over.

The reference's ``SentenceTransformersRM._embed`` (``lotus/models/sentence_transformers_rm.py:49-76``) runs the encoder
on the GPU, then copies every batch to the host (``.cpu().numpy()``), stacks fp32 on the host, pickles it, and the
vector store copies it back - two PCIe crossings and a 3 GB pickle per million rows.  ``DeviceRM`` is the same ``RM``
plugin surface (``lotus/models/rm.py:10-85``) with the batches left where the encoder wrote them: ``_embed`` returns ONE
device tensor ``[n, d]``, which ``HipVS.index`` packs in place and ``HipVS.__call__`` takes as queries.

The encoder itself is out of scope: ``encode`` is any callable ``list[str] -> torch.Tensor [b, d]``
(e.g. ``lambda b: SentenceTransformer(...).encode(b, convert_to_tensor=True)``)."""
from __future__ import annotations

from typing import Callable

import numpy as np

from .compat import RM


class DeviceRM(RM):
    """``RM`` whose embeddings never leave the device.

    Args:
        encode: ``list[str] -> torch.Tensor [b, d]`` (any float dtype, on the device the vector store uses).
        max_batch_size: documents per ``encode`` call (``sentence_transformers_rm.py:27``).
        normalize_embeddings: L2-normalise rows on the device (``sentence_transformers_rm.py:30,71``).
        dtype: storage type handed on - ``"float16"`` halves the HBM and triples the search rate, ``"float32"`` keeps the
            encoder's precision (stored as an fp16 hi|lo pair by ``HipVS``)."""

    def __init__(self, encode: Callable, max_batch_size: int = 64, normalize_embeddings: bool = True,
                 dtype: str = "float32") -> None:
        super().__init__()
        if dtype not in ("float16", "float32"):
            raise ValueError("dtype must be 'float16' or 'float32'")
        self.encode = encode
        self.max_batch_size = int(max_batch_size)
        self.normalize_embeddings = bool(normalize_embeddings)
        self.dtype = dtype

    def _embed(self, docs):
        import torch

        if hasattr(docs, "tolist"):
            docs = docs.tolist()
        docs = list(docs)
        out = None
        for i in range(0, len(docs), self.max_batch_size):
            emb = self.encode(docs[i:i + self.max_batch_size])
            if not torch.is_tensor(emb):
                emb = torch.as_tensor(np.asarray(emb))
            if emb.dim() == 1:
                emb = emb[None, :]
            if out is None:  # one destination buffer, written batch by batch: no list of batches, no final stack
                out = torch.empty((len(docs), emb.shape[1]), dtype=getattr(torch, self.dtype), device=emb.device)
            e = emb.to(torch.float32)
            if self.normalize_embeddings:
                e = torch.nn.functional.normalize(e, dim=1)
            out[i:i + e.shape[0]] = e.to(out.dtype)
        if out is None:
            return torch.empty((0, 0), dtype=getattr(torch, self.dtype))
        return out

    def convert_query_to_query_vector(self, queries):
        """As ``RM.convert_query_to_query_vector`` (``rm.py:53-85``), plus: device tensors pass through like ndarrays."""
        if isinstance(queries, np.ndarray) or (hasattr(queries, "is_cuda") and hasattr(queries, "data_ptr")):
            return queries
        if isinstance(queries, str):
            queries = [queries]
        elif hasattr(queries, "tolist"):
            queries = queries.tolist()
        return self._embed(queries)
