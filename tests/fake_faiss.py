"""A faiss-shaped module backed by the CPU oracle (test infrastructure).

faiss-cpu cannot be installed here, so to run the reference's own ``FaissVS`` (``lotus/vector_store/faiss_vs.py``)
and ``lotus.utils.cluster`` (``lotus/utils.py:14-72``) UNMODIFIED this module provides exactly the symbols those files
touch: ``METRIC_INNER_PRODUCT``, ``METRIC_L2``, ``index_factory``, ``Index.add/search/ntotal/d``, ``write_index``,
``read_index``, ``Kmeans(d, k, niter=, verbose=).train(x)`` with ``.index.search`` and ``.centroids``."""
import numpy as np

import oracle
from lotus_amd import faiss_io

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


class Index:
    def __init__(self, d, metric):
        self.d = int(d)
        self.metric_type = metric
        self._x = np.zeros((0, self.d), np.float32)

    @property
    def ntotal(self):
        return self._x.shape[0]

    def add(self, x):
        x = np.ascontiguousarray(x, dtype="float32")
        assert x.shape[1] == self.d
        self._x = np.concatenate([self._x, x], axis=0)

    def reset(self):
        self._x = np.zeros((0, self.d), np.float32)

    def search(self, x, k):
        x = np.ascontiguousarray(x, dtype="float32")
        assert x.shape[1] == self.d
        assert k > 0
        return oracle.flat_search(self._x, x, k, self.metric_type)


def index_factory(d, description, metric=METRIC_L2):
    assert description == "Flat", "only the reference's default factory string is restated"
    return Index(d, metric)


def IndexFlatL2(d):
    return Index(d, METRIC_L2)


def IndexFlatIP(d):
    return Index(d, METRIC_INNER_PRODUCT)


def write_index(index, path):
    faiss_io.write_index_flat(path, index._x, index.metric_type)


def read_index(path):
    x, metric = faiss_io.read_index_flat(path)
    idx = Index(x.shape[1], metric)
    idx.add(x)
    return idx


class Kmeans:
    def __init__(self, d, k, niter=25, verbose=False, **kw):
        self.d, self.k, self.niter, self.verbose = d, k, niter, verbose
        self.seed = kw.get("seed", 1234)
        self.max_points_per_centroid = kw.get("max_points_per_centroid", 256)
        self.centroids = None
        self.index = None
        self.obj = None

    def train(self, x):
        r = oracle.kmeans_faiss(x, self.k, niter=self.niter, seed=self.seed,
                                max_points_per_centroid=self.max_points_per_centroid, final_assign=False)
        self.centroids = r.centroids
        self.obj = r.obj
        self.index = IndexFlatL2(self.d)
        self.index.add(r.centroids)
        return float(r.obj[-1]) if len(r.obj) else 0.0
