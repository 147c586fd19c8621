"""CPU oracle for the LOTUS embedding-retrieval hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic the reference delegates to
``faiss-cpu`` 1.13.0 (reference ``uv.lock:573-574``; call sites
``lotus/vector_store/faiss_vs.py:14,23,24,63,64,67,75`` and
``lotus/utils.py:61,62,65``).  faiss is a third-party wheel that is neither
vendored under ``/root/reference`` nor installable here, so the algorithm is
restated from its published behaviour and anchored on the
reference's own call sites.

PARITY UNPINNED: the reference holds no golden vector, known-answer test or
fixture for this path (every test in ``.github/tests/rm_tests.py`` needs a
downloaded embedding model and asserts matched strings only), and faiss itself
cannot be run here.  The oracle is therefore pinned only by (i) hand-checkable
cases, (ii) agreement between its two independent implementations (numpy/BLAS
and plain C), (iii) an exact float64 brute-force cross-check and (iv) agreement
with scikit-learn's brute-force neighbours / Lloyd k-means (an independent
library, not the reference's dependency; tests/test_oracle.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; nothing under ``lotus_amd/`` does.
"""
from .flat import (METRIC_INNER_PRODUCT, METRIC_L2, flat_search, as_f32, pack_keys, unpack_keys,
                   flat_search_exact64)
from .kmeans import kmeans_faiss, rand_perm, KMeansResult, pair_distances, flipped_rows
from .dedup import range_self_join, dedup_components, dedup_keep_mask
