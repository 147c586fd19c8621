"""lotus_amd - MI355X-native embedding-retrieval hot path for LOTUS (lotus-data/lotus).

Drop-in for the reference's ``lotus.vector_store.FaissVS`` (``lotus/vector_store/faiss_vs.py``) and
``lotus.utils.cluster`` (``lotus/utils.py:14-72``): hand-written HIP kernels for gfx950 behind a C ABI
(``include/lotus_hip.h``), driven from Python through ctypes.  There is no CPU fallback: every compute call
raises if ``liblotus_hip.so`` or a GPU is missing.
"""
from .compat import RM, VS, RMOutput, HAVE_LOTUS  # noqa: F401
from .vs import HipVS, METRIC_INNER_PRODUCT, METRIC_L2  # noqa: F401

__all__ = ["HipVS", "VS", "RM", "RMOutput", "METRIC_INNER_PRODUCT", "METRIC_L2", "HAVE_LOTUS"]
__version__ = "0.1.0"
