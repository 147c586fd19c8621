"""lotus_amd - MI355X-native embedding-retrieval hot path for LOTUS (lotus-data/lotus).

Drop-in for the reference's ``lotus.vector_store.FaissVS`` (``lotus/vector_store/faiss_vs.py``) and
``lotus.utils.cluster`` (``lotus/utils.py:14-72``): hand-written HIP kernels for gfx950 behind a C ABI
(``include/lotus_hip.h``), driven from Python through ctypes.  There is no CPU fallback: every compute call
raises if ``liblotus_hip.so`` or a GPU is missing.
"""
from .compat import RM, VS, RMOutput, HAVE_LOTUS  # noqa: F401
from .vs import HipVS, METRIC_INNER_PRODUCT, METRIC_L2  # noqa: F401
from .rm import DeviceRM  # noqa: F401



def install(accessors: bool = False) -> None:
    """Plug the GPU path into an importable LOTUS: ``lotus.utils.cluster`` -> GPU k-means (always), and with
    ``accessors=True`` also ``df.sem_sim_join`` / ``df.sem_search`` / ``df.sem_dedup`` -> the loop-free
    implementations in :mod:`lotus_amd.ops` (same frames; ``sem_dedup`` keeps each group's FIRST value).
    ``lotus.settings.configure(vs=HipVS(), rm=...)`` remains the user's call.  :func:`uninstall` restores everything."""
    from . import cluster as _cluster

    _cluster.install()
    if accessors:
        from . import _accessor_patch

        _accessor_patch.install()


def uninstall() -> None:
    from . import _accessor_patch, cluster as _cluster

    _cluster.uninstall()
    _accessor_patch.uninstall()


__all__ = ["HipVS", "DeviceRM", "VS", "RM", "RMOutput", "METRIC_INNER_PRODUCT", "METRIC_L2", "HAVE_LOTUS", "install", "uninstall"]
__version__ = "0.1.0"
