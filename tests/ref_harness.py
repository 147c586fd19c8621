"""Import the REAL reference package (``/root/reference/lotus``) in a container that lacks its heavy third-party
dependencies, by stubbing the module names it imports but the hot path never calls, and by
mapping ``faiss`` onto the oracle-backed shim ``tests/fake_faiss.py``.  Only used by tests; absent on the GPU box."""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

REFERENCE = "/root/reference"
_STUBS = ["sentence_transformers", "litellm", "litellm.types", "litellm.types.utils", "litellm.exceptions",
          "litellm.utils", "litellm.caching", "litellm.caching.caching", "openai", "openai._exceptions", "dotenv",
          "tiktoken", "backoff", "tokenizers"]


class _Stub(types.ModuleType):
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "lotus"))


def import_lotus():
    """-> the reference's `lotus` module (imported once)."""
    if "lotus" in sys.modules and getattr(sys.modules["lotus"], "__file__", "").startswith(REFERENCE):
        return sys.modules["lotus"]
    if not available():
        raise ImportError("reference checkout not present")
    for name in _STUBS:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _Stub(name)
    # the real faiss when a wheel is importable (then the reference's operators run on its own arithmetic); else the
    # faiss-shaped double over the CPU oracle (tests/fake_faiss.py; tests/test_faiss_pin.py holds the two to each other)
    try:
        if os.environ.get("LOTUS_TESTS_FORCE_FAKE_FAISS") == "1":
            raise ImportError("forced")
        import faiss as _real_faiss  # noqa: F401
    except ImportError:
        import fake_faiss

        sys.modules["faiss"] = fake_faiss
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    for k in [k for k in sys.modules if k == "lotus" or k.startswith("lotus.")]:
        del sys.modules[k]
    # lotus_amd.compat must re-bind to the real ABCs after lotus becomes importable
    for k in [k for k in sys.modules if k == "lotus_amd" or k.startswith("lotus_amd.")]:
        del sys.modules[k]
    import lotus  # noqa: F401

    return sys.modules["lotus"]
