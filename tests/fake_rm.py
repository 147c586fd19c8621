"""Deterministic stand-in for an embedding model (tests only): a bag-of-words embedder whose word vectors are
seeded by the word's CRC and pulled towards a topic vector, so that the reference's scenario tests
(.github/tests/rm_tests.py) keep their meaning without downloading a model."""
import zlib

import numpy as np

TOPICS = {
    "math": ["probability", "random", "processes", "optimization", "methods", "engineering", "riemannian",
             "geometry", "math", "statistics", "linear", "algebra", "markov", "chains", "fundamentals"],
    "food": ["cooking", "food", "sciences", "gourmet", "home", "culinary", "basics"],
    "history": ["history", "atlantic", "world"],
    "potter": ["harry", "potter", "james"],
}
_WORD_TOPIC = {w: t for t, ws in TOPICS.items() for w in ws}
DIM = 64


def _unit(seed):
    v = np.random.default_rng(seed).standard_normal(DIM)
    return v / np.linalg.norm(v)


def word_vec(word):
    w = word.lower().strip(".,")
    v = _unit(zlib.crc32(w.encode()))
    t = _WORD_TOPIC.get(w)
    if t is not None:
        v = 0.6 * v + 0.8 * _unit(zlib.crc32(("topic:" + t).encode()))
    return v


def embed(texts, dtype=np.float32):
    out = []
    for t in texts:
        v = sum(word_vec(w) for w in str(t).split())
        out.append(v / np.linalg.norm(v))
    return np.asarray(out, dtype=dtype)


def make_rm(RM, dtype=np.float32):
    class FakeRM(RM):
        def __init__(self):
            super().__init__()
            self.calls = 0

        def _embed(self, docs):
            self.calls += 1
            if hasattr(docs, "tolist"):
                docs = docs.tolist()
            return embed(list(docs), dtype)

    return FakeRM()
