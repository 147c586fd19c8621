"""Hand-pack faiss IndexFlat files byte by byte - independently of lotus_amd.faiss_io - following the field order of faiss
``impl/index_write.cpp`` (``write_index``: fourcc; ``write_index_header``: d, ntotal, two dummies, is_trained, metric_type
[, metric_arg]; ``WRITEXBVECTOR(codes)``: size in 4-byte units, then the bytes).  The reference persists exactly this file
(``lotus/vector_store/faiss_vs.py:30``).  Output: tests/golden/faiss_flat_ip_3x4.index, faiss_flat_l2_2x3.index.

faiss itself cannot run in this image, so these bytes are a restatement too - but one made without the code under test."""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def pack(fourcc: bytes, d: int, rows, metric: int) -> bytes:
    out = bytearray()
    out += fourcc                                  # uint32 fourcc, written as its 4 characters
    out += struct.pack("<i", d)                    # int d
    out += struct.pack("<q", len(rows))            # idx_t ntotal
    out += struct.pack("<q", 1 << 20)              # idx_t dummy
    out += struct.pack("<q", 1 << 20)              # idx_t dummy
    out += struct.pack("<B", 1)                    # bool is_trained
    out += struct.pack("<i", metric)               # MetricType metric_type (METRIC_INNER_PRODUCT = 0, METRIC_L2 = 1)
    flat = [v for r in rows for v in r]
    out += struct.pack("<Q", len(flat))            # size_t: codes.size() / 4  (codes are the float32 rows as bytes)
    for v in flat:
        out += struct.pack("<f", v)
    return bytes(out)


IP_ROWS = [[1.0, 0.0, 0.0, 0.0], [0.0, 0.5, -0.25, 2.0], [3.0, 1.5, 0.125, -1.0]]
L2_ROWS = [[0.5, -1.0, 4.0], [8.0, 0.0, 0.0625]]

if __name__ == "__main__":
    open(os.path.join(HERE, "faiss_flat_ip_3x4.index"), "wb").write(pack(b"IxFI", 4, IP_ROWS, 0))
    open(os.path.join(HERE, "faiss_flat_l2_2x3.index"), "wb").write(pack(b"IxF2", 3, L2_ROWS, 1))
