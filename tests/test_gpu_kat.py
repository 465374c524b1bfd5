"""The reference's own byte-level known-answer vectors (tests/kat_vectors.py = SURVEY.md Appendix B, plus the crafted
varying-DELTA run with a zero first delta) fed DIRECTLY through the C ABI to the HIP path: the decoded Arrow buffers
must hold the values the reference's unit tests expect -- no oracle in between.

Signed integer vectors are the DATA stream of a Long column; unsigned ones are the LENGTH stream of a Binary column
(unsigned RLE, array_decoder/string.rs:55-56) whose offsets then carry the values; byte RLE is a Byte column, boolean a
Boolean column, zigzag varints the DATA stream of a Decimal(38, 0) column with a constant-zero SECONDARY scale."""
import numpy as np
import pytest

import gpu_util as G
import kat_vectors as K
from orc_rust_amd import gen

pytestmark = pytest.mark.gpu

LONG, BINARY, BYTE, BOOLEAN, DECIMAL = 4, 8, 1, 0, 14
DATA, LENGTH, SECONDARY = 1, 2, 5


def values_of(res, ci, dtype):
    assert res.status()[0] == 0, res.status()
    return np.concatenate([np.frombuffer(res.batch(b, ci)["values"], dtype=dtype) for b in range(res.n_batches)])


@pytest.mark.parametrize("batch_size", [8192, 3])
def test_integer_rle_vectors(batch_size):
    for name, data, want, signed, version, _nbits in K.INT_RLE:
        n = len(want)
        enc = 2 if version == 2 else 0
        stream = np.array(data, dtype=np.uint8)
        if signed:
            res = G.gpu_decode(n, [{"column_id": 1, "orc_type": LONG, "encoding": enc}], [(1, DATA, stream)], batch_size=batch_size)
            assert values_of(res, 0, np.int64).tolist() == list(want), name
        else:
            blob = np.zeros(int(sum(want)), dtype=np.uint8)
            res = G.gpu_decode(n, [{"column_id": 1, "orc_type": BINARY, "encoding": enc}], [(1, LENGTH, stream), (1, DATA, blob)],
                               batch_size=batch_size)
            assert res.status()[0] == 0, (name, res.status())
            lens = np.concatenate([np.diff(res.batch(b, 0)["offsets"]) for b in range(res.n_batches)])
            assert lens.tolist() == list(want), name
        res.free()


def test_integer_rle_vectors_in_one_stream():
    """All signed vectors back to back in ONE stream (runs follow each other like in a real column)."""
    stream, want = [], []
    for name, data, exp, signed, version, _ in K.INT_RLE:
        if signed and version == 2:
            stream += list(data) * 3
            want += list(exp) * 3
    res = G.gpu_decode(len(want), [{"column_id": 1, "orc_type": LONG, "encoding": 2}], [(1, DATA, np.array(stream, dtype=np.uint8))], batch_size=100)
    assert values_of(res, 0, np.int64).tolist() == want
    res.free()


def test_byte_rle_vectors():
    for name, data, want in K.BYTE_RLE:
        res = G.gpu_decode(len(want), [{"column_id": 1, "orc_type": BYTE, "encoding": 0}], [(1, DATA, np.array(data, dtype=np.uint8))])
        assert values_of(res, 0, np.uint8).tolist() == want, name
        res.free()


def test_boolean_vectors():
    for name, data, want in K.BOOLEAN:
        n = len(want)
        res = G.gpu_decode(n, [{"column_id": 1, "orc_type": BOOLEAN, "encoding": 0}], [(1, DATA, np.array(data, dtype=np.uint8))])
        assert res.status()[0] == 0
        bits = np.unpackbits(np.frombuffer(res.batch(0, 0)["values"], dtype=np.uint8), bitorder="little")[:n]
        assert bits.tolist() == want, name
        res.free()


def test_varint_i128_vectors():
    for name, data, want in K.VARINT_I128:
        n = len(want)
        cols = [{"column_id": 1, "orc_type": DECIMAL, "encoding": 2, "precision": 38, "scale": 0}]
        streams = [(1, DATA, np.array(data, dtype=np.uint8)), (1, SECONDARY, gen.rle2(np.zeros(n, dtype=np.int64), signed=True))]
        res = G.gpu_decode(n, cols, streams)
        raw = values_of(res, 0, np.uint64).reshape(n, 2)
        got = [(int(lo) | (int(hi) << 64)) - ((1 << 128) if int(hi) >> 63 else 0) for lo, hi in raw]
        assert got == want, name
        res.free()


def test_rlev2_eof_is_out_of_spec():  # rle_v2/mod.rs:118-126: one value more than the stream holds
    res = G.gpu_decode(6, [{"column_id": 1, "orc_type": LONG, "encoding": 2}], [(1, DATA, np.array([0x0A, 0x27, 0x10], dtype=np.uint8))])
    assert res.status()[0] == 2
    res.free()
