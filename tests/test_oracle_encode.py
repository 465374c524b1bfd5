"""The encoder oracle (oracle/oo_encode.c) pinned: the reference's own writer vectors, the ORC specification's examples its reader
tests hold (an encoder following the same rules must reproduce them), and round trips through the pinned decoders."""
import numpy as np
import pytest

import oracle_lib as O


def test_reference_writer_vector_patched_base():
    # rle_v2/mod.rs:559-572 (writer_test_patched_base; data from the ORC v2 specification)
    lit = [2030, 2000, 2020, 1000000, 2040, 2050, 2060, 2070, 2080, 2090, 2100, 2110, 2120, 2130, 2140, 2150, 2160, 2170, 2180, 2190]
    expected = bytes([0x8e, 0x13, 0x2b, 0x21, 0x07, 0xd0, 0x1e, 0x00, 0x14, 0x70, 0x28, 0x32, 0x3c, 0x46, 0x50, 0x5a, 0x64, 0x6e,
                      0x78, 0x82, 0x8c, 0x96, 0xa0, 0xaa, 0xb4, 0xbe, 0xfc, 0xe8])
    assert O.enc_rle2_variable_run(lit, 8, signed=False) == expected
    assert O.enc_rle2(lit, 8, signed=False) == expected


def test_reference_writer_vector_direct_over_patched_base():
    # rle_v2/mod.rs:574-591 (writer_test_choose_direct_over_patched_base)
    lit = [0, 7, 6, 4, 5, 7, 0, 5, 6, 1, 4, 6, 5, 5, 3, 6, 7, 31, 17, 3]
    expected = bytes([0x4e, 0x13, 0, 7, 6, 4, 5, 7, 0, 5, 6, 1, 4, 6, 5, 5, 3, 6, 7, 31, 17, 3])
    assert O.enc_rle2_variable_run(lit, 8, signed=False) == expected


def test_specification_examples_the_reference_reads():
    # short repeat, rle_v2/mod.rs:713-716: 10000 x 5
    assert O.enc_rle2([10000] * 5, 8, signed=False) == bytes([0x0a, 0x27, 0x10])
    # direct, rle_v2/mod.rs:600-603
    assert O.enc_rle2([23713, 43806, 57005, 48879], 8, signed=False) == bytes([0x5e, 0x03, 0x5c, 0xa1, 0xab, 0x1e, 0xde, 0xad, 0xbe, 0xef])
    # delta, the specification's primes (rle_v2/delta.rs tests / mod.rs reader_test)
    assert O.enc_rle2([2, 3, 5, 7, 11, 13, 17, 19, 23, 29], 8, signed=False) == bytes([0xc6, 0x09, 0x02, 0x02, 0x22, 0x42, 0x42, 0x46])
    # byte RLE, byte.rs:344-355: 100 zeros; [0x44, 0x45]
    assert O.enc_byte_rle(np.zeros(100, np.uint8)) == bytes([0x61, 0x00])
    assert O.enc_byte_rle(np.array([0x44, 0x45], np.uint8)) == bytes([0xfe, 0x44, 0x45])
    # booleans, boolean.rs:181-203: 800 false; two literal bytes
    assert O.enc_boolean(np.zeros(100, np.uint8), 800) == bytes([0x61, 0x00])
    bits = [0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1]
    assert O.enc_boolean(np.packbits(bits, bitorder="little"), 16) == bytes([0xfe, 0b01000100, 0b01000101])


def _shapes(rng, n, nbits=64):
    yield "random64", rng.integers(-2**62, 2**62, n)
    yield "small", rng.integers(0, 100, n)
    yield "runs", np.repeat(rng.integers(-1000, 1000, n // 5 + 1), rng.integers(1, 14, n // 5 + 1))[:n]
    yield "long_runs", np.repeat(rng.integers(-5, 5, n // 300 + 1), rng.integers(1, 1400, n // 300 + 1))[:n]
    yield "ascending", np.cumsum(rng.integers(0, 50, n))
    yield "descending", -np.cumsum(rng.integers(0, 5, n))
    yield "steps", np.arange(n) * 7 - 300
    v = rng.integers(0, 200, n)
    v[rng.integers(0, n, max(1, n // 40))] = rng.integers(1 << (nbits // 2 - 2), 1 << (nbits - 4), max(1, n // 40))
    yield "outliers", v
    v = rng.integers(0, 200, n)
    v[:: 300] = 1 << (nbits - 6)
    yield "sparse_outliers", v
    yield "extremes", rng.choice(np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max, 0, -1, 1], dtype=np.int64), n)
    yield "mixed", np.concatenate([np.repeat(7, 600), rng.integers(0, 9, 50), np.arange(40), np.repeat(-3, 2), rng.integers(-9, 9, 700)])[:n]


@pytest.mark.parametrize("int_bytes", [2, 4, 8])
@pytest.mark.parametrize("signed", [True, False])
def test_round_trip_through_the_pinned_decoder(int_bytes, signed):
    rng = np.random.default_rng(5 + int_bytes)
    kinds = np.zeros(5, dtype=np.uint64)
    for n in (1, 2, 3, 4, 11, 511, 512, 513, 1025, 5000):
        for name, v in _shapes(rng, n, int_bytes * 8):
            v = np.asarray(v, dtype=np.int64)
            if int_bytes < 8:
                v = v.astype({2: np.int16, 4: np.int32}[int_bytes]).astype(np.int64)  # wrapped into N
            if not signed:
                v = np.abs(np.maximum(v, -(2**62)))  # lengths: never negative
                if int_bytes < 8:
                    v = np.minimum(v, (1 << (int_bytes * 8 - 1)) - 1)
            data, stats = O.enc_rle2(v, int_bytes, signed, with_stats=True)
            kinds += stats
            st, back = O.int_rle(data, len(v), version=2, signed=signed, nbits=int_bytes * 8)
            assert st == 0, (name, n)
            assert np.array_equal(back, v), (name, n)
    assert all(kinds[:4] > 0), kinds  # every sub-encoding was written


def test_byte_and_boolean_round_trips():
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 127, 128, 129, 130, 131, 260, 5000):
        for v in (rng.integers(0, 256, n), np.repeat(rng.integers(0, 256, n // 4 + 1), rng.integers(1, 9, n // 4 + 1))[:n],
                  np.repeat(rng.integers(0, 3, n // 100 + 1), rng.integers(1, 400, n // 100 + 1))[:n], np.zeros(n)):
            v = np.asarray(v, dtype=np.uint8)
            st, back = O.byte_rle(O.enc_byte_rle(v), len(v))
            assert st == 0 and np.array_equal(back.view(np.uint8), v)
            bits = (v & 1).astype(np.uint8)
            st, back = O.boolean(O.enc_boolean(np.packbits(bits, bitorder="little"), len(bits)), len(bits))
            assert st == 0 and np.array_equal(back.astype(np.uint8), bits)
