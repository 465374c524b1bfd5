"""Pins the oracle's block codecs (published-format restatements, oracle/oo_codecs.c) against
independent implementations available in this image: Python's zlib (raw DEFLATE) and
pyarrow.Codec (snappy raw, lz4 block, zstd frame).  The reference's crates (flate2, snap,
lz4_flex, zstd; Cargo.toml:41-49) are not vendored, so these are the codec-level anchors."""
import zlib

import numpy as np
import pytest

import oracle_lib as O

pa = pytest.importorskip("pyarrow")


def corpora():
    rng = np.random.default_rng(7)
    yield b""
    yield b"a"
    yield b"abcabcabcabcabcabcabcabcabcabc" * 50
    yield bytes(rng.integers(0, 256, 70000, dtype=np.uint8))  # incompressible
    yield bytes(rng.integers(0, 4, 300000, dtype=np.uint8))  # low entropy -> Huffman heavy
    yield bytes(np.repeat(rng.integers(0, 256, 3000, dtype=np.uint8), rng.integers(1, 200, 3000)))  # runs
    words = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK", b" carefully ", b" quickly final "]
    yield b"".join(words[i] for i in rng.integers(0, len(words), 40000))
    yield (np.arange(100000, dtype=np.int64) * 7).tobytes()


def test_inflate_raw_vs_zlib():
    for data in corpora():
        for level in (0, 1, 6, 9):
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = c.compress(data) + c.flush()
            assert O.codec("zlib", comp, len(data) + 16) == data
    # fixed-Huffman block
    c = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
    data = b"hello hello hello hello"
    assert O.codec("zlib", c.compress(data) + c.flush(), 64) == data
    assert O.codec("zlib", b"\x07\x00\x00", 64) is None  # reserved block type


@pytest.mark.parametrize("name,codec", [("snappy", "snappy"), ("lz4", "lz4_raw"), ("zstd", "zstd")])
def test_codec_vs_pyarrow(name, codec):
    if not pa.Codec.is_available(codec):
        pytest.skip(codec)
    levels = [None] if name != "zstd" else [1, 3, 9, 19]
    for data in corpora():
        if name == "lz4" and len(data) == 0:
            continue
        for lvl in levels:
            cd = pa.Codec(codec, compression_level=lvl) if lvl is not None else pa.Codec(codec)
            comp = cd.compress(data, asbytes=True)
            got = O.codec(name, comp, len(data) + 16)
            assert got == data, (name, lvl, len(data))


def test_truncated_blocks_fail():
    data = b"abcabcabcabcabcabcabcabcabcabc" * 50
    for name, codec in (("snappy", "snappy"), ("lz4", "lz4_raw"), ("zstd", "zstd")):
        comp = pa.Codec(codec).compress(data, asbytes=True)
        assert O.codec(name, comp[:-3], len(data) + 16) is None
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    assert O.codec("zlib", comp[:-2], len(data) + 16) is None


def test_chunked_stream_framing():
    """compression.rs:244-275: original chunks pass through, compressed ones are decoded,
    values may straddle chunk boundaries (scripts/write.py uses 32-byte chunks)."""
    payload = bytes(range(200))
    parts = []
    for i in range(0, 200, 32):
        blk = payload[i:i + 32]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(blk) + c.flush()
        if len(comp) < len(blk):
            hdr = len(comp) << 1
            parts.append(bytes([hdr & 0xFF, (hdr >> 8) & 0xFF, (hdr >> 16) & 0xFF]) + comp)
        else:
            hdr = (len(blk) << 1) | 1
            parts.append(bytes([hdr & 0xFF, (hdr >> 8) & 0xFF, (hdr >> 16) & 0xFF]) + blk)
    st, out = O.stream_decompress(b"".join(parts), "zlib", 32)
    assert st == O.OK and out == payload
    st, out = O.stream_decompress(payload, "none")
    assert st == O.OK and out == payload
