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


def test_lzo1x_round_trips_and_rejections():
    """LZO1X (compression.rs:174-183).  No LZO library in the image: the oracle's decoder is pinned at file level by the
    reference's two LZO fixtures (test_oracle_files.py); here it must undo tests/lzo_enc.py (every instruction form) and
    reject what lzokay rejects."""
    import lzo_enc
    for data in corpora():
        for m2 in (True, False):
            comp = lzo_enc.compress(data, use_m2=m2)
            assert O.codec("lzo", comp, len(data) + 16) == data, (len(data), m2)
    data = b"abcabcabcabcabcabcabcabcabcabc" * 50
    comp = lzo_enc.compress(data)
    assert O.codec("lzo", comp[:-3], len(data) + 16) is None          # no end marker
    assert O.codec("lzo", comp + b"\x00", len(data) + 16) is None      # input left over
    assert O.codec("lzo", comp, len(data) - 1) is None                 # output overrun
    assert O.codec("lzo", b"\x11\x00", 16) is None                     # shorter than the end marker
    assert O.codec("lzo", b"\x16abcde" + b"\x20\x40\x00" + b"\x11\x00\x00", 64) is None  # look-behind before the output
    # the state-dependent short matches: 2 bytes from <= 1 KiB after 1..3 literals, 3 bytes from 2049.. after a literal run
    rng = np.random.default_rng(5)
    lits = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
    s = bytearray(lzo_enc._literal_run(lits))                          # state 4
    d = 2049 + 3 * 4 + 1                                               # distance (inst >> 2) + (H << 2) + 2049
    s += bytes([(1 << 2) | 2, 3])                                      # D = 1, H = 3, S = 2: three bytes from 2062 back, then 2 literals
    s += b"XY"                                                         # state 2
    s += bytes([(2 << 2) | 0, 1])                                      # D = 2, H = 1: two bytes from 2 + 4 + 1 = 7 back, S = 0
    s += b"\x11\x00\x00"
    want = bytearray(lits)
    want += want[len(want) - 2062:len(want) - 2062 + 3]
    want += b"XY"
    want += want[len(want) - 7:len(want) - 7 + 2]
    assert d == 2062 and O.codec("lzo", bytes(s), len(want) + 16) == bytes(want)


def with_checksum(frame, content, wrong=False):
    """A Zstandard frame as the encoder made it (no checksum) -> the same frame with Content_Checksum_Flag set and the
    low 32 bits of XXH64(content, seed 0) behind its last block (RFC 8878 3.1.1)."""
    import xxhash
    assert frame[:4] == b"\x28\xb5\x2f\xfd" and not frame[4] & 4
    ck = xxhash.xxh64(content, seed=0).intdigest() & 0xffffffff
    if wrong:
        ck ^= 0x00010000
    return frame[:4] + bytes([frame[4] | 4]) + frame[5:] + ck.to_bytes(4, "little")


def test_xxh64_and_the_zstandard_content_checksum():
    """libzstd (behind the reference's zstd crate, compression.rs:151-159) verifies a frame's content checksum when the header
    flags one: the oracle's XXH64 against the xxhash package, a flagged frame with the right checksum decodes, a wrong one fails."""
    import ctypes as C
    import xxhash
    L = O.lib()
    L.oo_xxh64.restype = C.c_uint64
    L.oo_xxh64.argtypes = [C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(3)
    for n in list(range(0, 70)) + [255, 256, 257, 4095, 65536, 100003]:
        b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert L.oo_xxh64(b, n) == xxhash.xxh64(b, seed=0).intdigest(), n
    for data in corpora():
        if not data:
            continue
        comp = pa.Codec("zstd").compress(data, asbytes=True)
        assert O.codec("zstd", with_checksum(comp, data), len(data) + 16) == data
        assert O.codec("zstd", with_checksum(comp, data, wrong=True), len(data) + 16) is None
        # two frames in one chunk, the second one's checksum wrong: the chunk fails
        two = with_checksum(comp, data) + with_checksum(comp, data, wrong=True)
        assert O.codec("zstd", with_checksum(comp, data) * 2, 2 * len(data) + 16) == data * 2
        assert O.codec("zstd", two, 2 * len(data) + 16) is None
