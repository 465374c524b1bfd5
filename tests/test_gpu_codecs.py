"""GPU parity, block decompressors on their own: DOUBLE columns carry arbitrary bytes (the DATA
stream is raw IEEE-754, float.rs:53-75), so every shape of compressed block can be pushed through
the chunk decoders and compared with the oracle byte for byte.  The blocks come from REAL encoders
(pyarrow's Snappy / LZ4-raw / Zstandard, Python's zlib) and from a hand-rolled Snappy writer that
forces the element forms real encoders rarely emit (4-byte offsets, 1..4 extra length bytes,
overlapping copies of every small distance)."""
import zlib

import numpy as np
import pyarrow as pa
import pytest

import gpu_util as G

pytestmark = pytest.mark.gpu

DOUBLE, DATA = 6, 1


def frame(raw, compress, block):
    """ORC chunk framing (compression.rs:113-123): original chunk when compression does not pay."""
    parts = []
    for p in range(0, len(raw), block):
        blk = raw[p:p + block]
        comp = compress(blk)
        if comp is not None and len(comp) < len(blk):
            h = len(comp) << 1
            parts.append(bytes([h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF]) + comp)
        else:
            h = (len(blk) << 1) | 1
            parts.append(bytes([h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF]) + blk)
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


def shapes(seed):
    rng = np.random.default_rng(seed)
    rnd = lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    words = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK", b"DELIVER IN PERSON", b"NONE", b"TAKE BACK RETURN"]
    text = b" ".join(words[i] for i in rng.integers(0, len(words), 60000))
    far = rnd(30000)
    far2 = rnd(100000)
    out = {
        "zeros": bytes(600000),                                             # distance-1 overlapping copies, very long matches
        "period7": (b"abcdefg" * 90000)[:600000],                           # distance < length
        "period3": (b"xyz" * 200000)[:600000],
        "random": rnd(300000),                                              # literal only (original chunks)
        "random+zeros": rnd(70000) + bytes(50000) + rnd(3000) + bytes(200000),  # long literals inside compressed chunks
        "text": text,                                                       # short literals and copies, small distances
        "bitmap": (rng.random(400000) < 0.43).astype(np.uint8).tobytes().replace(b"\x01", b"\xff"),  # token-heavy
        "far30k": far + rnd(500) + far + rnd(700) + far,                    # distances of ~30 KB (behind a 32 KiB ring's near zone)
        "far100k": far2 + rnd(100) + far2,                                  # distances of 100 KB (Zstandard windows)
        "short": b"orc",                                                    # smaller than any header
    }
    return {k: v + bytes((-len(v)) % 8) for k, v in out.items()}


CODECS = {
    "snappy": lambda b: pa.Codec("snappy").compress(b, asbytes=True),
    "lz4": lambda b: pa.Codec("lz4_raw").compress(b, asbytes=True),
    "zstd": lambda b: pa.Codec("zstd", compression_level=3).compress(b, asbytes=True),
    "zlib": lambda b: (lambda c: c.compress(b) + c.flush())(zlib.compressobj(6, zlib.DEFLATED, -15)),
}


@pytest.mark.parametrize("kind", ["snappy", "lz4", "zlib", "zstd"])
@pytest.mark.parametrize("block", [262144, 65536, 1000])
def test_real_encoders(kind, block):
    cols, streams, names = [], [], []
    for name, raw in shapes(block).items():
        if len(raw) == 0:
            continue
        cid = len(cols) + 1
        cols.append({"column_id": cid, "orc_type": DOUBLE, "encoding": 0})
        streams.append((cid, DATA, frame(raw, CODECS[kind], block)))
        names.append((name, len(raw) // 8))
    # the stripe has as many rows as its shortest column needs; every column is compared over its own length
    for ci, c in enumerate(cols):
        n = names[ci][1]
        res = G.gpu_decode(n, [c], [streams[ci]], compression=kind, block_size=block)
        assert res.status()[0] == 0, (kind, block, names[ci], res.status())
        G.assert_column_parity(res, 0, c, [streams[ci]], n, 8192, compression=kind, block_size=block, what=(kind, block, names[ci]))
        res.free()


# ---- hand-rolled Snappy: the element forms encoders rarely produce ---------------------------------

def sn_varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def sn_literal(data, force_ext=0):
    n = len(data) - 1
    if force_ext == 0 and n < 60:
        return bytes([n << 2]) + data
    nb = max(force_ext, 1 if n < 256 else 2 if n < 65536 else 3 if n < (1 << 24) else 4)
    return bytes([(59 + nb) << 2]) + n.to_bytes(nb, "little") + data


def sn_copy(off, ln, form):
    if form == 1:
        assert 4 <= ln <= 11 and off < 2048
        return bytes([1 | ((ln - 4) << 2) | ((off >> 8) << 5), off & 0xFF])
    if form == 2:
        assert 1 <= ln <= 64 and off < 65536
        return bytes([2 | ((ln - 1) << 2)]) + off.to_bytes(2, "little")
    assert 1 <= ln <= 64
    return bytes([3 | ((ln - 1) << 2)]) + off.to_bytes(4, "little")


def build_snappy(seed, total):
    """(plain, block): random element soup; every copy is applied to a Python model of the output."""
    rng = np.random.default_rng(seed)
    plain = bytearray()
    body = bytearray()
    while len(plain) < total:
        r = rng.random()
        if r < 0.30 or len(plain) < 16:
            ln = int(rng.choice([1, 2, 5, 59, 60, 61, 64, 65, 200, 255, 256, 257, 5000, 70000]))
            data = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
            ext = int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
            if ext and ln - 1 >= (1 << (8 * ext)):
                ext = 0
            body += sn_literal(data, ext)
            plain += data
            continue
        form = int(rng.choice([1, 2, 2, 3]))
        ln = int(rng.integers(4, 12)) if form == 1 else int(rng.integers(1, 65))
        maxoff = min(len(plain), 2047 if form == 1 else 65535 if form == 2 else len(plain))
        pick = rng.random()
        if pick < 0.35:
            off = int(rng.integers(1, min(maxoff, 8) + 1))        # overlapping, tiny distances
        elif pick < 0.7:
            off = int(rng.integers(1, min(maxoff, 300) + 1))       # inside the current group of elements
        else:
            off = int(rng.integers(1, maxoff + 1))                  # anywhere, far ones included (form 3: > 64 KiB)
        body += sn_copy(off, ln, form)
        start = len(plain) - off
        for k in range(ln):
            plain.append(plain[start + k])
    plain = bytes(plain)
    pad = (-len(plain)) % 8
    if pad:
        body += sn_literal(bytes(pad))
        plain += bytes(pad)
    return plain, sn_varint(len(plain)) + bytes(body)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_handmade_snappy_elements(seed):
    plain, block = build_snappy(seed, 250000)
    assert len(block) < (1 << 23)
    h = len(block) << 1
    stream = np.frombuffer(bytes([h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF]) + block, dtype=np.uint8).copy()
    n = len(plain) // 8
    bs = 1 << 19  # the chunk's plain size must fit the declared compression block size
    assert len(plain) <= bs
    c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
    res = G.gpu_decode(n, [c], [(1, DATA, stream)], compression="snappy", block_size=bs)
    assert res.status()[0] == 0, res.status()
    got = b"".join(bytes(res.batch(b, 0)["values"]) for b in range(res.n_batches))
    assert got == plain
    G.assert_column_parity(res, 0, c, [(1, DATA, stream)], n, 8192, compression="snappy", block_size=bs, what=("handmade", seed))


@pytest.mark.parametrize("case", ["offset0", "offset_too_far", "literal_overrun", "length_mismatch", "truncated_tag"])
def test_malformed_snappy_is_rejected_like_the_oracle(case):
    data = bytes(range(64)) * 8
    body = sn_literal(data[:40])
    if case == "offset0":
        body += sn_copy(0, 8, 2)
    elif case == "offset_too_far":
        body += sn_copy(41, 8, 2)
    elif case == "literal_overrun":
        body += bytes([(59 + 2) << 2]) + (5000).to_bytes(2, "little") + b"abc"
    elif case == "length_mismatch":
        body += sn_literal(data[:24])
    else:
        body += bytes([2 | (7 << 2), 5])  # 2-byte-offset copy with one offset byte missing
    block = sn_varint(512) + body
    h = len(block) << 1
    stream = np.frombuffer(bytes([h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF]) + block, dtype=np.uint8).copy()
    c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
    res = G.gpu_decode(64, [c], [(1, DATA, stream)], compression="snappy", block_size=4096)
    assert res.status()[0] != 0
    G.assert_column_parity(res, 0, c, [(1, DATA, stream)], 64, 8192, compression="snappy", block_size=4096, what=case)


# ---- hand-made LZ4 blocks: long length extensions, tiny offsets, the forms of the block's end -------------------------
def lz4_seq(lit, off=None, mlen=0):
    """One LZ4 sequence: literals, then (off, mlen >= 4) unless it is the block's last one (off None)."""
    ll = len(lit)
    ml = 0 if off is None else mlen - 4
    out = bytearray([(min(ll, 15) << 4) | min(ml, 15)])
    if ll >= 15:
        r = ll - 15
        out += b"\xff" * (r // 255) + bytes([r % 255])
    out += lit
    if off is not None:
        out += off.to_bytes(2, "little")
        if ml >= 15:
            r = ml - 15
            out += b"\xff" * (r // 255) + bytes([r % 255])
    return bytes(out)


def build_lz4(seed, total):
    rng = np.random.default_rng(seed)
    plain, body = bytearray(), bytearray()
    while len(plain) < total:
        kind = int(rng.integers(0, 6))
        ll = [0, 3, 14, 15, 270, 70000][kind] if rng.random() < 0.5 else int(rng.integers(0, 40))
        lit = rng.integers(0, 256, ll, dtype=np.uint8).tobytes()
        plain += lit
        if not plain:
            plain += b"x"
            lit = b"x"
        off = int(rng.choice([1, 2, 3, 7, 64, 1000, 65535]))
        off = max(1, min(off, len(plain)))
        mlen = int(rng.choice([4, 5, 18, 19, 20, 274, 600, 9000]))
        body += lz4_seq(lit, off, mlen)
        start = len(plain) - off
        for k in range(mlen):
            plain.append(plain[start + k])
    tail = rng.integers(0, 256, int(rng.integers(0, 30)), dtype=np.uint8).tobytes()
    pad = (-(len(plain) + len(tail))) % 8
    tail += bytes(pad)
    plain += tail
    body += lz4_seq(tail)
    return bytes(plain), bytes(body)


def lz4_stream(block):
    h = len(block) << 1
    return np.frombuffer(bytes([h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF]) + block, dtype=np.uint8).copy()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_handmade_lz4_sequences(seed):
    plain, block = build_lz4(seed, 200000)
    assert len(block) < len(plain) and len(block) < (1 << 23)
    bs = 1 << 20
    assert len(plain) <= bs
    n = len(plain) // 8
    c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
    stream = lz4_stream(block)
    res = G.gpu_decode(n, [c], [(1, DATA, stream)], compression="lz4", block_size=bs)
    assert res.status()[0] == 0, res.status()
    got = b"".join(bytes(res.batch(b, 0)["values"]) for b in range(res.n_batches))
    assert got == plain
    G.assert_column_parity(res, 0, c, [(1, DATA, stream)], n, 8192, compression="lz4", block_size=bs, what=("lz4 handmade", seed))


@pytest.mark.parametrize("case", ["offset0", "offset_too_far", "ends_with_match", "literal_overrun", "truncated_extension", "output_past_block", "empty"])
def test_malformed_lz4_is_rejected_like_the_oracle(case):
    data = bytes(range(64)) * 8
    if case == "offset0":
        block = lz4_seq(data[:40], 0, 8) + lz4_seq(data[:16])
    elif case == "offset_too_far":
        block = lz4_seq(data[:40], 41, 8) + lz4_seq(data[:16])
    elif case == "ends_with_match":
        block = lz4_seq(data[:40], 8, 24)
    elif case == "literal_overrun":
        block = bytes([0xF0, 200]) + data[:20]
    elif case == "truncated_extension":
        block = lz4_seq(data[:40], 8, 4) + bytes([0x4F]) + data[:4] + (8).to_bytes(2, "little") + b"\xff"  # match length extension cut off
    elif case == "output_past_block":
        block = lz4_seq(data[:40], 1, 9000) + lz4_seq(data[:8])
    else:
        block = b""
    c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
    stream = lz4_stream(block) if block else np.array([0, 0, 0], dtype=np.uint8)
    res = G.gpu_decode(64, [c], [(1, DATA, stream)], compression="lz4", block_size=4096)
    assert res.status()[0] != 0
    G.assert_column_parity(res, 0, c, [(1, DATA, stream)], 64, 8192, compression="lz4", block_size=4096, what=case)


# ---- LZO1X (compression.rs:174-183) -------------------------------------------------------------------------
def lzo_shapes(seed):
    rng = np.random.default_rng(seed)
    rnd = lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    words = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK", b"DELIVER IN PERSON", b"NONE", b"TAKE BACK RETURN"]
    far = rnd(20000)
    far2 = rnd(40000)
    out = {
        "zeros": bytes(200000),                                                  # distance-1 overlapping copies, very long matches
        "period7": (b"abcdefg" * 30000)[:200000],
        "random": rnd(100000),                                                   # literal runs only
        "random+zeros": rnd(30000) + bytes(20000) + rnd(3000) + bytes(70000),    # long literal runs between matches
        "text": b" ".join(words[i] for i in rng.integers(0, len(words), 20000)),  # short literals and copies, small distances (M2)
        "far20k": far + rnd(500) + far + rnd(700) + far,                         # M4 distances
        "far40k": far2 + rnd(100) + far2,                                        # ... near the 48 KiB limit
        "short": b"orc",
    }
    return {k: v + bytes((-len(v)) % 8) for k, v in out.items()}


@pytest.mark.parametrize("block", [262144, 65536, 1000])
@pytest.mark.parametrize("m2", [True, False])
def test_lzo_streams_of_the_test_encoder(block, m2):
    import lzo_enc
    for name, raw in lzo_shapes(block).items():
        c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
        stream = frame(raw, lambda b: lzo_enc.compress(b, use_m2=m2), block)
        n = len(raw) // 8
        res = G.gpu_decode(n, [c], [(1, DATA, stream)], compression="lzo", block_size=block)
        assert res.status()[0] == 0, (name, block, res.status())
        got = b"".join(bytes(res.batch(b, 0)["values"]) for b in range(res.n_batches))
        assert got == raw, (name, block)
        G.assert_column_parity(res, 0, c, [(1, DATA, stream)], n, 8192, compression="lzo", block_size=block, what=("lzo", name, block))
        res.free()


def lzo_chunk(block):
    h = len(block) << 1
    return np.frombuffer(bytes([h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF]) + block, dtype=np.uint8).copy()


def test_lzo_state_dependent_short_matches():
    import lzo_enc
    rng = np.random.default_rng(5)
    lits = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
    s = bytearray(lzo_enc._literal_run(lits))
    s += bytes([(1 << 2) | 2, 3]) + b"XY"      # after a literal run: 3 bytes from 1 + 12 + 2049 back, then 2 literals
    s += bytes([(2 << 2) | 1, 1]) + b"Z"       # after 2 literals: 2 bytes from 2 + 4 + 1 back, then 1 literal
    s += bytes([(0 << 2) | 0, 0])              # after 1 literal: 2 bytes from 1 back
    s += b"\x11\x00\x00"
    want = bytearray(lits)
    want += want[len(want) - 2062:len(want) - 2062 + 3] + b"XY"
    want += want[len(want) - 7:len(want) - 7 + 2] + b"Z"
    want += want[len(want) - 1:len(want)] * 2
    want += bytes((-len(want)) % 8)
    # (pad the plain text to whole doubles with a second chunk of zeros)
    pad = len(want) - (len(lits) + 3 + 2 + 2 + 1 + 2)
    stream = np.concatenate([lzo_chunk(bytes(s)), lzo_chunk(lzo_enc.compress(bytes(pad)))]) if pad else lzo_chunk(bytes(s))
    c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
    n = len(want) // 8
    res = G.gpu_decode(n, [c], [(1, DATA, stream)], compression="lzo", block_size=8192)
    assert res.status()[0] == 0, res.status()
    assert b"".join(bytes(res.batch(b, 0)["values"]) for b in range(res.n_batches)) == bytes(want)
    G.assert_column_parity(res, 0, c, [(1, DATA, stream)], n, 8192, compression="lzo", block_size=8192, what="lzo short matches")


@pytest.mark.parametrize("case", ["no_end_marker", "input_left_over", "lookbehind", "truncated_length", "too_short", "cut_literals",
                                  "flip0", "flip1", "flip2", "flip3", "flip4", "flip5"])
def test_malformed_lzo_is_rejected_like_the_oracle(case):
    import lzo_enc
    data = (bytes(range(64)) * 40 + b"tail of the block ....")[:2560]
    good = lzo_enc.compress(data)
    if case == "no_end_marker":
        block = good[:-3]
    elif case == "input_left_over":
        block = good + b"\x00\x00"
    elif case == "lookbehind":
        block = b"\x16abcde" + b"\x20\x40\x00" + b"\x11\x00\x00"
    elif case == "truncated_length":
        block = b"\x16abcde" + b"\x20\x00\x00\x00"
    elif case == "too_short":
        block = b"\x11\x00"
    elif case == "cut_literals":
        block = good[:len(good) // 2]
    else:
        rng = np.random.default_rng(int(case[4:]))
        b = bytearray(good)
        for _ in range(3):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        block = bytes(b)
    c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
    stream = lzo_chunk(block)
    res = G.gpu_decode(320, [c], [(1, DATA, stream)], compression="lzo", block_size=4096)
    # (a flipped bit may leave a stream that still decodes: then the bytes must agree; else the error kind)
    G.assert_column_parity(res, 0, c, [(1, DATA, stream)], 320, 8192, compression="lzo", block_size=4096, what=case)
    if not case.startswith("flip"):
        assert res.status()[0] != 0


@pytest.mark.parametrize("kind", ["zlib", "lzo"])
def test_a_chunk_may_expand_past_the_compression_block_size(kind):
    """flate2 and lzokay grow their output (compression.rs:142-150, :174-183): a chunk that expands to more than the file's
    compression block size decodes -- no conforming writer emits one, but the reference reads it.  Here such a chunk first
    fails in its block-sized slot, gets the head-room the oracle gives the crates (max(block size, 4 MiB)) and the call
    runs again: same bytes as the oracle; beyond that head-room both reject it."""
    import lzo_enc
    comp = CODECS["zlib"] if kind == "zlib" else lzo_enc.compress
    raw = bytes(range(256)) * 64 + bytes(40000)  # 56 384 bytes in ONE chunk of a file whose block size says 4096
    block = comp(raw)
    c = {"column_id": 1, "orc_type": DOUBLE, "encoding": 0}
    stream = lzo_chunk(block)
    n = len(raw) // 8
    res = G.gpu_decode(n, [c], [(1, DATA, stream)], compression=kind, block_size=4096)
    assert res.status()[0] == 0, res.status()
    assert b"".join(bytes(res.batch(b, 0)["values"]) for b in range(res.n_batches)) == raw
    G.assert_column_parity(res, 0, c, [(1, DATA, stream)], n, 8192, compression=kind, block_size=4096, what=("oversize", kind))
    res.free()
    # ... next to ordinary chunks of the same stream, and a second column that needs no second run
    raw2 = bytes(range(200)) * 40
    stream2 = np.concatenate([frame(raw2, comp, 4096), lzo_chunk(block), frame(raw2, comp, 4096)])
    n2 = (2 * len(raw2) + len(raw)) // 8
    cols = [c, {"column_id": 2, "orc_type": DOUBLE, "encoding": 0}]
    streams = [(1, DATA, stream2), (2, DATA, frame(bytes(n2 * 8), comp, 4096))]
    res = G.gpu_decode(n2, cols, streams, compression=kind, block_size=4096)
    assert res.status()[0] == 0, res.status()
    for ci, cc in enumerate(cols):
        G.assert_column_parity(res, ci, cc, streams, n2, 8192, compression=kind, block_size=4096, what=("oversize in a stream", kind, ci))
    res.free()
    # more than 4 MiB out of one chunk: rejected by both
    big = comp(bytes(5 << 20))
    stream3 = lzo_chunk(big)
    res = G.gpu_decode(1000, [c], [(1, DATA, stream3)], compression=kind, block_size=4096)
    assert res.status()[0] == 9
    G.assert_column_parity(res, 0, c, [(1, DATA, stream3)], 1000, 8192, compression=kind, block_size=4096, what=("beyond the head-room", kind))
    res.free()
