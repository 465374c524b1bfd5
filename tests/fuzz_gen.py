"""Seeded random stripes for the differential fuzz tests (GPU path vs oracle): random column types, encodings, null
densities, batch sizes and codecs; optionally one random byte of one random stream is overwritten."""
import numpy as np

from orc_rust_amd import gen

LONG, INT, SHORT, DATE, BYTE, BOOLEAN, FLOAT, DOUBLE, TIMESTAMP, STRING, BINARY, DECIMAL = 4, 3, 2, 15, 1, 0, 5, 6, 9, 7, 8, 14
VARCHAR, CHAR, TIMESTAMP_INSTANT = 16, 17, 18
PRESENT, DATA, LENGTH, DICT, SECONDARY = 0, 1, 2, 3, 5


def ints(rng, k, bits):
    """k integers of one of seven shapes (random, ramps, short repeats, sorted, outliers, long repeats, tiny range)."""
    lim = 1 << (bits - 1)
    kind = rng.integers(0, 7)
    if kind == 0:
        v = rng.integers(-lim, lim, k)
    elif kind == 1:
        v = (np.arange(k) * int(rng.integers(1, 50))) % lim
    elif kind == 2:
        v = np.repeat(rng.integers(-lim, lim, k // 5 + 1), 5)[:k]
    elif kind == 3:
        v = np.clip(np.cumsum(rng.integers(0, 300, k)), -lim, lim - 1)
    elif kind == 4:
        v = rng.integers(0, 100, k)
        idx = rng.choice(max(k, 1), max(1, k // 30), replace=False) if k else []
        if k:
            v[idx] = lim - 1
    elif kind == 5:
        v = np.repeat(rng.integers(-100, 100, k // 700 + 1), 700)[:k]
    else:
        v = rng.integers(0, 7, k)
    return np.asarray(v, dtype=np.int64)


def make_case(seed, corrupt, hits=1, wide=False, rows=None):
    """-> (n_rows, compression, block_size, batch_size, columns, streams, corrupted (column, kind) or None)
    hits > 1: that many more (stream, byte) pairs are overwritten, drawn from a second generator.
    wide: also Binary / Varchar / Char / TimestampInstant columns and decimals of other precisions and scales
    (a separate family of cases: the draws differ from the wide=False ones of the same seed)."""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 17, 511, 512, 513, 4097, 20000, 70001]))
    if rows is not None:
        n = int(rows)  # big stripes: many scan tiles / expansion groups per stream
    comp = str(rng.choice(["none", "none", "snappy", "lz4", "zlib", "zstd"]))
    block = int(rng.choice([64, 1000, 4096, 262144]))
    batch = int(rng.choice([1, 100, 1024, 8192, 10000]))
    if batch == 1 and n > 600:
        batch = 100
    c = (lambda b: gen.compress_stream(b, comp, block)) if comp != "none" else (lambda b: b)
    cols, streams = [], []
    ncols = int(rng.integers(1, 5))
    for ci in range(ncols):
        cid = ci + 1
        kinds = [LONG, INT, SHORT, DATE, BYTE, BOOLEAN, DOUBLE, FLOAT, TIMESTAMP, STRING, DECIMAL]
        if wide:
            kinds += [BINARY, BINARY, VARCHAR, CHAR, TIMESTAMP_INSTANT, DECIMAL]
        typ = int(rng.choice(kinds))
        nullf = float(rng.choice([0.0, 0.0, 0.1, 0.5, 0.95]))
        has_p = nullf > 0 or rng.random() < 0.2
        present = (rng.random(n) >= nullf).astype(np.uint8)
        k = int(present.sum()) if has_p else n
        if has_p: streams.append((cid, PRESENT, c(gen.boolean(present))))
        v2 = bool(rng.random() < 0.8)
        enc = 2 if v2 else 0
        rle = (lambda v, s: gen.rle2(v, signed=s)) if v2 else (lambda v, s: gen.rle1(v, signed=s))
        if typ in (LONG, INT, SHORT, DATE):
            bits = {LONG: 62, INT: 31, SHORT: 15, DATE: 31}[typ]
            cols.append(dict(column_id=cid, orc_type=typ, encoding=enc)); streams.append((cid, DATA, c(rle(ints(rng, k, bits), True))))
        elif typ == BYTE:
            cols.append(dict(column_id=cid, orc_type=typ, encoding=0)); streams.append((cid, DATA, c(gen.byte_rle(ints(rng, k, 8).astype(np.int8)))))
        elif typ == BOOLEAN:
            cols.append(dict(column_id=cid, orc_type=typ, encoding=0)); streams.append((cid, DATA, c(gen.boolean((rng.random(k) < rng.random()).astype(np.uint8)))))
        elif typ == DOUBLE:
            cols.append(dict(column_id=cid, orc_type=typ, encoding=0)); streams.append((cid, DATA, c(rng.standard_normal(k).view(np.uint8))))
        elif typ == FLOAT:
            cols.append(dict(column_id=cid, orc_type=typ, encoding=0)); streams.append((cid, DATA, c(rng.standard_normal(k).astype(np.float32).view(np.uint8))))
        elif typ in (TIMESTAMP, TIMESTAMP_INSTANT):
            secs = rng.integers(-2_000_000_000, 2_000_000_000, k); nanos = rng.integers(0, 1_000_000, k) * 1000
            cols.append(dict(column_id=cid, orc_type=typ, encoding=enc))
            streams.append((cid, DATA, c(rle(secs, True)))); streams.append((cid, SECONDARY, c(rle(np.where(nanos == 0, 0, (nanos // 1000 << 3) | 2), False))))
        elif typ in (STRING, BINARY, VARCHAR, CHAR):
            words = [b"AIR", b"FOB", b"MAIL", b"RAIL", b"REG AIR", b"SHIP", b"TRUCK", b"", "héllo".encode(), b"x" * 70]
            if typ != BINARY and rng.random() < 0.5:
                cols.append(dict(column_id=cid, orc_type=typ, encoding=3 if v2 else 1, dictionary_size=len(words)))
                streams += [(cid, DATA, c(rle(rng.integers(0, len(words), k), False))), (cid, LENGTH, c(rle(np.array([len(w) for w in words], dtype=np.int64), False))),
                            (cid, DICT, c(np.frombuffer(b"".join(words), dtype=np.uint8)))]
            else:
                idx = rng.integers(0, len(words), k)
                cols.append(dict(column_id=cid, orc_type=typ, encoding=enc))
                streams += [(cid, LENGTH, c(rle(np.array([len(words[i]) for i in idx], dtype=np.int64), False))), (cid, DATA, c(np.frombuffer(b"".join(words[i] for i in idx), dtype=np.uint8)))]
        else:
            prec, scale = (38, 3) if not wide else (int(rng.choice([5, 18, 19, 38])), int(rng.choice([0, 2, 3, 5])))
            cols.append(dict(column_id=cid, orc_type=typ, encoding=enc, precision=prec, scale=scale))
            streams += [(cid, DATA, c(gen.varint128([int(x) for x in rng.integers(-10**15, 10**15, k)]))), (cid, SECONDARY, c(rle(rng.integers(0, 6, k), True)))]
    if corrupt and streams:
        si = int(rng.integers(0, len(streams)))
        cid_, kind_, data_ = streams[si]
        data_ = np.array(data_, dtype=np.uint8, copy=True)
        if len(data_):
            data_[int(rng.integers(0, len(data_)))] = int(rng.integers(0, 256))
            streams[si] = (cid_, kind_, data_)
    if corrupt and streams and hits < 0:
        # hits < 0: additionally cut -hits random streams short at a random length
        rng2 = np.random.default_rng(seed ^ 0xC07)
        for _ in range(-hits):
            sj = int(rng2.integers(0, len(streams)))
            cid_, kind_, data_ = streams[sj]
            data_ = np.array(data_, dtype=np.uint8, copy=True)
            streams[sj] = (cid_, kind_, data_[:int(rng2.integers(0, len(data_) + 1))].copy())
    if corrupt and streams and hits > 1:
        rng2 = np.random.default_rng(seed ^ 0x5EED)
        for _ in range(hits - 1):
            sj = int(rng2.integers(0, len(streams)))
            cid_, kind_, data_ = streams[sj]
            data_ = np.array(data_, dtype=np.uint8, copy=True)
            if len(data_):
                data_[int(rng2.integers(0, len(data_)))] = int(rng2.integers(0, 256))
                streams[sj] = (cid_, kind_, data_)
    return n, comp, block, batch, cols, streams, ((streams[si][0], streams[si][1]) if corrupt and streams else None)
