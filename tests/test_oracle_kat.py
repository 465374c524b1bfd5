"""Pins the CPU oracle against every byte-level known-answer vector of the reference's own
unit tests (sources cited per test)."""
import numpy as np

import kat_vectors as K
import oracle_lib as O


def rle(data, expected, signed=False, version=2, nbits=64):
    st, got = O.int_rle(bytes(data), len(expected), version=version, signed=signed, nbits=nbits)
    assert st == O.OK
    assert got.tolist() == list(expected)


def test_chunk_header():  # compression.rs:354-370
    import ctypes
    o = ctypes.c_int()
    assert O.lib().oo_decode_chunk_header(bytes([0b1011, 0, 0]), ctypes.byref(o)) == 5 and o.value == 1
    assert O.lib().oo_decode_chunk_header(bytes([0b01000000, 0b00001101, 0b00000011]), ctypes.byref(o)) == 100_000 and o.value == 0


def test_rlev2_reader_test():  # rle_v2/mod.rs:587-619
    rle([2, 1, 64, 5, 80, 1, 1], [1, 1, 1, 1, 1, 0, 1, 0, 1, 0, 0, 1, 1, 1, 1])
    rle([0x5E, 0x03, 0x5C, 0xA1, 0xAB, 0x1E, 0xDE, 0xAD, 0xBE, 0xEF], [23713, 43806, 57005, 48879])
    rle([102, 9, 0, 126, 224, 7, 208, 0, 126, 79, 66, 64, 0, 127, 128, 8, 2, 0, 128, 192, 8, 22, 0, 130, 0, 8, 42],
        [2030, 2000, 2020, 1000000, 2040, 2050, 2060, 2070, 2080, 2090])
    rle([196, 9, 2, 2, 74, 40, 166], [2, 3, 5, 7, 11, 13, 17, 19, 23, 29])
    rle([0xC6, 0x09, 0x02, 0x02, 0x22, 0x42, 0x42, 0x46], [2, 3, 5, 7, 11, 13, 17, 19, 23, 29])
    rle([7, 1], [1] * 10)


def test_rlev2_short_repeat():  # rle_v2/mod.rs:622-626
    rle([0x0A, 0x27, 0x10], [10000] * 5)


def test_rlev2_direct_signed():  # rle_v2/mod.rs:636-640
    rle([110, 3, 0, 185, 66, 1, 86, 60, 1, 189, 90, 1, 125, 222], [23713, 43806, 57005, 48879], signed=True)


def test_rlev2_patched_base():  # rle_v2/mod.rs:650-659
    rle([0x8E, 0x09, 0x2B, 0x21, 0x07, 0xD0, 0x1E, 0x00, 0x14, 0x70, 0x28, 0x32, 0x3C, 0x46, 0x50, 0x5A, 0xFC, 0xE8],
        [2030, 2000, 2020, 1000000, 2040, 2050, 2060, 2070, 2080, 2090])


from kat_vectors import PATCHED_1, PATCHED_1_EXPECTED  # noqa: E402


def test_rlev2_patched_base_java():  # rle_v2/mod.rs:662-692 (Java-generated, 226 values)
    rle(PATCHED_1, PATCHED_1_EXPECTED, signed=True)
    # in batches that straddle the run
    st, got = O.int_rle(bytes(PATCHED_1), 226, signed=True, chunks=[7, 100, 119])
    assert st == O.OK and got.tolist() == PATCHED_1_EXPECTED


def test_every_shared_vector():
    """tests/kat_vectors.py: the same bytes tests/test_gpu_kat.py feeds to the HIP path."""
    for name, data, want, signed, version, nbits in K.INT_RLE:
        st, got = O.int_rle(bytes(data), len(want), version=version, signed=signed, nbits=nbits)
        assert st == O.OK and got.tolist() == list(want), name
    for name, data, want in K.BYTE_RLE:
        st, v = O.byte_rle(bytes(data), len(want))
        assert st == O.OK and (v.astype(np.int64) & 0xFF).tolist() == want, name
    for name, data, want in K.BOOLEAN:
        st, v = O.boolean(bytes(data), len(want))
        assert st == O.OK and v.tolist() == want, name
    for name, data, want in K.VARINT_I128:
        assert O.varint128(bytes(data), len(want)) == (O.OK, want), name


def test_rlev2_eof_is_out_of_spec():  # rle_v2/mod.rs:118-126
    st, got = O.int_rle(bytes([0x0A, 0x27, 0x10]), 6)
    assert st == O.OUT_OF_SPEC


def test_rlev1():  # rle_v1.rs:435-466
    rle([0x61, 0x00, 0x07], [7] * 100, version=1)
    rle([0x61, 0xFF, 0x64], list(range(100, 0, -1)), version=1)
    rle([0x7F, 0xFF, 0x96, 0x01, 0x11, 0xFF, 0x14], list(range(150, 0, -1)), version=1)
    rle([0xFB, 0x02, 0x03, 0x06, 0x07, 0x0B], [2, 3, 6, 7, 11], version=1)
    rle([0xFB, 0x02, 0x03, 0x06, 0x07, 0x0B, 0x00, 0x01, 0x01, 0xFE, 0x00, 0x80, 0x02], [2, 3, 6, 7, 11, 1, 2, 3, 0, 256], version=1)
    rle([0x01, 0x02, 0x02, 0x01, 0x02, 0x01, 0xFF, 0xFF, 0x01], [2, 4, 6, 8, 1, 3, 5, 7, 255], version=1)


def test_byte_rle():  # byte.rs:344-356, :430-434
    st, v = O.byte_rle(bytes([0x61, 0x00]), 100)
    assert st == 0 and v.tolist() == [0] * 100
    st, v = O.byte_rle(bytes([0x01, 0x01]), 4)
    assert st == 0 and v.tolist() == [1] * 4
    st, v = O.byte_rle(bytes([0xFE, 0x44, 0x45]), 2)
    assert st == 0 and v.tolist() == [0x44, 0x45]
    st, v = O.byte_rle(bytes([0x07, 0x00, 0xFD, 0x0B, 0x0C, 0x0D, 0x11, 0x05]), 33)
    assert st == 0 and v.tolist() == [0] * 10 + [11, 12, 13] + [5] * 20


def test_boolean():  # boolean.rs:177-211
    st, v = O.boolean(bytes([0x61, 0x00]), 800)
    assert st == 0 and not v.any()
    st, v = O.boolean(bytes([0xFE, 0x44, 0x45]), 16)
    assert st == 0 and v.tolist() == [0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1]
    st, v = O.boolean(bytes([0xFF, 0x80]), 8)
    assert st == 0 and v.tolist() == [1, 0, 0, 0, 0, 0, 0, 0]


def test_varint():  # integer/util.rs:770-806
    for data, want in [([0x00], 0), ([0x01], 1), ([0x7F], 127), ([0x80, 0x01], 128), ([0x81, 0x01], 129),
                       ([0xFF, 0x7F], 16383), ([0x80, 0x80, 0x01], 16384), ([0x81, 0x80, 0x01], 16385)]:
        assert O.varint(bytes(data)) == (O.OK, want)
    assert O.varint(bytes([0xFF] * 10 + [0x01]))[0] == O.VARINT_TOO_LARGE
    assert O.varint(bytes([0x80, 0x80]))[0] == O.IO_ERROR
    # narrower NInt: checked_shl fails once offset >= bit size
    assert O.varint(bytes([0x80] * 5 + [0x01]), nbits=32)[0] == O.VARINT_TOO_LARGE
    assert O.varint(bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F]), nbits=32) == (O.OK, -1)


def test_zigzag():  # integer/util.rs:623-637
    def zz(u):
        n, out = u, []
        while True:
            b = n & 0x7F
            n >>= 7
            out.append(b | (0x80 if n else 0))
            if not n:
                return bytes(out)
    for u, want in [(0, 0), (1, -1), (2, 1), (3, -2), (4, 2), (5, -3), (6, 3), (7, -4), (8, 4), (9, -5)]:
        assert O.varint(zz(u), signed=True) == (O.OK, want)
    assert O.varint(zz((1 << 64) - 2), signed=True) == (O.OK, (1 << 63) - 1)
    assert O.varint(zz((1 << 64) - 1), signed=True) == (O.OK, -(1 << 63))


def test_varint_i128():  # encoding/decimal.rs:67-128
    assert O.varint128(bytes([0x00, 0x02, 0x01, 0xC8, 0x01, 0x90, 0x03]), 5) == (O.OK, [0, 1, -1, 100, 200])
    assert O.varint128(bytes([0x14, 0x28, 0x3C, 0x50, 0x64]), 5) == (O.OK, [10, 20, 30, 40, 50])
    assert O.varint128(bytes([0x14, 0x28]), 3)[0] == O.IO_ERROR


def test_delta_semantics():  # delta.rs:195-310
    # fixed +10 x100 from 0 / fixed -63 x150 from 10000 (header built by hand: DELTA, width code 0)
    def delta_fixed(base, delta, n, signed):
        def uv(u):
            out = []
            while True:
                b = u & 0x7F
                u >>= 7
                out.append(b | (0x80 if u else 0))
                if not u:
                    return out
        zz = lambda v: (v << 1) ^ (v >> 63) if v >= 0 else ((-v) << 1) - 1
        hdr = [0xC0 | (((n - 1) >> 8) & 1), (n - 1) & 0xFF]
        return bytes(hdr + uv(zz(base) if signed else base) + uv(zz(delta)))
    rle(delta_fixed(0, 10, 100, True), [10 * i for i in range(100)], signed=True)
    rle(delta_fixed(10000, -63, 150, True), [10000 - 63 * i for i in range(150)], signed=True)
    # delta_base == 0 subtracts |0| (delta.rs:77-82): a fixed run of zeros delta is just repeats
    rle(delta_fixed(5, 0, 4, False), [5, 5, 5, 5])
    # overflow is an error, not a wrap (i64 and narrower NInt, delta.rs:287-310)
    st, _ = O.int_rle(delta_fixed((1 << 63) - 2, 1, 5, True), 5, signed=True)
    assert st == O.OUT_OF_SPEC
    st, got = O.int_rle(delta_fixed(-(1 << 31), (1 << 31) - 1, 3, True), 3, signed=True, nbits=32)
    assert st == O.OK and got.tolist() == [-(1 << 31), -1, 2147483646]
    st, _ = O.int_rle(delta_fixed(2147483646, 1, 3, True), 3, signed=True, nbits=32)
    assert st == O.OUT_OF_SPEC


def test_timestamp_rules():  # encoding/timestamp.rs:121-192
    base = 1_420_070_400
    assert O.decode_timestamp(base, 0, 0) == (O.OK, base * 10**9)
    assert O.decode_timestamp(base, 1, (123 << 3) | 5) == (O.OK, (base + 1) * 10**9 + 123 * 10**6)
    # ORC-763: negative seconds with nanos > 999_999 subtract one second
    assert O.decode_timestamp(base, -base - 5, (2_000_000 << 3)) == (O.OK, -6 * 10**9 + 2_000_000)
    assert O.decode_timestamp(base, -base - 5, (999_999 << 3)) == (O.OK, -5 * 10**9 + 999_999)
    # loss of precision / overflow are errors
    assert O.decode_timestamp(base, 0, (1 << 3), unit=2)[0] == O.DECODE_TIMESTAMP
    assert O.decode_timestamp(base, 1 << 40, 0)[0] == O.DECODE_TIMESTAMP
    assert O.decode_timestamp(base, 1 << 40, 0, unit=0) == (O.OK, base + (1 << 40))


def test_decimal_types_arrow_refuses():  # array_decoder/decimal.rs:96-100 (with_precision_and_scale), :59-60 (`as u8` / `as i8`)
    """Every batch of a Decimal column is given its precision and scale by arrow-rs, which refuses a precision outside 1..=38, a scale
    above 38 and a positive scale above the precision: an ArrowError of the batch, behind whatever its streams fail at."""
    import numpy as np
    from orc_rust_amd import gen
    vals = [1, -2, 300]
    DATA, SECONDARY = 1, 5  # Stream.Kind
    streams = {DATA: bytes(gen.varint128(vals)), SECONDARY: bytes(gen.rle2(np.full(3, 2, dtype=np.int64), signed=True))}
    for precision, scale, want in ((10, 2, O.OK), (38, 38, O.OK), (1, 0, O.OK), (0, 0, O.ARROW), (39, 2, O.ARROW), (5, 6, O.ARROW), (38, 39, O.ARROW),
                                   (256 + 10, 2, O.OK), (256, 0, O.ARROW), (10, 256 + 2, O.OK), (10, 255, O.OK)):  # (255 as i8 = -1: negative scales pass)
        c = O.Column(14, 2, streams, precision=precision, scale=scale)
        assert c.next_batch(3)["status"] == want, (precision, scale)
    # a stream that fails in the batch comes first
    c = O.Column(14, 2, {DATA: bytes(gen.varint128([10**20]))[:1], SECONDARY: streams[SECONDARY]}, precision=0, scale=0)
    assert c.next_batch(3)["status"] == O.IO_ERROR


def test_decimal_scale_repair():  # array_decoder/decimal.rs:138-166
    assert O.fix_scale(12345, 2, 2) == 12345
    assert O.fix_scale(12345, 2, 4) == 123
    assert O.fix_scale(-12345, 2, 4) == -123  # truncation toward zero
    assert O.fix_scale(12345, 5, 2) == 12345000
    assert O.fix_scale(-1, 38, 0) == -(10**38)  # fits i128
