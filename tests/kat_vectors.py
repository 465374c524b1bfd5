"""Byte-level known-answer vectors transcribed from the reference's own unit tests (the source
of each vector is cited next to it).  Data only: shared by tests/test_oracle_kat.py (pins the CPU oracle) and
tests/test_gpu_kat.py (feeds the same bytes through the C ABI to the HIP path)."""

PATCHED_1 = [
    144, 109, 4, 164, 141, 16, 131, 194, 0, 240, 112, 64, 60, 84, 24, 3, 193, 201, 128, 120, 60, 33, 4, 244, 3, 193, 192, 224,
    128, 56, 32, 15, 22, 131, 129, 225, 0, 112, 84, 86, 14, 8, 106, 193, 192, 228, 160, 64, 32, 14, 213, 131, 193, 192, 240, 121,
    124, 30, 18, 9, 132, 67, 0, 224, 120, 60, 28, 14, 32, 132, 65, 192, 240, 160, 56, 61, 91, 7, 3, 193, 192, 240, 120, 76, 29,
    23, 7, 3, 220, 192, 240, 152, 60, 52, 15, 7, 131, 129, 225, 0, 144, 56, 30, 14, 44, 140, 129, 194, 224, 120, 0, 28, 15, 8,
    6, 129, 198, 144, 128, 104, 36, 27, 11, 38, 131, 33, 48, 224, 152, 60, 111, 6, 183, 3, 112, 0, 1, 78, 5, 46, 2, 1, 1, 141,
    3, 1, 1, 138, 22, 0, 65, 1, 4, 0, 225, 16, 209, 192, 4, 16, 8, 36, 16, 3, 48, 1, 3, 13, 33, 0, 176, 0, 1, 94, 18, 0, 68, 0,
    33, 1, 143, 0, 1, 7, 93, 0, 25, 0, 5, 0, 2, 0, 4, 0, 1, 0, 1, 0, 2, 0, 16, 0, 1, 11, 150, 0, 3, 0, 1, 0, 1, 99, 157, 0, 1,
    140, 54, 0, 162, 1, 130, 0, 16, 112, 67, 66, 0, 2, 4, 0, 0, 224, 0, 1, 0, 16, 64, 16, 91, 198, 1, 2, 0, 32, 144, 64, 0, 12,
    2, 8, 24, 0, 64, 0, 1, 0, 0, 8, 48, 51, 128, 0, 2, 12, 16, 32, 32, 71, 128, 19, 76,
]
PATCHED_1_EXPECTED = [
    20, 2, 3, 2, 1, 3, 17, 71, 35, 2, 1, 139, 2, 2, 3, 1783, 475, 2, 1, 1, 3, 1, 3, 2, 32, 1, 2, 3, 1, 8, 30, 1, 3, 414, 1, 1,
    135, 3, 3, 1, 414, 2, 1, 2, 2, 594, 2, 5, 6, 4, 11, 1, 2, 2, 1, 1, 52, 4, 1, 2, 7, 1, 17, 334, 1, 2, 1, 2, 2, 6, 1, 266, 1,
    2, 217, 2, 6, 2, 13, 2, 2, 1, 2, 3, 5, 1, 2, 1, 7244, 11813, 1, 33, 2, -13, 1, 2, 3, 13, 1, 92, 3, 13, 5, 14, 9, 141, 12, 6,
    15, 25, -1, -1, -1, 23, 1, -1, -1, -71, -2, -1, -1, -1, -1, 2, 1, 4, 34, 5, 78, 8, 1, 2, 2, 1, 9, 10, 2, 1, 4, 13, 1, 5, 4,
    4, 19, 5, -1, -1, -1, 34, -17, -200, -1, -943, -13, -3, 1, 2, -1, -1, 1, 8, -1, 1483, -2, -1, -1, -12751, -1, -1, -1, 66, 1,
    3, 8, 131, 14, 5, 1, 2, 2, 1, 1, 8, 1, 1, 2, 1, 5, 9, 2, 3, 112, 13, 2, 2, 1, 5, 10, 3, 1, 1, 13, 2, 3, 4, 1, 3, 1, 1, 2, 1,
    1, 2, 4, 2, 207, 1, 1, 2, 4, 3, 3, 2, 2, 16,
]

# (name, stream bytes, expected values, signed, rle version, NInt bits)
INT_RLE = [
    ("rle_v2/mod.rs:589 mixed SR/DIRECT/SR", [2, 1, 64, 5, 80, 1, 1], [1, 1, 1, 1, 1, 0, 1, 0, 1, 0, 0, 1, 1, 1, 1], False, 2, 64),
    ("rle_v2/mod.rs:629 direct u16", [0x5E, 0x03, 0x5C, 0xA1, 0xAB, 0x1E, 0xDE, 0xAD, 0xBE, 0xEF], [23713, 43806, 57005, 48879], False, 2, 64),
    ("rle_v2/mod.rs:598 patched base wide", [102, 9, 0, 126, 224, 7, 208, 0, 126, 79, 66, 64, 0, 127, 128, 8, 2, 0, 128, 192, 8, 22, 0, 130, 0, 8, 42],
     [2030, 2000, 2020, 1000000, 2040, 2050, 2060, 2070, 2080, 2090], False, 2, 64),
    ("rle_v2/mod.rs:608 delta alt packing", [196, 9, 2, 2, 74, 40, 166], [2, 3, 5, 7, 11, 13, 17, 19, 23, 29], False, 2, 64),
    ("rle_v2/mod.rs:643 delta primes", [0xC6, 0x09, 0x02, 0x02, 0x22, 0x42, 0x42, 0x46], [2, 3, 5, 7, 11, 13, 17, 19, 23, 29], False, 2, 64),
    ("rle_v2/mod.rs:616 short repeat tiny", [7, 1], [1] * 10, False, 2, 64),
    ("rle_v2/mod.rs:622 short repeat", [0x0A, 0x27, 0x10], [10000] * 5, False, 2, 64),
    ("rle_v2/mod.rs:636 direct signed", [110, 3, 0, 185, 66, 1, 86, 60, 1, 189, 90, 1, 125, 222], [23713, 43806, 57005, 48879], True, 2, 64),
    ("rle_v2/mod.rs:650 patched base", [0x8E, 0x09, 0x2B, 0x21, 0x07, 0xD0, 0x1E, 0x00, 0x14, 0x70, 0x28, 0x32, 0x3C, 0x46, 0x50, 0x5A, 0xFC, 0xE8],
     [2030, 2000, 2020, 1000000, 2040, 2050, 2060, 2070, 2080, 2090], False, 2, 64),
    ("rle_v2/mod.rs:662 patched base, Java generated", PATCHED_1, PATCHED_1_EXPECTED, True, 2, 64),
    # crafted: a VARYING delta run whose first delta is zero.  delta.rs:77-82 picks "subtract" for
    # delta_base <= 0, so the packed deltas 1 2 3 4 are subtracted (Apache ORC's Java / C++ readers would add them).
    ("delta.rs:77-82 varying delta, zero first delta", [0xC6, 0x05, 0x64, 0x00, 0x12, 0x34], [100, 100, 99, 97, 94, 90], False, 2, 64),
    ("delta.rs:77-82 varying delta, zero first delta, signed", [0xC6, 0x05, 0xC8, 0x01, 0x00, 0x12, 0x34], [100, 100, 99, 97, 94, 90], True, 2, 64),
    ("rle_v1.rs:435 run", [0x61, 0x00, 0x07], [7] * 100, False, 1, 64),
    ("rle_v1.rs:439 descending run", [0x61, 0xFF, 0x64], list(range(100, 0, -1)), False, 1, 64),
    ("rle_v1.rs:443 two runs", [0x7F, 0xFF, 0x96, 0x01, 0x11, 0xFF, 0x14], list(range(150, 0, -1)), False, 1, 64),
    ("rle_v1.rs:455 literals", [0xFB, 0x02, 0x03, 0x06, 0x07, 0x0B], [2, 3, 6, 7, 11], False, 1, 64),
    ("rle_v1.rs:459 literals + run + literals", [0xFB, 0x02, 0x03, 0x06, 0x07, 0x0B, 0x00, 0x01, 0x01, 0xFE, 0x00, 0x80, 0x02],
     [2, 3, 6, 7, 11, 1, 2, 3, 0, 256], False, 1, 64),
    ("rle_v1.rs:448 mixed", [0x01, 0x02, 0x02, 0x01, 0x02, 0x01, 0xFF, 0xFF, 0x01], [2, 4, 6, 8, 1, 3, 5, 7, 255], False, 1, 64),
]

# (name, stream bytes, expected byte values)
BYTE_RLE = [
    ("byte.rs:344 run", [0x61, 0x00], [0] * 100),
    ("byte.rs:348 short run", [0x01, 0x01], [1] * 4),
    ("byte.rs:352 literals", [0xFE, 0x44, 0x45], [0x44, 0x45]),
    ("byte.rs:430 run literals run", [0x07, 0x00, 0xFD, 0x0B, 0x0C, 0x0D, 0x11, 0x05], [0] * 10 + [11, 12, 13] + [5] * 20),
]

# (name, stream bytes, expected bools)
BOOLEAN = [
    ("boolean.rs:177 run of zero bytes", [0x61, 0x00], [0] * 800),
    ("boolean.rs:190 literals", [0xFE, 0x44, 0x45], [0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1]),
    ("boolean.rs:203 one byte", [0xFF, 0x80], [1, 0, 0, 0, 0, 0, 0, 0]),
]

# (name, stream bytes, expected i128 values)
VARINT_I128 = [
    ("encoding/decimal.rs:67", [0x00, 0x02, 0x01, 0xC8, 0x01, 0x90, 0x03], [0, 1, -1, 100, 200]),
    ("encoding/decimal.rs:111", [0x14, 0x28, 0x3C, 0x50, 0x64], [10, 20, 30, 40, 50]),
]
