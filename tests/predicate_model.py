"""TEST INFRASTRUCTURE: a protobuf encoder for the index messages of the ORC format (RowIndex, ColumnStatistics,
BloomFilterIndex: format/orc_proto.proto) and plain-Python restatements of the Bloom filter hashing of
src/bloom_filter.rs (hash_long :127-141, murmur3_64_orc :164-215, add_hash / test_hash :76-125), used to BUILD test inputs."""
import struct

M64 = (1 << 64) - 1


def varint(v):
    out = bytearray()
    v &= M64
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def zigzag(v):
    return ((v << 1) ^ (v >> 63)) & M64


def f_varint(num, v):
    return varint(num << 3) + varint(v)


def f_sint(num, v):
    return f_varint(num, zigzag(v))


def f_bytes(num, b):
    return varint((num << 3) | 2) + varint(len(b)) + bytes(b)


def f_double(num, v):
    return varint((num << 3) | 1) + struct.pack("<d", v)


def column_statistics(number_of_values=None, has_null=None, integer=None, double=None, string=None, bucket=None, decimal=None, date=None,
                      timestamp_utc=None, binary_sum=None):
    """integer / double / date: (min, max); string: dict(minimum=, maximum=, lower_bound=, upper_bound=); bucket: true count;
    decimal: (min, max) strings; timestamp_utc: (minimumUtc, maximumUtc)."""
    out = b""
    if number_of_values is not None:
        out += f_varint(1, number_of_values)
    if integer is not None:
        out += f_bytes(2, f_sint(1, integer[0]) + f_sint(2, integer[1]) + f_sint(3, 0))
    if double is not None:
        out += f_bytes(3, f_double(1, double[0]) + f_double(2, double[1]))
    if string is not None:
        s = b""
        for num, key in ((1, "minimum"), (2, "maximum"), (4, "lower_bound"), (5, "upper_bound")):
            if key in string:
                s += f_bytes(num, string[key].encode())
        out += f_bytes(4, s)
    if bucket is not None:
        out += f_bytes(5, f_bytes(1, varint(bucket)))
    if decimal is not None:
        out += f_bytes(6, f_bytes(1, decimal[0].encode()) + f_bytes(2, decimal[1].encode()))
    if date is not None:
        out += f_bytes(7, f_sint(1, date[0]) + f_sint(2, date[1]))
    if binary_sum is not None:
        out += f_bytes(8, f_sint(1, binary_sum))
    if timestamp_utc is not None:
        out += f_bytes(9, f_sint(1, timestamp_utc[0]) + f_sint(2, timestamp_utc[1]) + f_sint(3, timestamp_utc[0]) + f_sint(4, timestamp_utc[1]))
    if has_null is not None:
        out += f_varint(10, 1 if has_null else 0)
    return out


def row_index(entries):
    """entries: per row group the bytes of its ColumnStatistics, or None (an entry without statistics)."""
    out = b""
    for st in entries:
        e = f_bytes(1, varint(0) + varint(0))  # positions (packed)
        if st is not None:
            e += f_bytes(2, st)
        out += f_bytes(1, e)
    return out


def bloom_index(filters, utf8=False):
    """filters: per row group (num_hash_functions, [u64 words])."""
    out = b""
    for k, words in filters:
        raw = b"".join(struct.pack("<Q", w) for w in words)
        out += f_bytes(1, f_varint(1, k) + (f_bytes(3, raw) if utf8 else f_bytes(2, raw)))
    return out


def hash_long(value):
    def sar(x, n):  # arithmetic shift of a 64-bit two's complement value
        x &= M64
        return ((x >> n) | (M64 << (64 - n) if x >> 63 else 0)) & M64
    key = value & M64
    key = ((~key) + (key << 21)) & M64
    key ^= sar(key, 24)
    key = (key + (key << 3) + (key << 8)) & M64
    key ^= sar(key, 14)
    key = (key + (key << 2) + (key << 4)) & M64
    key ^= sar(key, 28)
    key = (key + (key << 31)) & M64
    return key


def murmur3_64(data):
    c1, c2 = 0x87C37B91114253D5, 0x4CF5AD432745937F
    rotl = lambda v, r: ((v << r) | (v >> (64 - r))) & M64
    h1 = 104729
    nb = len(data) // 8
    for i in range(nb):
        k1 = struct.unpack_from("<Q", data, 8 * i)[0]
        k1 = rotl((k1 * c1) & M64, 31) * c2 & M64
        h1 ^= k1
        h1 = (rotl(h1, 27) * 5 + 1390208809) & M64
    tail = data[8 * nb:]
    if tail:
        k1 = int.from_bytes(tail, "little")
        k1 = rotl((k1 * c1) & M64, 31) * c2 & M64
        h1 ^= k1
    h1 ^= len(data)
    h1 ^= h1 >> 33
    h1 = h1 * 0xFF51AFD7ED558CCD & M64
    h1 ^= h1 >> 33
    h1 = h1 * 0xC4CEB9FE1A85EC53 & M64
    h1 ^= h1 >> 33
    return h1


def bloom_bits(hash64, k, n_words):
    """The bit indices add_hash sets."""
    def i32(x):
        x &= 0xFFFFFFFF
        return x - (1 << 32) if x >> 31 else x
    h1, h2 = i32(hash64), i32(hash64 >> 32)
    out = []
    for i in range(1, k + 1):
        c = i32(h1 + i * h2)
        if c < 0:
            c = ~c
        out.append((c & 0xFFFFFFFF) % (n_words * 64))
    return out


def bloom_with(hashes, k, n_words):
    words = [0] * n_words
    for h in hashes:
        for b in bloom_bits(h, k, n_words):
            words[b // 64] |= 1 << (b % 64)
    return (k, words)
