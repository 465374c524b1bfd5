"""A small LZO1X ENCODER for the tests (no LZO library in the image): greedy matching over a hash of 4-byte windows,
instructions M2 / M3 / M4 as the format describes them (Linux kernel Documentation/staging/lzo.rst), literal runs, the
0x11 0x00 0x00 end marker.  Not a port of any encoder and not tuned: it only has to produce streams that a conforming
decoder (lzokay: the reference's crate) accepts, with every instruction form in them.  `force` picks forms real encoders
rarely emit."""


def _length_ext(value):
    """zero bytes + final byte for a length field whose short form is 0: value >= 1"""
    out = bytearray()
    while value > 255:
        out.append(0)
        value -= 255
    out.append(value)
    return bytes(out)


def _literal_run(lits):
    n = len(lits)
    assert n >= 4
    if n <= 18:
        return bytes([n - 3]) + lits
    return bytes([0]) + _length_ext(n - 18) + lits


def _match(dist, length, use_m2=True):
    """(instruction bytes with S = 0, index of the byte that carries S)"""
    assert length >= 3 and 1 <= dist <= 49151
    if use_m2 and length <= 8 and dist <= 2048:
        d = dist - 1
        b = bytearray([((length - 1) << 5) | ((d & 7) << 2), d >> 3])
        return b, 0
    if dist <= 16384:
        b = bytearray()
        if length - 2 <= 31:
            b.append(0x20 | (length - 2))
        else:
            b.append(0x20)
            b += _length_ext(length - 2 - 31)
        s_at = len(b)
        b += ((dist - 1) << 2).to_bytes(2, "little")
        return b, s_at
    d = dist - 16384
    h = (d >> 14) & 1
    b = bytearray()
    if length - 2 <= 7:
        b.append(0x10 | (h << 3) | (length - 2))
    else:
        b.append(0x10 | (h << 3))
        b += _length_ext(length - 2 - 7)
    s_at = len(b)
    b += ((d & 0x3FFF) << 2).to_bytes(2, "little")
    return b, s_at


def tokens(data, max_dist=49151, min_len=3, max_len=4000):
    """greedy LZ77: [(literal bytes, dist, length)], the last entry has dist 0 (trailing literals)"""
    n = len(data)
    table = {}
    out = []
    i = lit0 = 0
    while i + 4 <= n:
        key = data[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and 0 < i - cand <= max_dist:
            ln = 4
            while i + ln < n and ln < max_len and data[cand + ln] == data[i + ln]:
                ln += 1
            if ln >= max(min_len, 4):
                out.append((data[lit0:i], i - cand, ln))
                for k in range(i + 1, min(i + ln, n - 3), 7):  # a few positions inside the match keep the table fresh
                    table[data[k:k + 4]] = k
                i += ln
                lit0 = i
                continue
        i += 1
    out.append((data[lit0:], 0, 0))
    return out


def compress(data, use_m2=True, first_byte_form=True):
    data = bytes(data)
    toks = tokens(data)
    out = bytearray()
    s_slot = None  # (index in out of the byte whose two low bits take S)
    first = True
    for lits, dist, length in toks:
        t = len(lits)
        if t:
            if s_slot is not None and t <= 3:
                out[s_slot] |= t
                out += lits
            elif first and first_byte_form and t <= 238 and t not in (1, 2, 3):
                out.append(17 + t)
                out += lits
            elif first and t <= 3:
                out.append(17 + t)
                out += lits
            else:
                out += _literal_run(lits)  # (state 0 here: the match before carried S = 0, or this is the start of the stream)
        first = False
        if length:
            b, s_at = _match(dist, length, use_m2)
            s_slot = len(out) + s_at
            out += b
        else:
            s_slot = None
    return bytes(out) + b"\x11\x00\x00"
