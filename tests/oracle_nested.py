"""The CPU oracle for NESTED columns (TEST INFRASTRUCTURE, part of the oracle: the product never imports it).

A restatement of the reference's composite array decoders over the oracle's C primitives (oracle/liborc_oracle.so: the
PRESENT bit decoder, Integer RLE, byte RLE, and the flat column decoder with a parent's validity):

    derive_present_vec / merge_parent_present   array_decoder/mod.rs:199-252
    StructArrayDecoder::next_batch              array_decoder/struct_decoder.rs:58-78
    ListArrayDecoder::next_batch                array_decoder/list.rs:63-87
    MapArrayDecoder::next_batch                 array_decoder/map.rs:74-104
    UnionArrayDecoder::next_batch               array_decoder/union.rs:69-136

A node's next_batch(n, parent_present) returns a pyarrow array of n rows, or raises OracleError(status) -- the reference's
`?`: any decoder's error fails the whole batch.  numpy throughout; no per-row Python loops.
"""
import ctypes as C

import numpy as np
import pyarrow as pa

import arrow_util as A
import oracle_lib as O

LIST, MAP, STRUCT, UNION = 10, 11, 12, 13
PRESENT, DATA, LENGTH = 0, 1, 2


class OracleError(Exception):
    def __init__(self, status, where=""):
        super().__init__("oracle status %d %s" % (status, where))
        self.status = status


class _Stream:
    """A decoder of the C oracle over one stream of the stripe (kept alive with its reader)."""

    def __init__(self, data, compression, block_size, make):
        L = O.lib()
        self.data = bytes(data)
        self.r = L.oo_reader_new(self.data, len(self.data), O.COMP[compression], block_size)
        self.d = make(L, self.r)


class Present(_Stream):
    """PresentDecoder (mod.rs:192-214): BooleanDecoder over the PRESENT stream."""

    def __init__(self, data, compression, block_size):
        super().__init__(data, compression, block_size, lambda L, r: L.oo_bool_new(r))

    def next_buffer(self, n):
        buf = np.zeros(max(n, 1), dtype=np.uint8)
        st = O.lib().oo_bool_decode(self.d, buf.ctypes.data, n)
        return None if st else buf[:n].astype(bool)


def derive_present(present, parent, n):
    """derive_present_vec (mod.rs:216-252).  Returns a bool array with at least one False, or None -- also when the PRESENT
    stream fails to decode: `_ => None` swallows the error and the batch is decoded as if every row were there."""
    if present is not None and parent is not None:
        own = present.next_buffer(int(parent.sum()))
        if own is None:
            return None
        out = np.zeros(n, dtype=bool)
        out[parent] = own  # merge_parent_present: the stream has one bit per row the parent has
    elif present is not None:
        out = present.next_buffer(n)
        if out is None:
            return None
    elif parent is not None:
        out = parent.copy()
    else:
        return None
    return out if not out.all() else None


def _mask(present):
    return None if present is None else pa.array(~present)


class Leaf:
    def __init__(self, f, stripe, cid):
        self.kind = f.types[cid].kind
        self.t = f.types[cid]
        self.col = f.oracle_column(stripe, cid)
        if self.col.status:
            raise OracleError(self.col.status, "column %d" % cid)

    def next_batch(self, n, parent):
        b = self.col.next_batch(n, None if parent is None else parent.astype(np.uint8))
        if b["status"]:
            raise OracleError(b["status"])
        return A.to_arrow(self.kind, b, self.t.precision, self.t.scale)


class Struct:
    def __init__(self, f, stripe, cid):
        t = f.types[cid]
        ps = stripe.streams.get((cid, PRESENT))
        self.present = Present(ps, f.compression_name, f.block_size) if ps is not None else None
        self.names = list(t.field_names)
        self.kids = [build(f, stripe, k) for k in t.subtypes]

    def next_batch(self, n, parent):
        present = derive_present(self.present, parent, n)
        kids = [k.next_batch(n, present) for k in self.kids]
        if not kids:
            return pa.array([{} if present is None or p else None for p in (present if present is not None else np.ones(n, bool))], type=pa.struct([]))
        return pa.StructArray.from_arrays(kids, names=self.names, mask=_mask(present))


class _Lengths(_Stream):
    def __init__(self, data, compression, block_size, version):
        super().__init__(data, compression, block_size, lambda L, r: L.oo_int_rle_new(r, version, 0, 64))

    def decode_spaced(self, n, present):
        """get_unsigned_int_decoder + decode / decode_spaced (encoding/mod.rs:64-91): zeros where the row is null"""
        k = n if present is None else int(present.sum())
        buf = np.zeros(max(k, 1), dtype=np.int64)
        st = O.lib().oo_int_rle_decode(self.d, buf.ctypes.data, k)
        if st:
            raise OracleError(st, "LENGTH")
        if present is None:
            return buf[:n].copy()
        out = np.zeros(n, dtype=np.int64)
        out[present] = buf[:k]
        return out


class List:
    def __init__(self, f, stripe, cid):
        t = f.types[cid]
        ps = stripe.streams.get((cid, PRESENT))
        self.present = Present(ps, f.compression_name, f.block_size) if ps is not None else None
        enc = stripe.encodings[cid][0] if cid < len(stripe.encodings) else 0
        # StreamMap::get: a missing stream is an empty one
        self.lengths = _Lengths(stripe.streams.get((cid, LENGTH), b""), f.compression_name, f.block_size, 2 if enc in (2, 3) else 1)
        self.kids = [build(f, stripe, k) for k in t.subtypes]

    def _offsets(self, n, parent):
        present = derive_present(self.present, parent, n)
        lengths = self.lengths.decode_spaced(n, present)
        total = int(lengths.sum())
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lengths, out=offsets[1:])
        if offsets[-1] > 0x7fffffff or (lengths < 0).any():
            raise OracleError(O.ARROW, "offsets")
        return present, pa.array(offsets.astype(np.int32), type=pa.int32()), total

    def next_batch(self, n, parent):
        present, offsets, total = self._offsets(n, parent)
        child = self.kids[0].next_batch(total, None)  # (the elements have no parent: list.rs:79)
        return pa.ListArray.from_arrays(offsets, child, mask=_mask(present))


class Map(List):
    def next_batch(self, n, parent):
        present, offsets, total = self._offsets(n, parent)
        keys = self.kids[0].next_batch(total, None)
        items = self.kids[1].next_batch(total, None)
        # map.rs:90-99: keys and values become a StructArray whose `keys` field is not nullable -- arrow-rs refuses a key column with a
        # null in it ("Found unmasked nulls for non-nullable StructArray field": ArrowError)
        if keys.null_count:
            raise OracleError(O.ARROW, "null keys")
        return pa.MapArray.from_arrays(offsets, keys, items, mask=_mask(present))


class _Tags(_Stream):
    def __init__(self, data, compression, block_size):
        super().__init__(data, compression, block_size, lambda L, r: L.oo_byte_rle_new(r))

    def decode_spaced(self, n, present):
        k = n if present is None else int(present.sum())
        buf = np.zeros(max(k, 1), dtype=np.int8)
        st = O.lib().oo_byte_rle_decode(self.d, buf.ctypes.data, k)
        if st:
            raise OracleError(st, "tags")
        if present is None:
            return buf[:n].copy()
        out = np.zeros(n, dtype=np.int8)
        out[present] = buf[:k]
        return out


class Union:
    def __init__(self, f, stripe, cid):
        t = f.types[cid]
        ps = stripe.streams.get((cid, PRESENT))
        self.present = Present(ps, f.compression_name, f.block_size) if ps is not None else None
        self.tags = _Tags(stripe.streams.get((cid, DATA), b""), f.compression_name, f.block_size)
        self.kids = [build(f, stripe, k) for k in t.subtypes]

    def next_batch(self, n, parent):
        present = derive_present(self.present, parent, n)
        tags = self.tags.decode_spaced(n, present)
        children = []
        for i, kid in enumerate(self.kids):
            cp = tags == i  # where the parent expects the value of the child
            if i == 0 and present is not None:
                cp = cp & present  # the Union's own nulls live in the first child (union.rs:104-116)
            children.append(kid.next_batch(n, cp))  # (always Some(&present): a child without nulls drops it again)
        return pa.UnionArray.from_sparse(pa.array(tags, type=pa.int8()), children, field_names=[str(i) for i in range(len(children))],
                                         type_codes=list(range(len(children))))


def build(f, stripe, cid):
    k = f.types[cid].kind
    return {STRUCT: Struct, LIST: List, MAP: Map, UNION: Union}.get(k, Leaf)(f, stripe, cid)


def read_column(f, cid, batch_size=8192):
    """Every batch of root column `cid` over all stripes: [pyarrow array], as NaiveStripeDecoder yields them
    (array_decoder/mod.rs:555-604: batches of batch_size rows, the last one of a stripe shorter)."""
    out = []
    for s in f.stripes:
        node = build(f, s, cid)
        left = s.number_of_rows
        while left > 0:
            n = min(batch_size, left)
            out.append(node.next_batch(n, None))
            left -= n
    return out
