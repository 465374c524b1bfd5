"""Minimal ORC container reader for the tests (TEST INFRASTRUCTURE ONLY).

Parses the file tail and stripe footers with a hand-written protobuf wire reader (no protoc in
the image) and hands out the raw per-(column, kind) stream bytes of each stripe -- exactly the
inputs available at the reference's L3 -> L2 seam (`Stripe { columns, stream_map, .. }`,
stripe.rs:119-182).  Footers are decompressed with the CPU oracle.  Used to feed the same
streams to the oracle and to the HIP path.  Follows reader/metadata.rs:180-247,
stripe.rs:127-182, schema.rs:390-495, column.rs:40-59 and format/orc_proto.proto.
"""
import oracle_lib as O

KIND_NAMES = ["BOOLEAN", "BYTE", "SHORT", "INT", "LONG", "FLOAT", "DOUBLE", "STRING", "BINARY", "TIMESTAMP", "LIST", "MAP",
              "STRUCT", "UNION", "DECIMAL", "DATE", "VARCHAR", "CHAR", "TIMESTAMP_INSTANT"]
COMPRESSION = ["none", "zlib", "snappy", "lzo", "lz4", "zstd"]
PRESENT, DATA, LENGTH, DICTIONARY_DATA, DICTIONARY_COUNT, SECONDARY, ROW_INDEX, BLOOM_FILTER, BLOOM_FILTER_UTF8 = range(9)
FLAT_KINDS = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 14, 15, 16, 17, 18}


def _varint(b, p):
    v = s = 0
    while True:
        c = b[p]
        p += 1
        v |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return v, p


def pb_fields(b):
    """Yield (field_number, wire_type, value) of one protobuf message."""
    p = 0
    n = len(b)
    while p < n:
        key, p = _varint(b, p)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, p = _varint(b, p)
        elif wt == 2:
            ln, p = _varint(b, p)
            v = b[p:p + ln]
            p += ln
        elif wt == 1:
            v = b[p:p + 8]
            p += 8
        elif wt == 5:
            v = b[p:p + 4]
            p += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield f, wt, v


def _packed(v, wt):
    if wt == 0:
        return [v]
    out, p = [], 0
    while p < len(v):
        x, p = _varint(v, p)
        out.append(x)
    return out


class OrcType:
    def __init__(self):
        self.kind = 12
        self.subtypes = []
        self.field_names = []
        self.maximum_length = 0
        self.precision = 0
        self.scale = 0


class Stripe:
    def __init__(self):
        self.offset = self.index_length = self.data_length = self.footer_length = self.number_of_rows = 0
        self.streams = {}  # (column, kind) -> bytes
        self.stream_list = []  # (kind, column, length) in file order
        self.encodings = []  # [(kind, dictionary_size)] by column id
        self.writer_timezone = None


class OrcFile:
    def __init__(self, path_or_bytes):
        if isinstance(path_or_bytes, (bytes, bytearray)):
            self.buf = bytes(path_or_bytes)
        else:
            with open(path_or_bytes, "rb") as f:
                self.buf = f.read()
        buf = self.buf
        ps_len = buf[-1]
        ps = buf[-1 - ps_len:-1]
        self.footer_length = self.metadata_length = 0
        self.compression = 0
        self.block_size = 262144
        for f, wt, v in pb_fields(ps):
            if f == 1:
                self.footer_length = v
            elif f == 2:
                self.compression = v
            elif f == 3:
                self.block_size = v
            elif f == 5:
                self.metadata_length = v
        self.compression_name = COMPRESSION[self.compression]
        end = len(buf) - 1 - ps_len
        footer = self._decompress(buf[end - self.footer_length:end])
        self.types = []
        self.stripes = []
        self.number_of_rows = 0
        self.row_index_stride = None
        for f, wt, v in pb_fields(footer):
            if f == 3:
                s = Stripe()
                for g, _, x in pb_fields(v):
                    if g == 1:
                        s.offset = x
                    elif g == 2:
                        s.index_length = x
                    elif g == 3:
                        s.data_length = x
                    elif g == 4:
                        s.footer_length = x
                    elif g == 5:
                        s.number_of_rows = x
                self.stripes.append(s)
            elif f == 4:
                t = OrcType()
                t.kind = 0
                for g, w2, x in pb_fields(v):
                    if g == 1:
                        t.kind = x
                    elif g == 2:
                        t.subtypes += _packed(x, w2)
                    elif g == 3:
                        t.field_names.append(bytes(x).decode())
                    elif g == 4:
                        t.maximum_length = x
                    elif g == 5:
                        t.precision = x
                    elif g == 6:
                        t.scale = x
                self.types.append(t)
            elif f == 6:
                self.number_of_rows = v
            elif f == 8:
                self.row_index_stride = v
        for s in self.stripes:
            self._read_stripe(s)

    def _decompress(self, raw):
        st, out = O.stream_decompress(raw, self.compression_name, self.block_size)
        if st:
            raise ValueError("footer decompress failed: %d" % st)
        return out

    def _read_stripe(self, s):
        fo = s.offset + s.index_length + s.data_length
        footer = self._decompress(self.buf[fo:fo + s.footer_length])
        off = s.offset
        for f, wt, v in pb_fields(footer):
            if f == 1:
                kind = col = length = 0
                for g, _, x in pb_fields(v):
                    if g == 1:
                        kind = x
                    elif g == 2:
                        col = x
                    elif g == 3:
                        length = x
                s.stream_list.append((kind, col, length))
                s.streams[(col, kind)] = self.buf[off:off + length]
                off += length
            elif f == 2:
                kind = dsz = 0
                for g, _, x in pb_fields(v):
                    if g == 1:
                        kind = x
                    elif g == 2:
                        dsz = x
                s.encodings.append((kind, dsz))
            elif f == 3:
                s.writer_timezone = bytes(v).decode()

    # -- schema helpers ------------------------------------------------------------------
    def root_columns(self):
        """[(name, column_id, OrcType)] of the root struct's children (schema.rs:154-162)."""
        if not self.types:
            return []
        root = self.types[0]
        return [(n, c, self.types[c]) for n, c in zip(root.field_names, root.subtypes)]

    def flat_columns(self):
        return [(n, c, t) for n, c, t in self.root_columns() if t.kind in FLAT_KINDS]

    def column_streams(self, stripe, column_id):
        """{Stream.Kind: bytes} for one column of one stripe (data streams only)."""
        return {k: v for (c, k), v in stripe.streams.items() if c == column_id and k in (PRESENT, DATA, LENGTH, DICTIONARY_DATA, SECONDARY)}

    def oracle_column(self, stripe, column_id, ts_unit=3, ts_base=None):
        t = self.types[column_id]
        enc, dsz = stripe.encodings[column_id] if column_id < len(stripe.encodings) else (0, 0)
        tz = None
        if ts_base is None:
            ts_base = 1420070400
            if stripe.writer_timezone and t.kind == 9:
                # Stripe::writer_tz (stripe.rs:167-171): base epoch in that zone, batches re-labelled to UTC
                import tz_table
                ts_base = tz_table.orc_epoch(stripe.writer_timezone)
                tz = tz_table.table(stripe.writer_timezone)
        return O.Column(t.kind, enc, self.column_streams(stripe, column_id), dictionary_size=dsz, precision=t.precision, scale=t.scale,
                        ts_unit=ts_unit, ts_base=ts_base, compression=self.compression_name, block_size=self.block_size, tz=tz)
