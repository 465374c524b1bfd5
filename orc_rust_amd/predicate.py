"""Host-side mirror of the reference's predicate API (src/predicate.rs): `Predicate` / `PredicateValue` / `ComparisonOp`,
flattened to the pre-order node list of include/orcgpu.h (orcgpu_predicate_node) for orcgpu_reader_set_predicate and
orcgpu_predicate_row_groups."""
import ctypes as C

EQ, NE, LT, LE, GT, GE, IS_NULL, IS_NOT_NULL, AND, OR, NOT = range(11)
PV_BOOLEAN, PV_INT8, PV_INT16, PV_INT32, PV_INT64, PV_FLOAT32, PV_FLOAT64, PV_UTF8 = range(8)


class PredicateNode(C.Structure):
    _fields_ = [("op", C.c_int32), ("n_children", C.c_uint32), ("column", C.c_char_p), ("value_type", C.c_int32), ("value_is_null", C.c_int32),
                ("i", C.c_int64), ("f", C.c_double), ("s", C.c_char_p), ("s_len", C.c_uint64)]


class ColumnIndex(C.Structure):
    _fields_ = [("name", C.c_char_p), ("row_index", C.c_void_p), ("row_index_len", C.c_uint64), ("bloom_index", C.c_void_p),
                ("bloom_index_len", C.c_uint64)]


class PredicateValue:
    """PredicateValue::{Boolean, Int8, ..., Utf8}(Option<_>) (predicate.rs:29-47): value None = the SQL NULL literal."""

    def __init__(self, kind, value):
        self.kind, self.value = kind, value

    Boolean = classmethod(lambda cls, v: cls(PV_BOOLEAN, v))
    Int8 = classmethod(lambda cls, v: cls(PV_INT8, v))
    Int16 = classmethod(lambda cls, v: cls(PV_INT16, v))
    Int32 = classmethod(lambda cls, v: cls(PV_INT32, v))
    Int64 = classmethod(lambda cls, v: cls(PV_INT64, v))
    Float32 = classmethod(lambda cls, v: cls(PV_FLOAT32, v))
    Float64 = classmethod(lambda cls, v: cls(PV_FLOAT64, v))
    Utf8 = classmethod(lambda cls, v: cls(PV_UTF8, v))


class Predicate:
    """Predicate::{Comparison, IsNull, IsNotNull, And, Or, Not} with the reference's constructors (predicate.rs:98-180)."""

    def __init__(self, op, column=None, value=None, children=()):
        self.op, self.column, self.value, self.children = op, column, value, list(children)

    @classmethod
    def comparison(cls, column, op, value):
        return cls(op, column, value)

    eq = classmethod(lambda cls, c, v: cls(EQ, c, v))
    ne = classmethod(lambda cls, c, v: cls(NE, c, v))
    lt = classmethod(lambda cls, c, v: cls(LT, c, v))
    lte = classmethod(lambda cls, c, v: cls(LE, c, v))
    gt = classmethod(lambda cls, c, v: cls(GT, c, v))
    gte = classmethod(lambda cls, c, v: cls(GE, c, v))
    is_null = classmethod(lambda cls, c: cls(IS_NULL, c))
    is_not_null = classmethod(lambda cls, c: cls(IS_NOT_NULL, c))

    @classmethod
    def and_(cls, predicates):
        return cls(AND, children=predicates)

    @classmethod
    def or_(cls, predicates):
        return cls(OR, children=predicates)

    @classmethod
    def not_(cls, predicate):
        return cls(NOT, children=[predicate])

    def flatten(self):
        """-> (ctypes array of PredicateNode in pre-order, objects to keep alive)."""
        out, keep = [], []

        def walk(p):
            n = PredicateNode()
            n.op = p.op
            n.n_children = len(p.children)
            if p.column is not None:
                b = p.column.encode()
                keep.append(b)
                n.column = b
            v = p.value
            if v is not None:
                n.value_type = v.kind
                n.value_is_null = 1 if v.value is None else 0
                if v.value is not None:
                    if v.kind == PV_UTF8:
                        b = v.value.encode() if isinstance(v.value, str) else bytes(v.value)
                        keep.append(b)
                        n.s, n.s_len = b, len(b)
                    elif v.kind in (PV_FLOAT32, PV_FLOAT64):
                        n.f = float(v.value)
                    else:
                        n.i = int(v.value)
            out.append(n)
            for c in p.children:
                walk(c)

        walk(self)
        arr = (PredicateNode * len(out))(*out)
        return arr, keep
