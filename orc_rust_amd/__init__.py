"""orc_rust_amd -- MI355X-native ORC stripe -> Arrow decoder (drop-in for orc-rust's
src/encoding + src/array_decoder hot path).  The compute path is hand-written HIP for gfx950 in
csrc/, reached through the C ABI of include/orcgpu.h; this package is the thin Python binding
used by the tests and the benchmark."""
from . import capi  # noqa: F401
from .arrow_reader import ArrowReader, ArrowReaderBuilder  # noqa: F401

__all__ = ["capi", "ArrowReader", "ArrowReaderBuilder"]
