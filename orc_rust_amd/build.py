"""Builds liborcgpu.so (hipcc, gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# ORCGPU_CFLAGS: extra compiler flags for development builds (e.g. -DORC_PROF: per-phase device timing, printed with
# ORCGPU_DEBUG=1); such a build goes to its own file so that the product library is never replaced by it.
_EXTRA = os.environ.get("ORCGPU_CFLAGS", "").split()
SO = os.path.join(CSRC, "liborcgpu.so" if not _EXTRA else "liborcgpu_dev%s.so" % os.environ.get("ORCGPU_DEV_TAG", ""))


def _sources():
    out = []
    for root, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith((".hip", ".inc", ".h")):
                out.append(os.path.join(root, f))
    out.append(os.path.join(os.path.dirname(HERE), "include", "orcgpu.h"))
    return out


def source_digest():
    """sha256 over the library's sources (names and bytes): what a PMC pass was collected for -- bench.py prices `roofline.traffic`
    with the committed passes only while the sources are still the ones they measured."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(_sources()):
        h.update(os.path.relpath(f, os.path.dirname(HERE)).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _fresh(srcs):
    return os.path.exists(SO) and all(os.path.getmtime(s) <= os.path.getmtime(SO) for s in srcs)


def build(force=False):
    """Returns the path of liborcgpu.so, compiling it first when it is missing or older than a source.
    Safe with several ranks starting at once: one of them builds (file lock), into a temporary name that
    replaces the library atomically; the others wait and then find it fresh."""
    import fcntl
    srcs = _sources()
    if not force and _fresh(srcs):
        return SO
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or not _fresh(srcs):
                hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
                tmp = SO + ".tmp.%d" % os.getpid()
                cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + _EXTRA + ["-o", tmp, os.path.join(CSRC, "orcgpu.hip")]
                subprocess.check_call(cmd, cwd=CSRC)
                os.replace(tmp, SO)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return SO


if __name__ == "__main__":
    print(build(force=True))
