"""Builds liborcgpu.so (hipcc, gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "liborcgpu.so")


def _sources():
    out = []
    for root, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith((".hip", ".inc", ".h")):
                out.append(os.path.join(root, f))
    out.append(os.path.join(os.path.dirname(HERE), "include", "orcgpu.h"))
    return out


def build(force=False):
    srcs = _sources()
    if not force and os.path.exists(SO) and all(os.path.getmtime(s) <= os.path.getmtime(SO) for s in srcs):
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO, os.path.join(CSRC, "orcgpu.hip")]
    subprocess.check_call(cmd, cwd=CSRC)
    return SO


if __name__ == "__main__":
    print(build(force=True))
