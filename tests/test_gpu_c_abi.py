"""include/orcgpu.h as a plain-C consumer sees it: tests/c_abi/abi_stripe.c is compiled with gcc (C99, -pedantic, -Werror)
against the header alone and linked with liborcgpu.so -- the compile-time proof that the header is a usable ABI (no C++,
no torch types).  On a GPU box the program is run: it stages, decodes, selects and exports a hand-made stripe and reads a
fixture file through the reader front end."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "abi_stripe.c")


def _compile(out):
    from orc_rust_amd import capi
    capi.load()  # builds liborcgpu.so when it is missing or stale
    libdir = os.path.dirname(capi.lib_path())
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC, "-L", libdir,
           "-l:" + os.path.basename(capi.lib_path()), "-Wl,-rpath," + libdir, "-o", out]
    subprocess.check_call(cmd)


def test_header_compiles_and_links_as_plain_c(tmp_path):
    exe = str(tmp_path / "abi_stripe")
    _compile(exe)
    # without a GPU the program must stop at orcgpu_open (exit code 2): the library has no CPU path
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 2 and b"no usable HIP device" in p.stderr


@pytest.mark.gpu
def test_plain_c_program_stages_decodes_selects_exports(tmp_path):
    exe = str(tmp_path / "abi_stripe")
    _compile(exe)
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "data", "test.orc"), "bigint_direct", "str_direct"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, (p.stdout.decode(), p.stderr.decode())
    assert b"c abi ok" in p.stdout and b"5 rows in batches of 2" in p.stdout
