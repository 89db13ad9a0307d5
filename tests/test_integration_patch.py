"""INTEGRATION.md section 2, executed: the reference's own `main` (matFact.c) with the three lines of the patch
applied to a TEMPORARY copy, compiled with the reference's other sources where they lie and linked against
libmatfact_hip.so.  Needs the reference checkout (build container); nothing of the reference is stored here.
Without a GPU the patched binary must fail the reference's way: `Error: ...` on stderr, exit status 255."""
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT, golden_in

REF = "/root/reference"

import importlib.util

_spec = importlib.util.spec_from_file_location("patch_reference_main",
                                               os.path.join(ROOT, "oracle", "patch_reference_main.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
patch = _mod.patch


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_reference_main_with_the_integration_patch(capi, tmp_path):
    patched = patch(open(os.path.join(REF, "matFact.c")).read())
    main_c = tmp_path / "matFact_patched.c"
    main_c.write_text(patched)
    exe = str(tmp_path / "matFact_patched")
    csrc = os.path.join(ROOT, "recommender-system_amd", "csrc")
    cmd = ["gcc", "-O2", "-w", "-fopenmp", "-I", REF, "-I", os.path.join(ROOT, "include"), "-o", exe, str(main_c),
           os.path.join(REF, "mat2d.c"), os.path.join(REF, "util.c"), os.path.join(REF, "datatypes.c"),
           "-L", csrc, "-lmatfact_hip", "-Wl,-rpath," + csrc]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([exe, golden_in("inst30-40-10-2-10")], capture_output=True)
    if capi.device_count() > 0:
        assert run.returncode == 0
        assert run.stdout == open(os.path.join(GOLDEN, "inst30-40-10-2-10.out"), "rb").read()
    else:
        assert run.returncode == 255 and run.stdout == b""
        assert run.stderr == b"Error: no usable HIP device\n"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["inst0", "inst30-40-10-2-10", "instML100k"])
def test_patched_reference_binary_reproduces_out_files(name, tmp_path):
    """oracle/_ref/matFact_patched = the reference's main + the INTEGRATION.md patch + libmatfact_hip.so, built in the
    build container (binary only travels).  On the GPU it must print the reference's `.out` byte for byte."""
    exe = os.path.join(ROOT, "oracle", "_ref", "matFact_patched")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/matFact_patched not built")
    path = golden_in(name)
    if path.endswith(".gz"):
        import gzip
        raw = gzip.open(path, "rb").read()
        path = str(tmp_path / (name + ".in"))
        open(path, "wb").write(raw)
    r = subprocess.run([exe, path], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(os.path.join(GOLDEN, name + ".out"), "rb").read()
