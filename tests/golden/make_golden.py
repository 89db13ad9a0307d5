#!/usr/bin/env python3
"""Generates the committed golden fixtures from the REAL reference (oracle/_ref, built from /root/reference
by oracle/Makefile).  Run in the build container only -- the reference does not exist on the GPU box.

  <name>.in[.gz]     copy of the reference's own sample input  (data file of the reference's test set)
  <name>.out         copy of the reference's own golden output (samples/<name>.out)
  <name>.factors.npz L and R after {1, 2, 10, iters} iterations, produced by calling the reference's
                     compiled matrix_factorization() (matFact.c:29) through oracle/_ref/libmatfact_ref.so,
                     plus the reference binary's stdout for cross-checking the .out copy.

Fixtures are data only (inputs and expected outputs); no reference source text is stored.
"""
import gzip
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SAMPLES = "/root/reference/samples"
HERE = os.path.dirname(os.path.abspath(__file__))

# name -> iteration counts to snapshot ("full" = the header's iteration count)
CASES = {
    "inst0": [1, 2, 10, "full"],
    "inst1": [1, 2, 10, "full"],
    "inst2": [1, 2, 10, "full"],
    "inst30-40-10-2-10": [1, 2, 10, "full"],
    "inst1000-1000-100-2-30": [1, "full"],
    "instML100k": [1, "full"],
}
GZIP = {"instML100k", "inst1000-1000-100-2-30"}


def main():
    O.build(ref=True)
    assert O.ref_available(), "oracle/_ref is missing"
    for name, snaps in CASES.items():
        src = os.path.join(SAMPLES, name + ".in")
        inst = O.parse_in(src)
        if name in GZIP:
            with open(src, "rb") as f, gzip.GzipFile(os.path.join(HERE, name + ".in.gz"), "wb", mtime=0) as g:
                shutil.copyfileobj(f, g)
        else:
            shutil.copyfile(src, os.path.join(HERE, name + ".in"))
        shutil.copyfile(os.path.join(SAMPLES, name + ".out"), os.path.join(HERE, name + ".out"))
        stdout = O.ref_cli(src, "serial")
        assert stdout == open(os.path.join(SAMPLES, name + ".out")).read(), name
        arrays = {}
        for s in snaps:
            it = inst.iters if s == "full" else s
            L, R, _B = O.ref_run(inst, iters=it)
            arrays["L_%s" % s] = L
            arrays["R_%s" % s] = R
            print(name, "iters", it, "done", flush=True)
        np.savez_compressed(os.path.join(HERE, name + ".factors.npz"), **arrays)
    # the .mats dumps of the three tiny instances (6-decimal prints of L, R, B per iteration)
    for name in ("inst0", "inst1", "inst2"):
        shutil.copyfile(os.path.join(SAMPLES, name + ".mats"), os.path.join(HERE, name + ".mats"))


if __name__ == "__main__":
    main()
