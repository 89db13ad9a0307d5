#!/usr/bin/env python3
"""Writes a TEMPORARY copy of the reference's matFact.c with the INTEGRATION.md patch applied (no B, one
mf_backend_run call instead of matrix_factorization + print_output) to the path given as argv[2].
Used by oracle/Makefile (target _ref/matFact_patched) and tests/test_integration_patch.py; the copy is never
stored in this repository -- only the compiled binary lands in oracle/_ref/ (git-ignored)."""
import re
import sys

PATCH_CALL = """
	{
		mf_problem prob = { users, items, features, iters, alpha, non_zero, (const mf_entry *) entries };
		int32_t *best = malloc(sizeof(int32_t) * (users > 0 ? users : 1));
		int rc = mf_backend_run(&prob, mat2d_data(L), mat2d_data(R), best, 0);
		if (rc != MF_OK) die(mf_backend_strerror(rc));
		for (int i = 0; i < users; i++)
			if (best[i] != -1) printf("%d\\n", best[i]);
		free(best);
	}
"""


def patch(src):
    out = src.replace('#include "datatypes.h"', '#include "datatypes.h"\n#include <stdint.h>\n#include "matfact_hip.h"', 1)
    out = re.sub(r"\tmat2d \*B = mat2d_new\(users, items\);\n", "", out)
    out = re.sub(r"\tmatrix_factorization\(B, L, R, entries, non_zero, iters, alpha\);\n", lambda m: PATCH_CALL, out)
    out = re.sub(r"\tprint_output\(B, entries\);\n", "", out)
    out = re.sub(r"\tmat2d_free\(B\);\n", "", out)
    out = out.replace('__end_benchmark("time", 1)', "")   # the root build appends `time : %f` to stdout
    assert "mf_backend_run" in out and "mat2d_new(users, items)" not in out and "print_output(B" not in out
    return out


if __name__ == "__main__":
    open(sys.argv[2], "w").write(patch(open(sys.argv[1]).read()))
