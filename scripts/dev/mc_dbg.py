import sys, os, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from test_mc_oracle import small_msa
from repeatresolver_amd.max_correlation import max_correlations, last_timing
lib = ctypes.CDLL("/root/repo/oracle/libmcoracle.so")
lib.mco_maxcorrs.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
rows = small_msa()
for mincov in (4, 12):
    got = max_correlations(rows, mincov)
    exp = np.zeros(len(rows[0]) * 5)
    lib.mco_maxcorrs(len(rows), len(rows[0]), b"".join(rows), mincov, exp.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    d = np.abs(got - exp)
    print(mincov, "max diff", d.max(), "n diff>1e-9", (d > 1e-9).sum(), "zero mismatch", ((got == 0) != (exp == 0)).sum(), last_timing())
    bad = np.nonzero(d > 1e-9)[0][:10]
    for v in bad: print("  var", v, "col", v // 5, "sym", v % 5, "got", got[v], "exp", exp[v])
    zm = np.nonzero((got == 0) != (exp == 0))[0][:10]
    for v in zm: print("  zero-mismatch var", v, "got", got[v], "exp", exp[v])
