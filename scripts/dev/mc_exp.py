"""dev: time pmc_maxcorrs of experimental library builds on a realigned MSA: mc_exp.py [workload] [library.so ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from repeatresolver_amd import _lib, datagen as dg
from repeatresolver_amd.realigner import PWReAligner
rows = [bytes(r) for r in dg.make_msa(sys.argv[1] if len(sys.argv) > 1 else "tree_medium")]
g = PWReAligner(rows, bandwidth=1000); g.trim_ends(); g.realign_round(); rows = g.export_rows(); g.close()
from repeatresolver_amd.max_correlation import max_correlations, last_timing
for lib in [None] + sys.argv[2:]:
    if lib:
        _lib._lib = None; _lib.LIB_PATH = os.path.abspath(lib)
    max_correlations(rows, 30); max_correlations(rows, 30)
    print(lib or "product", len(rows), len(rows[0]), last_timing(), flush=True)
