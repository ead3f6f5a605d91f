"""dev (GPU box, libpwr_diag.so): where the time of a batch's tail goes -- phase timers of k_trace_blk (the commit is three
launches since round 4: its phases are kernels, see scripts/kstats.sh).
usage: phases.py [workload] [rows] [key=value ...]"""
import ctypes, os, sys, time
sys.path.insert(0, ".")
from repeatresolver_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libpwr_diag.so")
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
wl = sys.argv[1] if len(sys.argv) > 1 else "tree_default"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[3:]}
rows = [bytes(r) for r in dg.make_msa(wl)]
g = PWReAligner(rows, bandwidth=1000, **opts)
g.trim_ends(); g.total_score()
lib = _lib.load()
lib.pwr_debug_phase_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
buf = (ctypes.c_uint64 * 32)()
g.realign_rows(0, 100)
lib.pwr_debug_phase_times(g._h, buf); g.reset_stats()
t0 = time.time(); g.realign_rows(100, n); dt = time.time() - t0
lib.pwr_debug_phase_times(g._h, buf)
st = g.stats()
d = list(buf)
print(f"{n} rows in {dt:.3f} s, {st['batches']} batches ({1e3*dt/st['batches']:.3f} ms each), committed {st['rows_committed']}")
for nm, o in (("bottom chunk", 16), ("top chunk", 20)):
    k = max(1, d[o + 3])
    print(f"trace {nm}: phase0 {d[o]/k/100:.1f} us, waiting for the chunk above {d[o+1]/k/100:.1f} us, phase1 {d[o+2]/k/100:.1f} us  (job 0 of {k} launches)")
print(f"k_trace_blk job 0: chunks per trace {d[26]/max(1,d[19]):.0f}, chunks that retraced {d[24]/max(1,d[19]):.1f}, looks of the bottom chunk {d[25]/max(1,d[19]):.1f}")
g.close()
