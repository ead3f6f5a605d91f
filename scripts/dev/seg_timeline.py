"""dev (GPU box, libpwr_diag.so): time line of the segments of single fills (window 1): dispatch skew, set-up, first row, end.
usage: seg_timeline.py [workload] [row ...] [key=value ...]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, ".")
from repeatresolver_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libpwr_diag.so")
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
args = [a for a in sys.argv[1:] if "=" not in a]
opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
wl = args[0] if args else "tree_default"
ks = [int(v) for v in args[1:]] or [0, 1, 2, 3]
NW = int(opts.get("waves", 9))
rows = [bytes(r) for r in dg.make_msa(wl)]
g = PWReAligner(rows, bandwidth=1000, window=1, **opts)
g.trim_ends(); g.total_score()
lib = _lib.load()
lib.pwr_debug_fill_diag.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
buf = (ctypes.c_uint64 * (32 * 4096))()
for k in ks:
    buf31 = None
    g.realign_row(k)
    lib.pwr_debug_fill_diag(g._h, buf)
    a = np.array(buf[31 * 4096:32 * 4096], dtype=np.uint64)[:680 * 6].reshape(680, 6)
    d = g.debug_last_job(); L = d["L"]
    use = a[:, 3] > 0
    # only entries of THIS launch: start times within 5 ms of the latest start
    t0s = a[:, 0].astype(np.float64)
    use &= t0s > t0s[use].max() - 5e5
    a = a[use]
    t_start = a[:, 0].astype(np.float64) * 0.01; t_setup = a[:, 1].astype(np.float64) * 0.01
    t_first = a[:, 2].astype(np.float64) * 0.01; t_end = a[:, 3].astype(np.float64) * 0.01
    rows_ = (a[:, 4] & np.uint64(0xfffff)).astype(np.float64); waitc = (a[:, 4] >> np.uint64(20)).astype(np.float64)
    base = t_start.min()
    print(f"row {k}: L={L}, {len(a)} segment-waves ({len(a)//NW} segments); kernel span {t_end.max()-base:.1f} us")
    print(f"   dispatch skew (start - first start): mean {np.mean(t_start-base):.1f} max {np.max(t_start-base):.1f} us")
    print(f"   set-up (start -> main loop): mean {np.mean(t_setup-t_start):.1f} max {np.max(t_setup-t_start):.1f} us")
    ok = t_first > 0
    print(f"   main loop -> first row done: mean {np.mean((t_first-t_setup)[ok]):.1f} max {np.max((t_first-t_setup)[ok]):.1f} us")
    run = t_end - t_first
    print(f"   first row -> end: mean {np.mean(run[ok]):.1f} max {np.max(run[ok]):.1f} us for {np.mean(rows_):.0f} rows = {1e3*np.mean(run[ok])/np.mean(rows_):.0f} ns/row; waiting {np.mean(waitc)/2400/np.mean(run[ok]):.1%} of it")
    print(f"   end times (us after first start): min {np.min(t_end-base):.1f} median {np.median(t_end-base):.1f} max {np.max(t_end-base):.1f}")
    segs = len(a) // NW
g.close()
