"""Dev aid (GPU box): per-wave counters of k_fill_v3 for a few single realignments (library built with -DPWR_DIAG).
usage: diag.py [workload] [rows...]"""
import ctypes, os, sys
sys.path.insert(0, ".")
from repeatresolver_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libpwr_diag.so")
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
wl = sys.argv[1] if len(sys.argv) > 1 else "tree_medium"
waves = int(os.environ.get("WAVES", "9"))
ks = [int(v) for v in sys.argv[2:] if "=" not in v] or [0, 1, 2]
opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2:] if "=" in a}
rows = [bytes(r) for r in dg.make_msa(wl)]
g = PWReAligner(rows, bandwidth=1000, window=1, waves=waves, **opts)
g.trim_ends(); g.total_score()
lib = _lib.load()
lib.pwr_debug_fill_diag.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
buf = (ctypes.c_uint64 * (32 * 4096))()
for k in ks:
    g.realign_row(k)
    mhz, us = g.debug_fill_clock()
    lib.pwr_debug_fill_diag(g._h, buf)
    L = buf[7]
    print(f"row {k}: L={L} fill {us:.1f} us = {1e3*us/max(L,1):.1f} ns/DP row, clock {mhz:.0f} MHz")
    base1 = min(buf[w*4096+9] for w in range(waves)); base2 = min(buf[w*4096+10] for w in range(waves))
    for w in range(waves):
        d = buf[w*4096:(w+1)*4096]
        print(f"  wave {w}: reached row 1024 at +{(d[9]-base1)*10} ns, row 2048 at +{(d[10]-base2)*10} ns (after the first wave to get there)")
        tot = d[0]
        print(f"  wave {w}: cycles {tot} ({tot/max(L,1):.0f}/row) wait fast {d[1]/max(tot,1):.2%} gen {d[2]/max(tot,1):.2%} setup {d[3]/max(tot,1):.2%} | interior {d[11]/max(d[4]>>32,1):.0f} cyc/row, other fast rows {d[8]/max(d[4]&0xffffffff,1):.0f} cyc/row (incl waits) | rows interior {d[4]>>32} generic {d[4]&0xffffffff} general {d[5]>>32} nowork {d[5]&0xffffffff} runs {d[6]>>32} switches {d[6]&0xffffffff}")
if os.environ.get("LOG"):
    logs = {}
    for w in range(waves):
        n = min(buf[w*4096+12], 1270)
        logs[w] = [(buf[w*4096+256+3*i] & 0xffffffff, buf[w*4096+257+3*i]) for i in range(n)]
        full = [(buf[w*4096+256+3*i], buf[w*4096+257+3*i], buf[w*4096+258+3*i]) for i in range(n)]
        if os.environ.get("LOG") == "2" and w in (0, 4):
            print(f"  ---- wave {w}: per stretch: end row, role, rows, cycles/row busy, cycles/row waiting")
            pt, pw = None, None
            for (xc, t, wt) in full:
                xx, cls_, nr = xc & 0xffffffff, (xc >> 32) & 0xff, xc >> 40
                if pt is not None and 1000 <= xx <= 1400:
                    dt = (t - pt) * 10 * 2.39
                    print(f"       x={xx:5d} role {cls_} rows {nr:2d}: busy {(dt-(wt-pw))/max(nr,1):6.0f}  wait {(wt-pw)/max(nr,1):6.0f}")
                pt, pw = t, wt
    evs = []
    for w in range(waves):
        ng = min(buf[w*4096+5] >> 32, 36)
        for e in range(ng):
            evs.append((buf[w*4096+17+3*e], w, buf[w*4096+16+3*e], buf[w*4096+18+3*e]))
    evs.sort()
    for i in range(20, 23):
        t0, w, xs, t1 = evs[i]
        prev = t1
        print(f"  -- wave {w} takes a strip at row {xs}: waited {(t1-t0)*10/1000:.2f} us; its progress afterwards (rows done, us since data, cycles/row over the block):")
        for (xx, t) in logs[w]:
            if t >= t1 and xx <= xs + 60:
                pass
        last_x = xs
        for (xx, t) in logs[w]:
            if t >= t1 and xx > xs and xx <= xs + 64:
                print(f"       x={xx:5d} +{(t-t1)*10/1000:6.2f} us  {(t-prev)*10*2.39/max(xx-last_x,1):6.0f} cyc/row")
                prev, last_x = t, xx
    for i in range(12, min(len(evs) - 1, 14)):
        t0, w, xs, t1 = evs[i]
        t0n, wn, xsn, t1n = evs[i + 1]
        lg = logs[w]
        tdone = next((t for (xx, t) in lg if xx > xsn and t >= t1), None)
        print(f"  wave {w} took strip at row {xs}: data at t={t1*10/1000:8.2f} us; next switcher (wave {wn}) needs row {xsn}; wave {w} had it done at +{(tdone-t1)*10/1000 if tdone else -1:6.2f} us; wave {wn} got its data at +{(t1n-t1)*10/1000:6.2f} us")
if os.environ.get("EVENTS"):
    tbase = min(buf[w*4096+17] for w in range(waves) if buf[w*4096+17])
    evs = []
    for w in range(waves):
        ng = min(buf[w*4096+5] >> 32, 36)
        for e in range(ng):
            xx, t0, t1 = buf[w*4096+16+3*e], buf[w*4096+17+3*e], buf[w*4096+18+3*e]
            evs.append((t0, w, xx, t1))
    for t0, w, xx, t1 in sorted(evs)[:60]:
        print(f"  t={(t0-tbase)*10/1000:8.2f} us wave {w} general row x={xx:5d} waited {(t1-t0)*10/1000:7.2f} us")
g.close()
