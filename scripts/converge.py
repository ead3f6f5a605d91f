"""dev tool: realign a synthetic workload until a round no longer improves the score (PW_ReAligner.c:1681-1754) and
print what each round cost.  usage: converge.py <workload> [max_rounds] [KEY=VALUE ...]"""
import sys, time, json
sys.path.insert(0, ".")
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
wl = sys.argv[1] if len(sys.argv) > 1 else "tree_medium"
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = [bytes(r) for r in dg.make_msa(wl)]
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[3:]}          # any knob of pwr_set_option, KEY=VALUE
g = PWReAligner(rows, bandwidth=1000, **opts)
g.trim_ends()
best = g.total_score()
g.realign_rows(0, 0)                 # the MSA into HBM before the clock starts (the reference's clock starts after MMA_Einlesen too, PW:1679)
print("rows", len(rows), "columns", len(rows[0]), "score", best, flush=True)
t_all = time.time()
out = []
for rnd in range(1, cap + 1):
    t0 = time.time()
    g.realign_round()
    s = g.total_score()
    st = g.stats()
    dt = time.time() - t0
    out.append({"round": rnd, "seconds": round(dt, 3), "score": s, "rows_changed_total": st["rows_changed"], "cells_total": st["cells_reference"],
                "batches_total": st["batches"], "seg_fails_total": st["seg_fails"], "warm_now": g.get_option("warm_now")})
    print(out[-1], flush=True)
    if s >= best:
        break
    best = s
tot = time.time() - t_all
print(json.dumps({"workload": wl, "rounds": len(out), "seconds_to_converge": round(tot, 2), "final_score": best, "cells": out[-1]["cells_total"],
                  "cells_per_s": out[-1]["cells_total"] / tot, "per_round": out}))
g.close()
