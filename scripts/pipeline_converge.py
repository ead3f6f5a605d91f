#!/usr/bin/env python3
"""BASELINE.json's second metric, wall-clock to convergence, for the three GPU steps chained through their drop-in binaries
as the reference's README chains them:  DataSimulator files -> InitialAligner -> PW_ReAligner (until a round no longer
improves the score, PW:1681-1754) -> MaxCorrelation.  usage: pipeline_converge.py [workload] [outdir]  -> one JSON line"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from repeatresolver_amd import datagen as dg

wl = sys.argv[1] if len(sys.argv) > 1 else "tree_default"
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/pipeline_converge"
os.makedirs(out, exist_ok=True)
cfg = dg.CONFIGS[wl]
prefix = os.path.join(out, "Sim")
t0 = time.time()
counts = dg.write_dataset(prefix, cfg)
t_sim = time.time() - t0
csrc = os.path.join(ROOT, "repeatresolver_amd", "csrc")
t0 = time.time()
p = subprocess.run([os.path.join(csrc, "InitialAligner"), prefix + "_Template.fasta", prefix + "Seq.fasta", "-o", prefix + "_MSA", "-s", prefix + "_SeqClass"],
                   capture_output=True, text=True)
t_ia = time.time() - t0
assert p.returncode == 0, p.stdout + p.stderr
ia_lines = p.stdout.splitlines()
t0 = time.time()
log = os.path.join(out, "pw.log")
with open(log, "w") as f:
    p = subprocess.run([os.path.join(csrc, "PW_ReAligner"), prefix + "_MSA", "-o", prefix + "_MSAreal"], stdout=f, stderr=subprocess.STDOUT)
t_pw = time.time() - t0
assert p.returncode == 0
lines = open(log, encoding="latin1").read().splitlines()
scores = [l for l in lines if l.startswith("OverallScore")]
# the same MSA through the API, nothing written: what the round loop itself costs (the drop-in's file work rides beside it)
from repeatresolver_amd.realigner import PWReAligner
t0 = time.time()
g = PWReAligner.from_file(prefix + "_MSA")
g.trim_ends()
best = g.total_score()
t_api_load = time.time() - t0
t0 = time.time()
api_rounds = 0
while True:
    g.realign_round()
    api_rounds += 1
    tot = g.total_score()
    if tot < best:
        best = tot
    else:
        break
t_api = time.time() - t0
g.close()
t0 = time.time()
p = subprocess.run([os.path.join(csrc, "MaxCorrelation"), "Sim_MSAreal"], capture_output=True, text=True, cwd=out)
t_mc = time.time() - t0
assert p.returncode == 0, p.stdout + p.stderr
mc_lines = p.stdout.splitlines()
mc_vals = [float(v) for v in open(os.path.join(out, "MaxCorrsOf_Sim_MSAreal")).read().split()]
print(json.dumps({"workload": wl, "max_correlation_s": round(t_mc, 1), "max_correlation_stdout": mc_lines,
                  "maxcorrs": {"n": len(mc_vals), "nonzero": sum(v > 0 for v in mc_vals), "max": max(mc_vals)}, "dataset": counts, "simulate_s": round(t_sim, 1),
                  "initial_aligner_s": round(t_ia, 2), "initial_aligner_stdout": ia_lines,
                  "pw_realigner_s": round(t_pw, 1), "rounds": len(scores) - 2,
                  "api_round_loop_s": round(t_api, 1), "api_rounds": api_rounds, "api_read_parse_upload_s": round(t_api_load, 1),
                  "pw_realigner_over_api_round_loop": round(t_pw / t_api, 3), "score_lines": scores,
                  "dims": [l for l in lines if l.startswith("Rows")], "msa_bytes": os.path.getsize(prefix + "_MSA"),
                  "msareal_bytes": os.path.getsize(prefix + "_MSAreal")}))
