#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Full-size digests for the two inputs that had none (round-3 verdict, item 4):

  --pipeline   the MSA `bench.py` runs by default (`--input pipeline`): the reads `pipeline.initial_msa` keeps
               (datagen.simulate_dataset("tree_default"), cut to their repeat part, >= min_aligned bases) are written
               as FASTA, aligned by the REFERENCE's own InitialAligner (oracle/_ref/initial_aligner, compiled from
               /root/reference/InitialAligner.c in place), and its MSA is then given to the REFERENCE's PW_ReAligner
               (oracle/_ref/pw_ref) for ONE round (the file it rewrites at PW:1741 is hashed, then the process is
               stopped).  -> tests/golden/pipeline_tree_default_round1.json (reference-made).
  --config3    one full round of `distributed_stress` (BASELINE.json configs[2]: 40 195 rows, more than the
               reference's Max_Seq_Anzahl 18000, PW:17, so the reference cannot hold it) with the CPU port
               oracle/libpworacle.so.  -> tests/golden/distributed_stress_round1.json, labelled PORT-made.

Only digests, score lines, dimensions and timings are committed.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from repeatresolver_amd import datagen as dg  # noqa: E402
import gen_fullscale as gf  # noqa: E402

IA = os.path.join(HERE, "_ref", "initial_aligner")
GOLD = os.path.join(ROOT, "tests", "golden")


def pipeline(work, threads):
    cfg = dg.CONFIGS["tree_default"]
    fx_path = os.path.join(GOLD, "pipeline_tree_default_round1.json")
    seq, _full, _starts, _cids, cut, _ = dg.simulate_dataset(cfg)
    reads = [r for r in cut if r is not None and len(r) >= cfg.min_aligned]
    tpath, rpath = os.path.join(work, "p_Template.fasta"), os.path.join(work, "p_Seq.fasta")
    with open(tpath, "wb") as f:
        f.write(b">\n" + dg.ASCII[seq].tobytes() + b"\n")
    dg.write_fasta(rpath, reads)
    fx = {"generator": "oracle/gen_pipeline_fixture.py --pipeline",
          "reference_builds": ["gcc -O2 InitialAligner.c -lpthread", "gcc -O2 -mcmodel=medium PW_ReAligner.c"],
          "workload": "tree_default through the pipeline: simulate_dataset -> cut reads >= min_aligned -> reference InitialAligner "
                      "(cut-off 0.30) -> reference PW_ReAligner, one round",
          "host_cpu": gf.host_cpu(), "reads": len(reads), "bases": int(sum(len(r) for r in reads)),
          "template": int(len(seq)), "bandwidth": 1000}
    msa = os.path.join(work, "p_MSA")
    if not os.path.exists(msa + ".done"):
        t0 = time.time()
        gf.log("[ia] reference InitialAligner on", len(reads), "reads,", threads, "threads")
        with open(os.path.join(work, "ia.stdout"), "wb") as so:
            subprocess.run([IA, "p_Template.fasta", "p_Seq.fasta", "-o", "p_MSA", "-s", "p_SeqClass", "-p", str(threads)],
                           cwd=work, check=True, stdout=so)
        fx["initial_aligner_seconds"] = round(time.time() - t0, 1)
        open(msa + ".done", "w").write(str(fx["initial_aligner_seconds"]))
    else:
        fx["initial_aligner_seconds"] = float(open(msa + ".done").read())
    cls = open(os.path.join(work, "p_SeqClass")).read().split()
    T = cls.count("r") if cls else None
    with open(msa, "rb") as f:
        first = f.readline()
    size = os.path.getsize(msa)
    T = size // len(first)
    assert size == T * len(first)
    fx.update(msa_rows=T, msa_columns=len(first) - 1, msa_sha256=gf.sha_file(msa),
              rejected_by_cutoff=cls.count("l"))
    gf.log("[ia] MSA", T, "x", len(first) - 1, fx["msa_sha256"], "in", fx["initial_aligner_seconds"], "s")
    json.dump(fx, open(fx_path, "w"), indent=1)

    out = os.path.join(work, "p_out.msa")
    recs, said = [], []
    plain_log = gf.log

    def keep(*a):                                          # the reference's "Rows .." and initial score lines pass through gf.log
        said.append(" ".join(str(x) for x in a))
        plain_log(*a)
    gf.log = keep
    try:
        gf.watch_reference(msa, out, T, 1, recs.append, "[pw_ref]")
    finally:
        gf.log = plain_log
    fx["rows_line"] = next(l.split("[pw_ref] ", 1)[1] for l in said if l.startswith("[pw_ref] Rows "))
    fx["initial_score_line"] = next(l.split("[pw_ref] initial ", 1)[1] for l in said if l.startswith("[pw_ref] initial "))
    r = recs[0]
    r["round"] = 1
    fx["round1"] = r
    init = [l for l in open(os.path.join(work, "ia.stdout"), errors="replace").read().splitlines()
            if l.startswith(("template length", "read count", "errorcutoff"))]
    fx["initial_aligner_stdout"] = init
    json.dump(fx, open(fx_path, "w"), indent=1)
    gf.log("[pipeline] fixture written", fx_path)


def config3(work):
    fx_path = os.path.join(GOLD, "distributed_stress_round1.json")
    m = dg.make_msa("distributed_stress")
    T, W = m.shape
    path = os.path.join(work, "c3_in.msa")
    dg.write_msa(path, m)
    del m
    sha = gf.sha_file(path)
    lib = C.CDLL(os.path.join(HERE, "libpworacle.so"))
    lib.pwo_load.restype = C.c_void_p
    lib.pwo_load.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    for f in ("pwo_trim", "pwo_compact", "pwo_realign_round"):
        getattr(lib, f).argtypes = [C.c_void_p]
        getattr(lib, f).restype = None
    lib.pwo_total_score.argtypes = [C.c_void_p]
    lib.pwo_total_score.restype = C.c_uint64
    lib.pwo_cells.argtypes = [C.c_void_p]
    lib.pwo_cells.restype = C.c_uint64
    lib.pwo_width.argtypes = [C.c_void_p]
    lib.pwo_export.argtypes = [C.c_void_p, C.c_void_p]
    lib.pwo_export.restype = None
    err = C.create_string_buffer(256)
    s = lib.pwo_load(path.encode(), 1000, err, 256)
    assert s, err.value
    lib.pwo_trim(s)
    lib.pwo_compact(s)
    W0 = lib.pwo_width(s)
    best = lib.pwo_total_score(s)
    gf.log("[config3]", T, "x", W, "->", W0, "initial score", best)
    t0 = time.time()
    lib.pwo_realign_round(s)
    dt = time.time() - t0
    tot = lib.pwo_total_score(s)
    Wk = lib.pwo_width(s)
    buf = np.empty((T, Wk), dtype=np.uint8)
    lib.pwo_export(s, buf.ctypes.data_as(C.c_void_p))
    outp = os.path.join(work, "c3_round1.msa")
    dg.write_msa(outp, buf)
    del buf
    fx = {"generator": "oracle/gen_pipeline_fixture.py --config3",
          "made_by": "PORT (oracle/pw_oracle.c, itself pinned to the reference by tests/golden/*.in.gz); the reference cannot hold "
                     "more than 18000 rows (PW:17)",
          "workload": "distributed_stress (truth-stacked, dg.make_msa)", "bandwidth": 1000, "host_cpu": gf.host_cpu(),
          "input_rows": T, "input_columns": W, "input_sha256": sha, "columns_after_trim": W0,
          "initial_score": int(best),
          "round1": {"score": int(tot), "improved": bool(tot < best), "rows": T, "columns": Wk,
                     "output_sha256": gf.sha_file(outp), "cells": int(lib.pwo_cells(s)), "port_seconds": round(dt, 1)}}
    json.dump(fx, open(fx_path, "w"), indent=1)
    gf.log("[config3] fixture written", fx)
    os.remove(outp)
    os.remove(path)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pipeline", action="store_true")
    ap.add_argument("--config3", action="store_true")
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--work", default="/tmp/pipefx")
    a = ap.parse_args()
    os.makedirs(a.work, exist_ok=True)
    if a.pipeline:
        pipeline(a.work, a.threads)
    if a.config3:
        config3(a.work)
