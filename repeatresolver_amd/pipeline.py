"""The front of the RepeatResolver pipeline as the reference chains it (RepeatResolver.c:150-230, README run order):
DataSimulator -> ReadCutter -> InitialAligner -> PW_ReAligner.  `initial_msa` produces what PW_ReAligner is really fed:
the simulated reads, cut to their repeat part, aligned into the template by the InitialAligner (on the GPU, include/pia.h)
and stacked by Building_MSA (IA:553-663)."""
import os
import shutil
import tempfile
import time

from . import datagen as dg
from .initial_aligner import InitialAligner


def initial_msa(cfg: dg.SimConfig, device: int = 0, cutoff: float = 0.30):
    """Returns (rows, info): rows = the lines of the `<ds>_MSA` file InitialAligner writes for the data set `cfg` describes
    (reads whose repeat part is shorter than cfg.min_aligned bases are left out: ReadCutter's mapping would not find them)."""
    t0 = time.time()
    seq, _full, _starts, _cids, cut, _ = dg.simulate_dataset(cfg)
    templ = dg.ASCII[seq].tobytes()
    reads = [dg.ASCII[r].tobytes() for r in cut if r is not None and len(r) >= cfg.min_aligned]
    t1 = time.time()
    g = InitialAligner(templ, device=device)
    place, dist = g.align(reads)
    st = g.stats()
    t2 = time.time()
    tmp = tempfile.mkdtemp(prefix="pia_msa_")
    try:
        msa_path, cls_path = os.path.join(tmp, "MSA"), os.path.join(tmp, "SeqClass")
        g.build_msa(msa_path, cls_path, reads, place, dist, cutoff)
        with open(msa_path, "rb") as f:
            rows = f.read().split(b"\n")
        with open(cls_path) as f:
            classes = f.read().split()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        g.close()
    if rows and rows[-1] == b"":
        rows.pop()
    info = {"reads": len(reads), "bases": sum(len(r) for r in reads), "template": len(templ), "rows": len(rows),
            "rejected_by_cutoff": classes.count("l"), "cells": st["cells"], "align_ms": st["last_align_ms"],
            "simulate_s": round(t1 - t0, 1), "align_s": round(t2 - t1, 2), "build_msa_s": round(time.time() - t2, 1)}
    return rows, info
