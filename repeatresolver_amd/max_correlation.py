"""Host-side mirror of the reference's MaxCorrelation (MaxCorrelation.c, "MC:") over the C ABI of include/pmc.h: for every
variation (column, symbol) of a realigned MSA the largest significance of its co-occurrence with a variation at least 20
columns away (MC:745-837).  The pair loop runs in libpwr.so's HIP kernels; there is no CPU path."""
import ctypes
import os
import subprocess

import numpy as np

from . import _lib
from .realigner import PwrError

CLI_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "MaxCorrelation")


def max_correlations(rows, mincov: int = 30, device: int = 0) -> np.ndarray:
    """rows: equally long byte strings (the lines of MSAreal).  Returns MaxCorrs[width * 5] (MC:839-905)."""
    lib = _lib.load()
    T, W = len(rows), len(rows[0])
    if any(len(r) != W for r in rows):
        raise ValueError("rows of unequal length")
    out = np.zeros(W * 5, dtype=np.float64)
    rc = lib.pmc_maxcorrs(T, W, b"".join(rows), mincov, device, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    if rc:
        raise PwrError(rc, lib.pwr_strerror(rc).decode())
    return out


def last_timing():
    lib = _lib.load()
    t = (ctypes.c_double * 5)()
    lib.pmc_last_timing(t)
    return {"total_ms": t[0], "bits_ms": t[1], "ranges_ms": t[2], "pairs_ms": t[3], "pairs": int(t[4])}


def run_file(msa_path, mincov=None, cwd=None, device=None):
    """The drop-in binary with the reference's argv (MC:916-1020); writes MaxCorrsOf_<msa_path> relative to cwd."""
    if not os.path.exists(CLI_PATH):
        raise RuntimeError(f"{CLI_PATH} is missing: build it with `make -C repeatresolver_amd/csrc`")
    cmd = [CLI_PATH, str(msa_path)]
    if mincov is not None:
        cmd += ["-c", str(mincov)]
    if device is not None:
        cmd += ["-g", str(device)]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=cwd)
    return p.returncode, p.stdout.splitlines()
