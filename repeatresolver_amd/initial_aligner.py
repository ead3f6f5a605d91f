"""Host-side mirror of the reference's InitialAligner (InitialAligner.c, "IA:") over the C ABI of include/pia.h.

The reference has no library interface here either: a process (`./InitialAligner template.fasta Seq.fasta -o msa -s
seqclass -e cutoff -p threads`, IA:667-770) around IntoAligner (IA:282-453) and Building_MSA (IA:553-663).  `InitialAligner`
below exposes the two halves; `run_files` is the drop-in binary.  All alignments run in libpwr.so's HIP kernels; there is
no CPU path."""
import ctypes
import os
import subprocess

import numpy as np

from . import _lib
from .realigner import PwrError

CLI_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "InitialAligner")


_LOWER = bytes.maketrans(b"ACGT", b"acgt")
_DROP = bytes(c for c in range(256) if c not in b"acgtACGT")


def _clean(seq: bytes) -> bytes:
    """what ReadingFasta keeps of a sequence (IA:184-201): aAcCgGtT, lower-cased"""
    return bytes(seq).translate(_LOWER, _DROP)


class InitialAligner:
    def __init__(self, template: bytes, device: int = 0):
        self._lib = _lib.load()
        self.template = _clean(template)
        self._h = ctypes.c_void_p()
        rc = self._lib.pia_create(ctypes.byref(self._h), self.template, len(self.template), device)
        if rc:
            raise PwrError(rc, self._lib.pwr_strerror(rc).decode())

    def close(self):
        if self._h:
            self._lib.pia_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def align(self, reads):
        """IntoAligner for every read: returns (placements, distances); placements[j][x] = template position of base x
        of read j or -1 (IA:420-446), distances[j] = Row[entry] (IA:333-352)."""
        reads = [_clean(r) for r in reads]
        n = len(reads)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(r) for r in reads])
        bases = b"".join(reads)
        align = np.empty(max(int(off[-1]), 1), dtype=np.int32)
        dist = np.empty(max(n, 1), dtype=np.int32)
        rc = self._lib.pia_align(self._h, n, bases, off.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                                 align.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), dist.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        if rc:
            raise PwrError(rc, self._lib.pwr_strerror(rc).decode())
        return [align[off[j]:off[j + 1]].copy() for j in range(n)], dist[:n].copy()

    def build_msa(self, msa_path, class_path, reads, placements, distances, cutoff=0.30):
        """Building_MSA (IA:553-663) on alignments from align()."""
        reads = [_clean(r) for r in reads]
        n = len(reads)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(r) for r in reads])
        al = np.concatenate([np.asarray(p, dtype=np.int32) for p in placements] + [np.zeros(1, np.int32)])
        ds = np.ascontiguousarray(np.asarray(list(distances) + [0], dtype=np.int32))
        rc = self._lib.pia_build_msa(os.fsencode(msa_path), os.fsencode(class_path), n, b"".join(reads),
                                     off.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                                     al.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ds.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                     float(cutoff), len(self.template))
        if rc:
            raise PwrError(rc, self._lib.pwr_strerror(rc).decode())

    def set_option(self, key: str, value: int):
        rc = self._lib.pia_set_option(self._h, key.encode(), int(value))
        if rc:
            raise PwrError(rc, self._lib.pwr_strerror(rc).decode())

    def stats(self):
        cells, ms = ctypes.c_uint64(), ctypes.c_double()
        self._lib.pia_get_stats(self._h, ctypes.byref(cells), ctypes.byref(ms))
        t = (ctypes.c_double * 6)()
        self._lib.pia_get_timing(self._h, t)
        return {"cells": cells.value, "fill_ms": ms.value,
                "last_align_ms": dict(zip(("total", "upload", "pass1", "pass2_and_traceback", "pass2_batches", "download"), t))}


def run_files(template_path, reads_path, msa_path=None, class_path=None, cutoff=None, device=None):
    """The drop-in binary with the reference's argv (IA:667-735); returns (exit code, stdout lines)."""
    if not os.path.exists(CLI_PATH):
        raise RuntimeError(f"{CLI_PATH} is missing: build it with `make -C repeatresolver_amd/csrc`")
    cmd = [CLI_PATH, str(template_path), str(reads_path)]
    if msa_path is not None:
        cmd += ["-o", str(msa_path)]
    if class_path is not None:
        cmd += ["-s", str(class_path)]
    if cutoff is not None:
        cmd += ["-e", repr(float(cutoff))]
    if device is not None:
        cmd += ["-g", str(device)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    return p.returncode, p.stdout.splitlines()
