import ctypes
import gzip
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        return json.load(f)["cases"]


def golden_input(name) -> bytes:
    with gzip.open(os.path.join(GOLDEN, name + ".in.gz"), "rb") as f:
        return f.read()


def golden_output(name):
    p = os.path.join(GOLDEN, name + ".out.gz")
    if not os.path.exists(p):
        return None
    with gzip.open(p, "rb") as f:
        return f.read()


class Oracle:
    """ctypes view of oracle/libpworacle.so (the CPU restatement; checker only)."""

    def __init__(self):
        odir = os.path.join(ROOT, "oracle")
        subprocess.run(["make", "-C", odir, "port"], check=True, stdout=subprocess.DEVNULL)
        self.bin = os.path.join(odir, "pw_oracle")
        lib = ctypes.CDLL(os.path.join(odir, "libpworacle.so"))
        vp, ci, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64
        lib.pwo_create.restype = vp
        lib.pwo_create.argtypes = [ci, ci, ctypes.c_char_p, ci]
        lib.pwo_destroy.argtypes = [vp]
        for n in ("pwo_rows", "pwo_width", "pwo_check_tallies", "pwo_dbg_L", "pwo_dbg_W_at_fill", "pwo_dbg_entry"):
            getattr(lib, n).restype = ci
            getattr(lib, n).argtypes = [vp]
        lib.pwo_row_length.restype = ci
        lib.pwo_row_length.argtypes = [vp, ci]
        lib.pwo_row_columns.restype = ci
        lib.pwo_row_columns.argtypes = [vp, ci, ctypes.POINTER(ci), ci]
        lib.pwo_cells.restype = u64
        lib.pwo_cells.argtypes = [vp]
        lib.pwo_trim.argtypes = [vp]
        lib.pwo_compact.argtypes = [vp]
        lib.pwo_realign_row.restype = ci
        lib.pwo_realign_row.argtypes = [vp, ci]
        lib.pwo_realign_round.argtypes = [vp]
        lib.pwo_total_score.restype = u64
        lib.pwo_total_score.argtypes = [vp]
        lib.pwo_export.argtypes = [vp, ctypes.c_char_p]
        lib.pwo_dbg_way.restype = ctypes.POINTER(ci)
        lib.pwo_dbg_way.argtypes = [vp]
        lib.pwo_dbg_seq.restype = ctypes.POINTER(ctypes.c_ubyte)
        lib.pwo_dbg_seq.argtypes = [vp]
        lib.pwo_dbg_tallies.restype = ctypes.POINTER(u64)
        lib.pwo_dbg_tallies.argtypes = [vp]
        lib.pwo_dbg_newcol.restype = ctypes.POINTER(ci)
        lib.pwo_dbg_newcol.argtypes = [vp]
        lib.pwo_dbg_newins.restype = ctypes.POINTER(ctypes.c_ubyte)
        lib.pwo_dbg_newins.argtypes = [vp]
        lib.pwo_dbg_M.restype = u64
        lib.pwo_dbg_M.argtypes = [vp, ci, ci]
        self.lib = lib

    def run_cli(self, inp: bytes, bandwidth: int, tmpdir, extra=()):
        ip = os.path.join(tmpdir, "in.msa")
        op = os.path.join(tmpdir, "out.msa")
        with open(ip, "wb") as f:
            f.write(inp)
        if os.path.exists(op):
            os.remove(op)
        p = subprocess.run([self.bin, ip, "-o", op, "-b", str(bandwidth), *extra], capture_output=True)
        out = open(op, "rb").read() if os.path.exists(op) else None
        return p.returncode, out, p.stdout.decode("latin1").splitlines()

    def create(self, rows, bandwidth):
        """rows: list of equal-length bytes."""
        T, W = len(rows), len(rows[0])
        h = self.lib.pwo_create(T, W, b"".join(rows), bandwidth)
        assert h, "oracle rejected the input"
        return h

    def row_columns(self, h, k, cap=40000):
        buf = (ctypes.c_int * cap)()
        n = self.lib.pwo_row_columns(h, k, buf, cap)
        assert 0 <= n <= cap
        return list(buf[:n])

    def export(self, h) -> list:
        T, W = self.lib.pwo_rows(h), self.lib.pwo_width(h)
        buf = ctypes.create_string_buffer(T * W + 1)
        self.lib.pwo_export(h, buf)
        raw = buf.raw[:T * W]
        return [raw[i * W:(i + 1) * W] for i in range(T)]


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


def split_rows(data: bytes):
    rows = data.split(b"\n")
    assert rows[-1] == b""
    return rows[:-1]
