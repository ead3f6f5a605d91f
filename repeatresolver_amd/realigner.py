"""Host-side mirror of the reference's PW_ReAligner interface on top of the C ABI (include/pwr.h).

The reference has no Python API; its interface is `./PW_ReAligner <MSA> [-o out] [-b bw]`
(PW_ReAligner.c:1610-1647) and, inside, the functions this class names its methods after.
All work happens in libpwr.so (HIP); nothing here computes."""
import ctypes
import os
import subprocess

from . import _lib


class PwrError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"{what}: {_lib.load().pwr_strerror(code).decode()} ({code})")
        self.code = code


def _check(code, what):
    if code != 0:
        raise PwrError(code, what)


class PWReAligner:
    """One MSA resident on one GPU.  rows: list of equal-length bytes over `acgtACGT-_ `."""

    def __init__(self, rows, bandwidth=1000, device=0, window=None, profile=False, fill=None, waves=None, slack=None, **options):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self.T = len(rows)
        width = len(rows[0]) if rows else 0
        if any(len(r) != width for r in rows):
            raise ValueError("all MSA rows must have the same length")
        _check(self._lib.pwr_create(ctypes.byref(self._h), self.T, width, b"".join(rows), bandwidth, device),
               "pwr_create")
        if window is not None:
            _check(self._lib.pwr_set_option(self._h, b"window", int(window)), "set window")
        if slack is not None:
            _check(self._lib.pwr_set_option(self._h, b"slack", int(slack)), "set slack")
        if waves is not None:
            _check(self._lib.pwr_set_option(self._h, b"waves", int(waves)), "set waves")
        if fill is not None:
            _check(self._lib.pwr_set_option(self._h, b"fill", int(fill)), "set fill")
        if profile:                                  # True / 1: HIP events around every fill launch; n > 1: around every n-th
            _check(self._lib.pwr_set_option(self._h, b"profile", int(profile)), "set profile")
        for key, value in options.items():           # any other knob of pwr_set_option (include/pwr.h), e.g. seg_rows
            _check(self._lib.pwr_set_option(self._h, key.encode(), int(value)), "set " + key)

    def set_option(self, key, value):
        """pwr_set_option (include/pwr.h); most knobs must be set before the first call that touches the device."""
        _check(self._lib.pwr_set_option(self._h, key.encode(), int(value)), "set " + key)

    def get_option(self, key):
        v = ctypes.c_long()
        _check(self._lib.pwr_get_option(self._h, key.encode(), ctypes.byref(v)), "get " + key)
        return v.value

    @classmethod
    def from_file(cls, path, **kw):
        with open(path, "rb") as f:
            data = f.read()
        rows = data.split(b"\n")
        if rows[-1] != b"":
            raise ValueError("last line is not newline-terminated")       # PW_ReAligner.c:134
        return cls(rows[:-1], **kw)

    def close(self):
        if self._h:
            self._lib.pwr_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trim_ends(self):                     # EntAlGapper, PW_ReAligner.c:459-645
        _check(self._lib.pwr_trim_ends(self._h), "pwr_trim_ends")

    def realign_row(self, k):                # Matrix_Filler(k), PW_ReAligner.c:1469-1531
        _check(self._lib.pwr_realign_row(self._h, k), "pwr_realign_row")

    def realign_round(self):                 # PW_ReAligner.c:1695-1737
        _check(self._lib.pwr_realign_round(self._h), "pwr_realign_round")

    def realign_rows(self, k0, n):          # rows k0..k0+n-1 of that loop (a slab of PW_ReAligner.c:1695)
        _check(self._lib.pwr_realign_rows(self._h, k0, n), "pwr_realign_rows")

    # ---- a slab with every batch's fills split over the GPUs of a node (include/pwr.h pwr_split_*; driver: intra_round.py)
    def split_begin(self, k0, n, rank, world):
        _check(self._lib.pwr_split_begin(self._h, k0, n, rank, world), "pwr_split_begin")

    def split_slot_bytes(self):
        sb, spr = ctypes.c_size_t(), ctypes.c_int()
        _check(self._lib.pwr_split_slot_bytes(self._h, ctypes.byref(sb), ctypes.byref(spr)), "pwr_split_slot_bytes")
        return sb.value, spr.value

    def split_stage(self, send_ptr):         # send_ptr: device address (int) of slots_per_rank * slot_bytes bytes
        _check(self._lib.pwr_split_stage(self._h, ctypes.c_void_p(send_ptr)), "pwr_split_stage")

    def split_commit(self, recv_ptr) -> int:  # recv_ptr: device address of the all-gathered records; returns rows left in the slab
        left = ctypes.c_int()
        _check(self._lib.pwr_split_commit(self._h, ctypes.c_void_p(recv_ptr), ctypes.byref(left)), "pwr_split_commit")
        return left.value

    def total_score(self) -> int:            # OverallScorePrint, PW_ReAligner.c:933-963
        v = ctypes.c_uint64()
        _check(self._lib.pwr_total_score(self._h, ctypes.byref(v)), "pwr_total_score")
        return v.value

    def dims(self):
        t, w = ctypes.c_int(), ctypes.c_int()
        _check(self._lib.pwr_dims(self._h, ctypes.byref(t), ctypes.byref(w)), "pwr_dims")
        return t.value, w.value

    def export_rows(self):                   # MMA_Auslesen, PW_ReAligner.c:1556-1598
        t, w = self.dims()
        buf = ctypes.create_string_buffer(t * w + 1)
        _check(self._lib.pwr_export_rows(self._h, buf, t * w), "pwr_export_rows")
        raw = buf.raw[:t * w]
        return [raw[i * w:(i + 1) * w] for i in range(t)]

    def snapshot_begin(self):                # MMA_Auslesen's file image, taken in stream order and copied out on a stream of its own
        sn = ctypes.c_void_p()
        _check(self._lib.pwr_snapshot_begin(self._h, ctypes.byref(sn)), "pwr_snapshot_begin")
        return sn

    def snapshot_wait(self, sn) -> bytes:    # the image (rows of `width` characters, each followed by a newline); frees the snapshot
        img, n, t, w = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int()
        try:
            _check(self._lib.pwr_snapshot_wait(sn, ctypes.byref(img), ctypes.byref(n), ctypes.byref(t), ctypes.byref(w)), "pwr_snapshot_wait")
            assert n.value == t.value * (w.value + 1)
            return ctypes.string_at(img.value, n.value) if n.value else b""
        finally:
            self._lib.pwr_snapshot_free(sn)

    def stats(self):
        s = _lib.PwrStats()
        _check(self._lib.pwr_get_stats(self._h, ctypes.byref(s)), "pwr_get_stats")
        d = {f: getattr(s, f) for f, _ in s._fields_}
        d["reject_reason"] = list(d["reject_reason"])
        return d

    def reset_stats(self):
        _check(self._lib.pwr_reset_stats(self._h), "pwr_reset_stats")

    def debug_last_job(self, cap=40000):
        L, e, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        way = (ctypes.c_int * cap)()
        nc = (ctypes.c_int * cap)()
        _check(self._lib.pwr_debug_last_job(self._h, ctypes.byref(L), ctypes.byref(e), ctypes.byref(w), way, nc, cap),
               "pwr_debug_last_job")
        n = L.value
        return {"L": n, "entry": e.value, "W": w.value, "way": list(way[:n]), "newcol": list(nc[:n]),
                "rounds": self._lib.pwr_debug_rounds(self._h)}

    def debug_row_columns(self, k, cap=40000):
        """Column ordinals of the bases of row k (tests)."""
        buf = (ctypes.c_int * cap)()
        n = self._lib.pwr_debug_row_columns(self._h, k, buf, cap)
        if n < 0:
            raise PwrError(n, "pwr_debug_row_columns")
        return list(buf[:n])

    def debug_fill_clock(self):
        mhz, us = ctypes.c_double(), ctypes.c_double()
        _check(self._lib.pwr_debug_fill_clock(self._h, ctypes.byref(mhz), ctypes.byref(us)), "pwr_debug_fill_clock")
        return mhz.value, us.value


def write_msa(path, rows):
    with open(path, "wb") as f:
        f.write(b"\n".join(rows) + b"\n")


def run_file(in_path, out_path="MSAreal", bandwidth=1000, device=0, max_rounds=-1):
    """pwr_run_file through the C ABI; returns (exit_code, stdout lines)."""
    cli = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "PW_ReAligner")
    if not os.path.exists(cli):
        raise RuntimeError(f"{cli} is missing: build it first (make -C repeatresolver_amd/csrc)")
    args = [cli, in_path, "-o", out_path, "-b", str(bandwidth), "-g", str(device)]
    if max_rounds >= 0:
        args += ["-r", str(max_rounds)]
    p = subprocess.run(args, capture_output=True)
    return p.returncode, p.stdout.decode("latin1").splitlines()
