"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/pwr.h declares (no compute calls: there is no GPU in the build container)."""
import os
import re
import subprocess

from conftest import ROOT


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "repeatresolver_amd", "csrc"), "all"], check=True,
                   stdout=subprocess.DEVNULL)


def test_library_exports_every_declared_symbol():
    _build()
    from repeatresolver_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "pwr.h")).read()
    declared = set(re.findall(r"\b(pwr_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_host_text_path_needs_no_gpu(tmp_path):
    """create / trim / score / export before the first device call run on the host (PW:93-241,
    PW:459-645): compare with the oracle's start-up state."""
    _build()
    from conftest import Oracle, golden_input, split_rows
    from repeatresolver_amd.realigner import PWReAligner
    o = Oracle()
    for name, bw in (("toy_a_b1000", 1000), ("edge_mixed_case", 8), ("ia_toy_b1000", 1000)):
        rows = split_rows(golden_input(name))
        g = PWReAligner(rows, bandwidth=bw)
        h = o.create(rows, bw)
        g.trim_ends()
        o.lib.pwo_trim(h)
        assert g.dims() == (o.lib.pwo_rows(h), o.lib.pwo_width(h))
        assert g.total_score() == o.lib.pwo_total_score(h)
        assert g.export_rows() == o.export(h)
        o.lib.pwo_destroy(h)
        g.close()


def test_cli_usage_and_missing_file(tmp_path):
    _build()
    cli = os.path.join(ROOT, "repeatresolver_amd", "csrc", "PW_ReAligner")
    p = subprocess.run([cli], capture_output=True)
    assert p.returncode == 0 and p.stdout == b"Usage: ./PW_ReAligner MApath\n"          # PW:1615
    p = subprocess.run([cli, str(tmp_path / "nope"), "-o", str(tmp_path / "o")], capture_output=True)
    assert p.returncode == 1                                                             # PW:121
    assert p.stdout.decode().splitlines()[-1] == "MA is missing."


def test_argument_and_input_errors(tmp_path):
    """Error behaviour of the boundary that needs no GPU (PW:121, PW:134, PW:14)."""
    _build()
    import ctypes
    from repeatresolver_amd import _lib
    from repeatresolver_amd.realigner import PWReAligner, PwrError
    import pytest
    with pytest.raises(PwrError) as e:
        PWReAligner([b"acgx", b"acgt"])                    # not in the alphabet of PW:165-222
    assert e.value.code == -4
    with pytest.raises(PwrError) as e:
        PWReAligner([b"acgt"], bandwidth=2001)             # Max_Bandwidth, PW:14
    assert e.value.code == -5
    with pytest.raises(PwrError):
        PWReAligner([b"acgt"], bandwidth=0)
    g = PWReAligner([b"acgt", b"ac-t"])                    # option ranges (no device call is made)
    for key, bad in (("window", 0), ("window", 129), ("fill", 0), ("fill", 1), ("fill", 2), ("fill", 5), ("threads", 256), ("waves", 7), ("waves", 16), ("nonsense", 1)):
        with pytest.raises(PwrError) as e:
            g.set_option(key, bad)
        assert e.value.code == -1, (key, bad)
    for key, good in (("window", 128), ("fill", 3), ("fill", 4), ("waves", 17), ("waves", 9), ("ptrace", 0), ("ptrace", 1)):
        g.set_option(key, good)
    g.close()
    lib = _lib.load()
    lib.pwr_read_msa_file.restype = ctypes.c_int
    lib.pwr_read_msa_file.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                      ctypes.POINTER(ctypes.POINTER(ctypes.c_ubyte)), ctypes.c_char_p, ctypes.c_size_t]
    T, W = ctypes.c_int(), ctypes.c_int()
    txt = ctypes.POINTER(ctypes.c_ubyte)()
    err = ctypes.create_string_buffer(256)

    def read(data):
        p = tmp_path / "m.msa"
        p.write_bytes(data)
        return lib.pwr_read_msa_file(str(p).encode(), ctypes.byref(T), ctypes.byref(W), ctypes.byref(txt), err, 256)
    assert read(b"acgt\nac-t\n") == 0 and (T.value, W.value) == (2, 4)
    assert read(b"acgt\nacgt") == -4                      # last line without newline, PW:134
    assert read(b"acgt\nacg\n") == -4                     # unequal lengths: refused
    assert read(b"") == -4
    assert lib.pwr_read_msa_file(str(tmp_path / "nope").encode(), ctypes.byref(T), ctypes.byref(W),
                                 ctypes.byref(txt), err, 256) == -4
    assert err.value == b"MA is missing."                  # PW:121
