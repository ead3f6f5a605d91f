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
    hdr = open(os.path.join(ROOT, "include", "pia.h")).read()
    declared = set(re.findall(r"\b(pia_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.PIA_EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    hdr = open(os.path.join(ROOT, "include", "pmc.h")).read()
    declared = set(re.findall(r"\b(pmc_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.PMC_EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_initial_aligner_host_side_needs_no_gpu(tmp_path):
    """FASTA reading (IA:66-262) and Building_MSA (IA:553-663) are host C: fed with the alignments the ORACLE computes,
    pia_build_msa must write the reference's files (tests/golden/ia_*); the CLI's usage and missing-file exits."""
    _build()
    import ctypes
    import gzip
    import json
    import numpy as np
    from conftest import GOLDEN
    from repeatresolver_amd import _lib
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, stdout=subprocess.DEVNULL)
    lib = _lib.load()
    ora = ctypes.CDLL(os.path.join(ROOT, "oracle", "libiaoracle.so"))
    ora.iao_align.restype = ctypes.c_long
    ora.iao_align.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                              ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    with open(os.path.join(GOLDEN, "ia_cases.json")) as f:
        cases = json.load(f)["cases"]
    for case in cases:
        t, r = tmp_path / "x_Template.fasta", tmp_path / "x_Seq.fasta"
        for kind, path in (("template", t), ("reads", r)):
            with gzip.open(os.path.join(GOLDEN, f"{case['name']}.{kind}.gz"), "rb") as f:
                path.write_bytes(f.read())
        tp, L2 = ctypes.c_void_p(), ctypes.c_int()
        assert lib.pia_read_template(str(t).encode(), ctypes.byref(tp), ctypes.byref(L2)) == 0
        templ = ctypes.string_at(tp, L2.value)
        n, bp, op = ctypes.c_int(), ctypes.c_void_p(), ctypes.c_void_p()
        assert lib.pia_read_fasta(str(r).encode(), ctypes.byref(n), ctypes.byref(bp), ctypes.byref(op)) == 0
        off = np.ctypeslib.as_array(ctypes.cast(op, ctypes.POINTER(ctypes.c_longlong)), shape=(n.value + 1,)).copy()
        bases = ctypes.string_at(bp, int(off[-1]))
        assert set(templ) <= set(b"acgt") and set(bases) <= set(b"acgt") and n.value == case["seqclass"].count("\n")
        align = np.empty(int(off[-1]), dtype=np.int32)
        dist = np.empty(n.value, dtype=np.int32)
        for j in range(n.value):
            rd = bases[off[j]:off[j + 1]]
            codes = ctypes.create_string_buffer(len(rd) * L2.value)
            a = (ctypes.c_int * len(rd))()
            dist[j] = ora.iao_align(rd, len(rd), templ, L2.value, a, None, codes)
            align[off[j]:off[j + 1]] = a
        msa, cls = tmp_path / "msa", tmp_path / "cls"
        assert lib.pia_build_msa(str(msa).encode(), str(cls).encode(), n.value, bases,
                                 off.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), align.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                 dist.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), case["cutoff"], L2.value) == 0
        assert cls.read_text() == case["seqclass"]
        with gzip.open(os.path.join(GOLDEN, f"{case['name']}.msa.gz"), "rb") as f:
            assert msa.read_bytes() == f.read()
    cli = os.path.join(ROOT, "repeatresolver_amd", "csrc", "InitialAligner")
    p = subprocess.run([cli], capture_output=True)
    assert p.returncode == 0 and p.stdout.startswith(b"Usage: ./InitialAligner_parallel template.fasta Seq.fasta\n")   # IA:272
    p = subprocess.run([cli, str(tmp_path / "nope_Template.fasta"), str(tmp_path / "x_Seq.fasta")], capture_output=True)
    assert p.returncode == 1                                                                                              # IA:226


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
    # the regular case goes through one read and row-parallel checks (1.8 GB at benchmark scale): same bytes, and anything
    # irregular -- here a file whose size fits but whose 501st line is cut in two -- still gets the line-by-line refusal
    big = b"".join(bytes(b"acgt-"[(i * 7 + j) % 5] for j in range(37)) + b"\n" for i in range(1000))
    assert read(big) == 0 and (T.value, W.value) == (1000, 37)
    assert bytes(txt[:1000 * 37]) == big.replace(b"\n", b"")
    bad = bytearray(big)
    bad[38 * 500 + 10] = ord("\n")
    bad[38 * 500 + 37] = ord("a")
    assert read(bytes(bad)) == -4 and b"line 501 has 10 characters" in err.value
    nul = bytearray(big)
    nul[38 * 3 + 5] = 0
    assert read(bytes(nul)) == -4
    assert lib.pwr_read_msa_file(str(tmp_path / "nope").encode(), ctypes.byref(T), ctypes.byref(W),
                                 ctypes.byref(txt), err, 256) == -4
    assert err.value == b"MA is missing."                  # PW:121
