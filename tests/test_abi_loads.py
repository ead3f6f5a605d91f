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
