"""-m gpu: the HIP MaxCorrelation (include/pmc.h) against the CPU restatement oracle/mc_oracle.c.  Floating
point: the significances must agree to 1e-9 (the tail sums are the same scheme, lgamma / exp / log10 differ by rounding
between device and host); zeros and the 98 + F saturations exactly where the oracle has them.  Parity with the REFERENCE
is unpinned (it needs GSL): see the oracle's header."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_output, split_rows
from test_mc_oracle import small_msa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mco():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libmcoracle.so"))
    lib.mco_maxcorrs.restype = ctypes.c_int
    lib.mco_maxcorrs.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]

    def run(rows, mincov):
        out = np.zeros(len(rows[0]) * 5)
        assert lib.mco_maxcorrs(len(rows), len(rows[0]), b"".join(rows), mincov, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
        return out
    return run


def _check(rows, mincov, mco):
    from repeatresolver_amd.max_correlation import last_timing, max_correlations
    got = max_correlations(rows, mincov)
    exp = mco(rows, mincov)
    assert np.array_equal(got == 0, exp == 0)
    assert np.allclose(got, exp, rtol=0, atol=1e-9), float(np.abs(got - exp).max())
    return got, last_timing()


@pytest.mark.parametrize("mincov", [4, 12, 30])
def test_small_msa(mincov, mco):
    got, t = _check(small_msa(), mincov, mco)
    if mincov == 12:
        assert (got > 3).sum() > 10 and t["pairs"] > 1000


@pytest.mark.parametrize("shape", [(63, 300), (64, 300), (65, 300), (129, 900), (40, 2200)])
def test_word_and_tile_boundaries(shape, mco):
    """row counts around the 64-bit words of the bit sets (sc = T / 64 + 1, MC:338), widths over several tiles"""
    T, W = shape
    _check(small_msa(seed=T + W, T=T, W=W), 10, mco)


def test_realigned_fixture_msas(mco):
    """what the pipeline feeds it: MSAreal files (the golden outputs of the realigner fixtures), blanks at the row ends"""
    n = 0
    for name in sorted(os.listdir(GOLDEN)):
        if not name.endswith(".out.gz"):
            continue
        rows = split_rows(golden_output(name[:-7]))
        if len(rows) < 12 or len(rows[0]) < 60:
            continue
        _check(rows, max(4, len(rows) // 3), mco)
        n += 1
    assert n >= 5


def test_saturated_significance(mco):
    """two perfectly linked variants in deep columns: the tail underflows 1e-99 and the value becomes 98 + F (MC:432)"""
    T, W = 900, 120
    rows = []
    for r in range(T):
        row = bytearray(b"a" * W)
        if r % 2:
            row[10] = ord("c"); row[70] = ord("g")
        rows.append(bytes(row))
    got, _ = _check(rows, 30, mco)
    assert got[10 * 5 + 1] == pytest.approx(99.0) and got[70 * 5 + 2] == pytest.approx(99.0)      # F = 1 for identical groups


def test_cli_writes_the_reference_file(tmp_path, mco):
    from repeatresolver_amd.max_correlation import run_file
    rows = small_msa(seed=8, T=90, W=400)
    (tmp_path / "MSAreal").write_bytes(b"\n".join(rows) + b"\n")
    rc, lines = run_file("MSAreal", mincov=12, cwd=str(tmp_path))
    assert rc == 0, lines
    assert "There are 90 sequences." in lines and "Siglength is 400." in lines and "MaxCorrsOf_MSAreal" in lines
    got = np.array([float(l) for l in (tmp_path / "MaxCorrsOf_MSAreal").read_text().split()])
    p = subprocess.run([os.path.join(ROOT, "oracle", "mc_oracle"), str(tmp_path / "MSAreal"), str(tmp_path / "exp"), "12"])
    assert p.returncode == 0
    exp_txt = (tmp_path / "exp").read_text().split()
    assert len(got) == 2000 == len(exp_txt)
    assert np.allclose(got, [float(v) for v in exp_txt], rtol=0, atol=1.5e-6)
    same = sum(a == b for a, b in zip((tmp_path / "MaxCorrsOf_MSAreal").read_text().split(), exp_txt))
    assert same >= 1990                                               # "%f": rounding may flip a last digit now and then
    rc, lines = run_file("nope", cwd=str(tmp_path))
    assert rc == 1 and lines[-1] == "MA is missing."                 # MC:283


def test_row_order_does_not_matter_at_scale():
    """A size-independent property on a realigned MSA too large for the CPU restatement (2 320 rows x 35 000 columns, 10^9
    pairs): every count is the size of an intersection of row sets, so shuffling the rows must not change a single bit of
    the result -- which also exercises the kernels' own reordering of the rows by first column and their word ranges."""
    import random
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.max_correlation import last_timing, max_correlations
    from repeatresolver_amd.realigner import PWReAligner
    rows = [bytes(r) for r in dg.make_msa("tree_medium")]
    g = PWReAligner(rows, bandwidth=1000)
    g.trim_ends()
    g.realign_round()
    rows = g.export_rows()
    g.close()
    a = max_correlations(rows, 30)
    t = last_timing()
    assert t["pairs"] > 5e8 and (a > 0).sum() > 10000 and a.max() > 20
    shuffled = rows[:]
    random.Random(4).shuffle(shuffled)
    b = max_correlations(shuffled, 30)
    assert np.array_equal(a, b)
    assert last_timing()["pairs"] == t["pairs"]
