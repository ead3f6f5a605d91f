"""The CPU restatement of MaxCorrelation (oracle/mc_oracle.c).  PARITY UNPINNED against the reference (it
needs GSL, which this image lacks): what can be pinned here is the hypergeometric tail, against scipy's independent
implementation, and the rest of the restatement against a literal numpy transcription of MC:745-837 on a small MSA."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def mco():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libmcoracle.so"))
    lib.mco_hyper_Q.restype = ctypes.c_double
    lib.mco_hyper_Q.argtypes = [ctypes.c_uint] * 4
    lib.mco_significance.restype = ctypes.c_double
    lib.mco_significance.argtypes = [ctypes.c_int] * 6
    lib.mco_maxcorrs.restype = ctypes.c_int
    lib.mco_maxcorrs.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    return lib


def test_hypergeometric_tail_against_scipy(mco):
    from scipy.stats import hypergeom
    rng = np.random.default_rng(1)
    for _ in range(4000):
        n1 = int(rng.integers(1, 600)); n2 = int(rng.integers(0, 600)); t = int(rng.integers(1, n1 + n2 + 1)); k = int(rng.integers(0, min(n1, t) + 1))
        q = mco.mco_hyper_Q(k, n1, n2, t)
        ref = hypergeom.sf(k, n1 + n2, n1, t)                       # P(X > k), gsl_cdf_hypergeometric_Q(k, n1, n2, t)
        assert q == pytest.approx(ref, rel=1e-10, abs=1e-300), (k, n1, n2, t)


def small_msa(seed=3, T=70, W=260):
    """rows with two linked variant positions every few columns, blanks at the row ends, some gaps"""
    rng = np.random.default_rng(seed)
    cons = rng.integers(0, 4, W)
    hap = rng.integers(0, 3, T)
    rows = []
    for r in range(T):
        a, b = int(rng.integers(0, W // 4)), int(W - rng.integers(0, W // 4))
        row = np.full(W, ord(" "), dtype=np.uint8)
        seq = cons.copy()
        for c in range(5, W, 11):
            if hap[r] == (c // 11) % 3:
                seq[c] = (cons[c] + 1) % 4                           # a variant shared by one haplotype
        noise = rng.random(W) < 0.03
        seq[noise] = rng.integers(0, 4, int(noise.sum()))
        row[a:b] = np.frombuffer(b"acgt", dtype=np.uint8)[seq[a:b]]
        gaps = (rng.random(W) < 0.04) & (np.arange(W) >= a) & (np.arange(W) < b)
        row[gaps] = ord("-")
        rows.append(row.tobytes())
    return rows


def literal_maxcorrs(rows, mincov, sig):
    """MC:745-837 with Python sets, sig = PositiveSignificance from the four counts and the two sizes"""
    T, W = len(rows), len(rows[0])
    code = {ord(c): k for k, c in enumerate("acgt-")}
    code.update({ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("_"): 4})
    G = [[set() for _ in range(5)] for _ in range(W)]
    LC = [set() for _ in range(W)]
    for r, row in enumerate(rows):
        for c, ch in enumerate(row):
            k = code.get(ch, 5)
            if k < 5:
                G[c][k].add(r); LC[c].add(r)
    out = np.zeros(W * 5)
    rel = lambda c, k: len(G[c][k]) > mincov // 4 and len(G[c][k]) < T
    for ii in range(W):
        baseno = sum(len(G[ii][k]) for k in range(4))
        for k in range(5):
            if not (rel(ii, k) and baseno > len(LC[ii]) // 2):
                continue
            for jj in range(ii + 20, W):
                cov = len(LC[ii] & LC[jj])
                if cov < mincov:
                    break
                for kk in range(5):
                    if rel(jj, kk):
                        z = sig(len(G[ii][k] & G[jj][kk]), cov, len(G[ii][k] & LC[jj]), len(G[jj][kk] & LC[ii]), len(G[ii][k]), len(G[jj][kk]))
                        out[ii * 5 + k] = max(out[ii * 5 + k], z); out[jj * 5 + kk] = max(out[jj * 5 + kk], z)
    return out


def test_restatement_against_literal_transcription(mco):
    rows = small_msa()
    T, W = len(rows), len(rows[0])
    got = np.zeros(W * 5)
    assert mco.mco_maxcorrs(T, W, b"".join(rows), 12, got.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
    exp = literal_maxcorrs(rows, 12, lambda *a: mco.mco_significance(*a))
    assert np.array_equal(got, exp)
    assert (got > 3).sum() > 10 and (got == 0).sum() > 10            # linked variants stand out, most variations do not exist


def test_significance_branches_against_exact_rationals(mco):
    """PositiveSignificance saturates in two steps (Z > 99 -> 99, MC:417; Z > 98 -> 98 + F, MC:432): a rounding difference in
    the tail right at those thresholds would flip a branch and move the value by up to one.  The tail of the restatement is
    compared here with the EXACT tail (Python integers: sum of C(n1, i) C(n2, t - i) over i > k, over C(n1 + n2, t)) for the
    counts whose -log10 lies nearest to 98 and 99 -- the branch taken must be the exact one, the value within 1e-9."""
    from math import comb
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    rng = np.random.default_rng(7)
    checked = near = 0
    for _ in range(60):
        n1 = int(rng.integers(150, 900)); n2 = int(rng.integers(150, 900)); t = int(rng.integers(120, n1 + n2 - 50))
        den = comb(n1 + n2, t)
        hi = min(n1, t)
        # exact upper tails P(X > k) for all k, from the top down
        tail = 0
        exact = {}
        for i in range(hi, -1, -1):
            exact[i] = tail                                          # P(X > i) * den
            tail += comb(n1, i) * comb(n2, t - i) if 0 <= t - i <= n2 else 0
        for k in range(hi):
            if exact[k] == 0:
                continue
            z_exact = -(Decimal(exact[k]) / Decimal(den)).log10()
            if not (Decimal(96) < z_exact < Decimal(101)):
                continue
            q = mco.mco_hyper_Q(k, n1, n2, t)
            z = -np.log10(q)
            assert abs(Decimal(float(z)) - z_exact) < Decimal("1e-9"), (k, n1, n2, t)
            assert (z > 99) == (z_exact > 99) and (z > 98.0) == (z_exact > 98), (k, n1, n2, t, float(z), z_exact)
            checked += 1
            near += abs(z_exact - 98) < Decimal("0.5") or abs(z_exact - 99) < Decimal("0.5")
            # the whole function on counts that produce this tail: schnitt - 1 = k, gr2 = n1, cov - gr2 = n2, gr1 = t
            zz = mco.mco_significance(k + 1, n1 + n2, t, n1, t + 5, n1 + 7)
            if z_exact > 98:
                f = 2.0 * (k + 1) / (2.0 * (k + 1) + (t + 5 - k - 1) + (n1 + 7 - k - 1))
                assert zz == pytest.approx(98.0 + f, abs=1e-12)
            else:
                assert zz == pytest.approx(float(z_exact), abs=1e-9)
    assert checked > 100 and near > 10
