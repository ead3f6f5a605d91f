"""-m gpu: the HIP path, called through the C ABI (ctypes on libpwr.so / the PW_ReAligner CLI),
against (1) the committed reference fixtures and (2) the CPU oracle, row by row."""
import os

import pytest

from conftest import golden_cases, golden_input, golden_output, split_rows

pytestmark = pytest.mark.gpu
CASES = golden_cases()


def _score_lines(lines):
    return [l for l in lines if l.startswith(("OverallScore", "Rows ", "bandwidth"))]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_cli_matches_reference_fixture(case, tmp_path):
    from repeatresolver_amd.realigner import run_file
    ip, op = str(tmp_path / "in.msa"), str(tmp_path / "out.msa")
    with open(ip, "wb") as f:
        f.write(golden_input(case["name"]))
    rc, lines = run_file(ip, op, bandwidth=case["bandwidth"])
    assert rc == case["exit_code"], lines
    assert _score_lines(lines) == case["stdout"]
    exp = golden_output(case["name"])
    assert os.path.exists(op) == case["wrote_output"]
    if exp is not None:
        assert open(op, "rb").read() == exp


STEP_CASES = [("toy_a_b1000", 1000, 2), ("toy_a_b50", 50, 2), ("tiny_b10", 10, 3), ("tiny_b2", 2, 3),
              ("lowcov_b300", 300, 3), ("edge_shift", 12, 2), ("deep_b200", 200, 1),
              ("holes_b300", 300, 2), ("edge_inner_blanks", 8, 3)]                 # rows with blank runs between their bases


def _row_by_row(name, bw, rounds, oracle, **opts):
    """Every single realignment: same Way, same entry column, same new placement, same MSA."""
    from repeatresolver_amd.realigner import PWReAligner
    rows = name if isinstance(name, list) else split_rows(golden_input(name))
    force64 = opts.pop("force64", None)
    g = PWReAligner(rows, bandwidth=bw, **opts)
    if force64 is not None:
        g.set_option("force64", force64)
    g.trim_ends()
    lib = oracle.lib
    h = oracle.create(rows, bw)
    lib.pwo_trim(h)
    assert g.dims() == (lib.pwo_rows(h), lib.pwo_width(h))
    assert g.total_score() == lib.pwo_total_score(h)
    assert g.export_rows() == oracle.export(h)
    T = len(rows)
    for rnd in range(rounds):
        for k in range(T):
            assert lib.pwo_realign_row(h, k) == 0
            g.realign_row(k)
            L = lib.pwo_dbg_L(h)
            if L > 0:
                d = g.debug_last_job()
                way = [lib.pwo_dbg_way(h)[x] for x in range(L)]
                exp_new = [(lib.pwo_dbg_newcol(h)[x] << 1) | lib.pwo_dbg_newins(h)[x] for x in range(L)]
                assert d["L"] == L, (rnd, k)
                assert d["way"] == way, (rnd, k)
                assert d["W"] == lib.pwo_dbg_W_at_fill(h), (rnd, k)
                assert d["entry"] == lib.pwo_dbg_entry(h), (rnd, k)
                assert d["newcol"] == exp_new, (rnd, k)
            if k % 7 == 0 or k == T - 1:
                lib.pwo_compact(h)
                assert g.export_rows() == oracle.export(h), (rnd, k)
        assert g.total_score() == lib.pwo_total_score(h)
    if force64:
        st = g.stats()
        assert st["rows_wide"] == st["rows_committed"] > 0
    lib.pwo_destroy(h)
    g.close()


@pytest.mark.parametrize("fill", [4, 3], ids=["v3", "v2"])
@pytest.mark.parametrize("name,bw,rounds", STEP_CASES, ids=[c[0] for c in STEP_CASES])
def test_row_by_row_against_oracle(name, bw, rounds, fill, oracle):
    _row_by_row(name, bw, rounds, oracle, fill=fill)


@pytest.mark.parametrize("fill", [4, 3], ids=["v3", "v2"])
@pytest.mark.parametrize("waves", [3, 4, 5, 8, 9, 17])
@pytest.mark.parametrize("name,bw,rounds", [STEP_CASES[0], STEP_CASES[4], STEP_CASES[6]], ids=["toy_a_b1000", "lowcov_b300", "deep_b200"])
def test_wave_geometries_row_by_row(name, bw, rounds, waves, fill, oracle):
    """All macro-strip widths (3/4/5/8/9/17 waves per DP) of the two wave-pipeline fill kernels (17 is k_fill_v3 only;
    k_fill_v2 falls back to 9 there)."""
    _row_by_row(name, bw, rounds, oracle, fill=fill, waves=waves)


@pytest.mark.parametrize("fill", [4, 3], ids=["v3", "v2"])
@pytest.mark.parametrize("window", [1, 3, 64])
def test_batched_rounds_match_sequential_oracle(window, fill, oracle):
    """Speculative batches of any size must give the row-sequential result (commit in row order,
    stale speculations recomputed)."""
    from repeatresolver_amd.realigner import PWReAligner
    for name, bw, rounds in (("toy_a_b50", 50, 3), ("lowcov_b300", 300, 3), ("deep_b200", 200, 2)):
        rows = split_rows(golden_input(name))
        g = PWReAligner(rows, bandwidth=bw, window=window, fill=fill)
        g.trim_ends()
        h = oracle.create(rows, bw)
        oracle.lib.pwo_trim(h)
        for _ in range(rounds):
            g.realign_round()
            oracle.lib.pwo_realign_round(h)
            assert g.total_score() == oracle.lib.pwo_total_score(h)
            assert g.export_rows() == oracle.export(h)
        st = g.stats()
        assert st["cells_reference"] == oracle.lib.pwo_cells(h)
        oracle.lib.pwo_destroy(h)
        g.close()


@pytest.mark.parametrize("name,bw,rounds", STEP_CASES, ids=[c[0] for c in STEP_CASES])
def test_force64_row_by_row(name, bw, rounds, oracle):
    """The 64-bit fallback fill (k_fill64, PW:30 / PW:271 arithmetic) forced on ordinary inputs: every realignment against the oracle."""
    _row_by_row(name, bw, rounds, oracle, force64=1)


def test_wide_scores_take_the_64bit_fill(oracle):
    """A stack so deep (180 000 rows) that the gather cannot prove the 32-bit range of the wave pipeline
    (largest tally x (2B + 4096) > 2^30): those jobs go through k_fill64 instead of being refused.  The reference binary
    cannot hold more than 18 000 rows (PW:17), so the checker is the CPU restatement alone."""
    import numpy as np
    from repeatresolver_amd.realigner import PWReAligner
    rng = np.random.default_rng(5)
    T, W = 180000, 36
    tmpl = rng.integers(0, 4, W)
    m = np.tile(tmpl, (T, 1))
    sub = rng.random((T, W)) < 0.05
    m[sub] = rng.integers(0, 4, int(sub.sum()))
    txt = np.frombuffer(b"acgt", dtype=np.uint8)[m]
    txt[rng.random((T, W)) < 0.06] = ord("-")
    txt[:, 0] = np.frombuffer(b"acgt", dtype=np.uint8)[tmpl[0]]            # a base at both ends of every row
    txt[:, -1] = np.frombuffer(b"acgt", dtype=np.uint8)[tmpl[-1]]
    rows = [bytes(r) for r in txt]
    g = PWReAligner(rows, bandwidth=1000, window=4)
    g.trim_ends()
    lib = oracle.lib
    h = oracle.create(rows, 1000)
    lib.pwo_trim(h)
    assert g.total_score() == lib.pwo_total_score(h)
    for k in range(6):                                   # one at a time: Way, entry, placement
        assert lib.pwo_realign_row(h, k) == 0
        g.realign_row(k)
        L = lib.pwo_dbg_L(h)
        d = g.debug_last_job()
        assert d["L"] == L and d["entry"] == lib.pwo_dbg_entry(h), k
        assert d["newcol"] == [(lib.pwo_dbg_newcol(h)[x] << 1) | lib.pwo_dbg_newins(h)[x] for x in range(L)], k
    g.realign_rows(6, 30)                                # speculative batches of wide jobs
    for k in range(6, 36):
        assert lib.pwo_realign_row(h, k) == 0
    lib.pwo_compact(h)
    assert g.dims() == (T, lib.pwo_width(h))
    for k in list(range(36)) + [T - 1]:
        assert g.debug_row_columns(k) == oracle.row_columns(h, k), k
    assert g.total_score() == lib.pwo_total_score(h)
    st = g.stats()
    assert st["cells_reference"] == lib.pwo_cells(h)
    assert st["rows_wide"] == st["rows_committed"] == 36             # every one of them went through k_fill64
    lib.pwo_destroy(h)
    g.close()


@pytest.mark.parametrize("window", [1, 3])
def test_stalled_fill_is_repeated_by_the_one_workgroup_kernel(window, oracle):
    """k_fill_v3's waves wait for each other across work-groups; a wave that waits too long (GPU shared / oversubscribed)
    gives its job up.  That must cost time only: the job and the next batches are filled by k_fill_v2 (Hdr::fallback).
    The test hook makes the first job of the next k_fill_v3 launch stall at once (the launches behind it are gated off by
    the fall-back, so one stall is what is seen)."""
    from repeatresolver_amd.realigner import PWReAligner
    name, bw = "lowcov_b300", 300
    rows = split_rows(golden_input(name))
    g = PWReAligner(rows, bandwidth=bw, window=window)
    g.trim_ends()
    g.total_score()
    h = oracle.create(rows, bw)
    oracle.lib.pwo_trim(h)
    g.realign_rows(0, 5)
    g.set_option("stall_test", 1)
    g.realign_rows(5, len(rows) - 5)                  # one stall, then 64 batches of k_fill_v2, then k_fill_v3 again
    oracle.lib.pwo_realign_round(h)
    assert g.total_score() == oracle.lib.pwo_total_score(h)
    assert g.export_rows() == oracle.export(h)
    assert g.stats()["stalls"] == 1
    for _ in range(3):                                # (more than 64 batches: k_fill_v3 is back)
        g.realign_round()
        oracle.lib.pwo_realign_round(h)
    oracle.lib.pwo_compact(h)
    assert g.export_rows() == oracle.export(h)
    assert g.stats()["stalls"] == 1
    oracle.lib.pwo_destroy(h)
    g.close()


@pytest.mark.parametrize("evcap", [0, 2])
def test_commit_renumbering_paths(evcap, oracle):
    """The commit renumbers the columns from the list of opened / emptied columns when they are few, by a pass over the
    width otherwise; the hook sets where one gives way to the other (0: always the pass; 2: mixed)."""
    from repeatresolver_amd.realigner import PWReAligner
    for name, bw, rounds in (("lowcov_b300", 300, 3), ("toy_a_b50", 50, 2), ("holes_b300", 300, 2)):
        rows = split_rows(golden_input(name))
        g = PWReAligner(rows, bandwidth=bw, window=3)
        g.set_option("evcap", evcap)
        g.trim_ends()
        h = oracle.create(rows, bw)
        oracle.lib.pwo_trim(h)
        for _ in range(rounds):
            g.realign_round()
            oracle.lib.pwo_realign_round(h)
            assert g.total_score() == oracle.lib.pwo_total_score(h)
            assert g.export_rows() == oracle.export(h)
        oracle.lib.pwo_destroy(h)
        g.close()


def test_commit_ahead_of_a_stale_row(oracle):
    """Many short rows scattered over a wide MSA: batches regularly hold a stale row followed by a valid one whose band
    interval is disjoint from it; that one commits ahead (the two commute) and the result is still the
    row-sequential one."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    cfg = dg.SimConfig(kind="Tree", copies=6, coverage=12, difference=0.01, repeat_len=6000, flank=1500,
                       length_scale=0.03, min_aligned=60, seed=31)
    rows = [bytes(r) for r in dg.build_msa(dg.simulate(cfg))]
    lib = oracle.lib
    for window in (4, 16):
        g = PWReAligner(rows, bandwidth=100, window=window)
        g.trim_ends()
        h = oracle.create(rows, 100)
        lib.pwo_trim(h)
        for _ in range(2):
            g.realign_round()
            lib.pwo_realign_round(h)
            assert g.total_score() == lib.pwo_total_score(h)
            assert g.export_rows() == oracle.export(h)
        st = g.stats()
        assert st["rows_ahead"] > 0, st
        assert st["cells_reference"] == lib.pwo_cells(h)
        lib.pwo_destroy(h)
        g.close()


def test_launch_tag_wraparound(oracle):
    """The mailbox words of k_fill_v3 and the hand-over words of k_trace_par carry a launch counter; when it wraps
    the arrays are cleared and counting restarts.  Start both counters just below their limits."""
    from repeatresolver_amd.realigner import PWReAligner
    name, bw = "lowcov_b300", 300
    rows = split_rows(golden_input(name))
    g = PWReAligner(rows, bandwidth=bw, window=3)
    g.trim_ends()
    g.total_score()                                  # first device call: allocates, counters at 0
    g.set_option("fill_epoch", (1 << 15) - 5)
    g.set_option("trace_epoch", (1 << 14) - 7)
    h = oracle.create(rows, bw)
    oracle.lib.pwo_trim(h)
    for _ in range(2):
        g.realign_round()
        oracle.lib.pwo_realign_round(h)
        assert g.total_score() == oracle.lib.pwo_total_score(h)
        assert g.export_rows() == oracle.export(h)
    oracle.lib.pwo_destroy(h)
    g.close()


def test_seeded_round_parity_and_invariants(oracle):
    """Fresh seeded input (not a fixture): two whole rounds through pwr_realign_round."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    cfg = dg.SimConfig(kind="Tree", copies=6, coverage=10, difference=0.01, repeat_len=2000, flank=600,
                       length_scale=0.12, min_aligned=150, seed=77)
    rows = [bytes(r) for r in dg.build_msa(dg.simulate(cfg))]
    g = PWReAligner(rows, bandwidth=400)
    g.trim_ends()
    lib = oracle.lib
    h = oracle.create(rows, 400)
    lib.pwo_trim(h)
    before = [r.replace(b"-", b"").replace(b" ", b"") for r in g.export_rows()]
    prev = g.total_score()
    for _ in range(2):
        g.realign_round()
        lib.pwo_realign_round(h)
        s = g.total_score()
        assert s == lib.pwo_total_score(h)
        assert s <= prev
        prev = s
        out = g.export_rows()
        assert out == oracle.export(h)
        # invariants of SURVEY 8a: base order preserved; rows are blank* (base|-)* blank*
        assert [r.replace(b"-", b"").replace(b" ", b"") for r in out] == before
        for r in out:
            core = r.strip(b" ")
            assert b" " not in core and (core == b"" or (core[:1] != b"-" and core[-1:] != b"-"))
    st = g.stats()
    assert st["cells_reference"] == lib.pwo_cells(h)
    lib.pwo_destroy(h)
    g.close()


def test_sections_on_gpu(oracle):
    """configs[3] in small: Window.py-style column sections, each realigned on the GPU on its own,
    must equal the oracle run on the same section (parity is per slice)."""
    from repeatresolver_amd.sharding import realign_sections
    from repeatresolver_amd.window import slice_sections
    from test_window_sharding import _oracle_worker
    rows = split_rows(golden_input("toy_a_b1000"))
    W = len(rows[0])
    secs = slice_sections(rows, [0, W // 3, 2 * W // 3, W])
    got = realign_sections(secs, bandwidth=200, max_rounds=2)          # the three sections run concurrently
    assert got == realign_sections(secs, bandwidth=200, max_rounds=2, concurrent=1)
    for p, sec in enumerate(secs):
        exp, _ = _oracle_worker(sec, 200, 0, 2)
        assert got[p] == exp


@pytest.mark.parametrize("bw", [2000, 1500, 1024, 1])
def test_wide_and_degenerate_bandwidths(bw, oracle):
    """Bandwidths above 1000 use the 9-wave / 4-columns-per-lane geometry; 1 is the smallest band."""
    from repeatresolver_amd.realigner import PWReAligner
    rows = split_rows(golden_input("toy_a_b1000"))
    g = PWReAligner(rows, bandwidth=bw)
    g.trim_ends()
    h = oracle.create(rows, bw)
    oracle.lib.pwo_trim(h)
    for _ in range(2):
        g.realign_round()
        oracle.lib.pwo_realign_round(h)
        assert g.total_score() == oracle.lib.pwo_total_score(h)
        assert g.export_rows() == oracle.export(h)
    oracle.lib.pwo_destroy(h)
    g.close()


def test_capacity_regrow_and_out_of_order_rows(oracle):
    """Tight allocation (slack 0) forces the device arrays to be regrown while columns are being
    opened; rows are realigned in an arbitrary order through pwr_realign_row."""
    from repeatresolver_amd.realigner import PWReAligner
    rows = split_rows(golden_input("lowcov_b300"))
    g = PWReAligner(rows, bandwidth=300, slack=0)
    g.trim_ends()
    h = oracle.create(rows, 300)
    oracle.lib.pwo_trim(h)
    T = len(rows)
    order = [(7 * i + 3) % T for i in range(T)] * 2
    for k in order:
        g.realign_row(k)
        assert oracle.lib.pwo_realign_row(h, k) == 0
    assert g.total_score() == oracle.lib.pwo_total_score(h)
    assert g.export_rows() == oracle.export(h)
    for _ in range(2):
        g.realign_round()
        oracle.lib.pwo_realign_round(h)
    assert g.total_score() == oracle.lib.pwo_total_score(h)
    assert g.export_rows() == oracle.export(h)
    oracle.lib.pwo_destroy(h)
    g.close()


def test_rows_without_bases_and_unsupported_states():
    """(rows with blanks between their bases are supported: fixtures holes_b300 / edge_inner_blanks)"""
    from repeatresolver_amd.realigner import PWReAligner, PwrError
    rows = [b"acgt-acgtacg", b"------------", b"ac-tgacgtacg", b"            ", b"-cgtgacgta--"]
    g = PWReAligner(rows, bandwidth=6, window=4)
    g.trim_ends()
    g.realign_round()                     # rows 1 and 3 have no bases: PW:1488
    out = g.export_rows()
    assert out[1].strip() == b"" and out[3].strip() == b""
    g.close()
    g = PWReAligner([b"--acgt--", b"acgtacgt"], bandwidth=6)   # not trimmed: '-' outside the bases
    with pytest.raises(PwrError) as e:
        g.realign_round()
    assert e.value.code == -7
    g.close()


def test_slabs_over_rows_without_bases(oracle):
    """The k loop runs over the rows that HAVE bases (a row without any is done wherever it stands, PW:1488; the sections of
    a Window.py cut are half made of such rows): slabs that begin, end or consist of empty rows, single empty rows, and the
    split-round calls over them leave the MSA the reference's round leaves."""
    from repeatresolver_amd.realigner import PWReAligner
    rows = split_rows(golden_input("toy_a_b1000"))
    blank = b" " * len(rows[0])
    T0 = len(rows)
    empty = {0, 1, 7, 8, 9, 20, T0 - 1}
    rows = [blank if k in empty else r for k, r in enumerate(rows)]
    lib = oracle.lib
    h = oracle.create(rows, 200)
    lib.pwo_trim(h)
    exp = []
    for rnd in range(2):
        lib.pwo_realign_round(h)
        exp.append((lib.pwo_total_score(h), oracle.export(h)))
    cells = lib.pwo_cells(h)
    lib.pwo_destroy(h)
    for mode in ("round", "slabs", "rows", "split"):
        g = PWReAligner(rows, bandwidth=200, window=3)
        g.trim_ends()
        for rnd in range(2):
            if mode == "round":
                g.realign_round()
            elif mode == "slabs":
                for k0, n in ((0, 2), (2, 5), (7, 3), (10, 10), (20, 1), (21, T0 - 21)):
                    g.realign_rows(k0, n)
            elif mode == "rows":
                for k in range(T0):
                    g.realign_row(k)
            else:
                import torch
                for k0, n in ((0, 2), (2, 6), (8, 2), (10, T0 - 10)):
                    g.split_begin(k0, n, 0, 1)
                    slot, per_rank = g.split_slot_bytes()
                    buf = torch.zeros(slot * per_rank, dtype=torch.uint8, device="cuda")
                    left, guard = n, 4 * n + 8
                    while left > 0 and guard > 0:
                        g.split_stage(buf.data_ptr())
                        left = g.split_commit(buf.data_ptr())
                        guard -= 1
                    assert left <= 0
            assert g.total_score() == exp[rnd][0], (mode, rnd)
            assert g.export_rows() == exp[rnd][1], (mode, rnd)
        st = g.stats()
        assert st["cells_reference"] == cells, mode
        assert st["rows_committed"] == 2 * (T0 - len(empty)), mode
        g.close()


def test_medium_properties_and_kernel_cross_check():
    """A shape too large for the oracle to follow row by row in the test budget (2.3 k rows x 37 k
    columns, 8*10^9 cells per round): size-independent properties, and the two fill kernels and several
    batch sizes against each other."""
    import numpy as np
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    rows = [bytes(r) for r in dg.make_msa("tree_medium")]
    ref = None
    for fill, window, waves in ((4, 8, 17), (4, 8, 5), (4, 4, 9), (3, 8, 5), (4, 1, 5), (3, 8, 9), (3, 8, 4)):
        g = PWReAligner(rows, bandwidth=1000, fill=fill, window=window, waves=waves)
        g.trim_ends()
        before = [r.replace(b"-", b"").replace(b" ", b"") for r in g.export_rows()]
        s0 = g.total_score()
        g.realign_round()
        s1 = g.total_score()
        out = g.export_rows()
        st = g.stats()
        g.close()
        assert s1 < s0
        assert [r.replace(b"-", b"").replace(b" ", b"") for r in out] == before          # bases and their order
        m = np.frombuffer(b"".join(out), dtype=np.uint8).reshape(len(out), -1)
        cov = (m != 0x20).sum(0).astype(np.int64)
        tot = 0
        for ch in b"ACGT-":                                                             # recount of PW:864-892
            nb = (m == ch).sum(0).astype(np.int64)
            tot += int((nb * (cov - nb)).sum())
        assert tot == s1
        assert (np.isin(m, list(b"ACGT")).sum(0) > 0).all()                             # no base-less column left
        for r in out[::97]:
            core = r.strip(b" ")
            assert b" " not in core and core[:1] != b"-" and core[-1:] != b"-"
        assert st["rows_committed"] == sum(1 for b in before if b)
        if ref is None:
            ref = (s1, out, st["cells_reference"])
        else:
            assert (s1, out, st["cells_reference"]) == ref


def test_snapshot_is_the_file_the_reference_would_write(oracle):
    """pwr_snapshot_*: the image of MMA_Auslesen's file (PW:1556-1598) taken in stream order -- the realignments that follow
    do not change it --, one at a time; the drop-in's writer thread hands exactly these bytes to the file (the CLI fixtures
    compare the files themselves)."""
    from repeatresolver_amd.realigner import PWReAligner, PwrError
    rows = split_rows(golden_input("toy_b_b1000"))
    g = PWReAligner(rows, bandwidth=1000)
    g.trim_ends()
    for rnd in range(2):
        g.realign_round()
        want = b"".join(r + b"\n" for r in g.export_rows())
        sn = g.snapshot_begin()
        with pytest.raises(PwrError) as e:
            g.snapshot_begin()                              # one at a time
        assert e.value.code == -1
        g.realign_rows(0, min(8, len(rows)))                # the state moves on at once; the image is the one taken
        assert g.snapshot_wait(sn) == want
    # ... and against the oracle after a further round: the same bytes the reference's file holds
    h = oracle.create(rows, 1000)
    oracle.lib.pwo_trim(h)
    g2 = PWReAligner(rows, bandwidth=1000)
    g2.trim_ends()
    g2.realign_round()
    oracle.lib.pwo_realign_round(h)
    oracle.lib.pwo_compact(h)
    assert g2.snapshot_wait(g2.snapshot_begin()) == b"".join(r + b"\n" for r in oracle.export(h))
    oracle.lib.pwo_destroy(h)
    g.close()
    g2.close()
