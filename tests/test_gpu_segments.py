"""-m gpu: the segmented fill (a DP filled as segments side by side, each warmed up from a free start and CHECKED against
its predecessor), the chunked traceback without a hand-over chain, and the two advisor findings of round 2
(commit ahead of a stale row whose interval has moved; rows committed ahead across a regrow).  Everything is compared with
the CPU oracle (oracle/, pinned to the compiled reference by tests/golden)."""
import numpy as np
import pytest

from conftest import golden_input, split_rows
from test_gpu_parity import _row_by_row

pytestmark = pytest.mark.gpu

SEG_CASES = [("toy_b_b1000", 1000, 1), ("toy_a_b1000", 1000, 2), ("toy_a_b50", 50, 2), ("lowcov_b300", 300, 2),
             ("deep_b200", 200, 1), ("holes_b300", 300, 2)]


@pytest.mark.parametrize("seg_rows,warm_pct,src_start", [(128, 200, 1), (64, 300, 0), (256, 150, 1), (128, 110, 1)])
@pytest.mark.parametrize("name,bw,rounds", SEG_CASES, ids=[c[0] for c in SEG_CASES])
def test_segmented_fill_row_by_row(name, bw, rounds, seg_rows, warm_pct, src_start, oracle):
    """Every realignment of the fixtures with their fills cut into small segments: same Way, entry column, placement, MSA --
    whether a warm-up starts from the column of the base before it alone (src_start 1, the default) or from the free start
    of PW:265 (0), and with a warm-up of 1.1 bandwidths that only the former can get away with (what it cannot is caught
    by the check and repeated)."""
    _row_by_row(name, bw, rounds, oracle, seg_rows=seg_rows, seg_max=64, warm_pct=warm_pct, src_start=src_start, seg_budget=0, seg_balance=seg_rows == 128)


@pytest.mark.parametrize("budget,minrows,smax", [(200, 64, 256), (48, 16, 256), (400, 32, 256), (7, 64, 5)])
@pytest.mark.parametrize("name,bw,rounds", SEG_CASES, ids=[c[0] for c in SEG_CASES])
def test_budgeted_segments_row_by_row(name, bw, rounds, budget, minrows, smax, oracle):
    """The same with the segments of a batch dealt from a budget by the rows' lengths (the default plan): finer than
    seg_rows when there is room (down to 16 own rows), coarser when the budget is small, capped per job."""
    _row_by_row(name, bw, rounds, oracle, seg_rows=128, seg_max=smax, warm_pct=200, seg_budget=budget, seg_minrows=minrows)


def test_steered_warm_up_settles_and_changes_nothing(oracle):
    """The length of the warm-ups follows the failures of the check (Hdr::warm_cur: a step down per segmented fill that
    passes, twenty up per failure, between warm_min_pct and warm_pct).  With the lower bound far too short it must come
    down, run into failures, go up again -- and the MSA is the reference's whatever it does."""
    from repeatresolver_amd.realigner import PWReAligner
    rows = split_rows(golden_input("toy_b_b1000"))
    lib = oracle.lib
    res = {}
    for adapt in (0, 1):
        g = PWReAligner(rows, bandwidth=1000, window=2, seg_rows=128, seg_max=64, warm_pct=200, warm_min_pct=10, warm_adapt=adapt)
        g.trim_ends()
        h = oracle.create(rows, 1000)
        lib.pwo_trim(h)
        seen = set()
        for rnd in range(2):
            for k0 in range(0, len(rows), 8):
                g.realign_rows(k0, min(8, len(rows) - k0))
                seen.add(g.get_option("warm_now"))
            lib.pwo_realign_round(h)
            assert g.total_score() == lib.pwo_total_score(h)
            assert g.export_rows() == oracle.export(h)
        st = g.stats()
        assert st["cells_reference"] == lib.pwo_cells(h)
        res[adapt] = (st["seg_fails"], st["cells_computed"], seen)
        lib.pwo_destroy(h)
        g.close()
    assert res[0][0] == 0 and res[0][2] == {200}, res[0]           # fixed: the full length, no failures
    assert res[1][0] > 0 and min(res[1][2]) < 140 and len(res[1][2]) > 3, res[1]
    assert res[1][1] < res[0][1]                                     # shorter warm-ups: fewer cells computed


def test_rows_whose_check_failed_warm_up_longer_from_then_on(oracle):
    """Failures of the check are a property of the row (measured at full scale: a row that failed once fails again in
    46 % of its later fills, any row in 8 %), so a row that fails gets `hard_up_pm` per mille of the bandwidth on top of
    the steered length from then on ("hard_rows" 1, default; 2: only counted; 0: round 3).  Same MSA as the reference's
    after every round whatever is marked."""
    from repeatresolver_amd.realigner import PWReAligner
    rows = split_rows(golden_input("toy_b_b1000"))
    lib = oracle.lib
    h = oracle.create(rows, 1000)
    lib.pwo_trim(h)
    exp = []
    for rnd in range(3):
        lib.pwo_realign_round(h)
        exp.append((lib.pwo_total_score(h), oracle.export(h)))
    lib.pwo_destroy(h)
    seen = {}
    for mode in (0, 2, 1):
        g = PWReAligner(rows, bandwidth=1000, window=2, seg_rows=128, seg_max=64, warm_pct=200, warm_min_pct=10, hard_rows=mode,
                        hard_up_pm=500)
        g.trim_ends()
        for rnd in range(3):
            g.realign_round()
            assert g.total_score() == exp[rnd][0], (mode, rnd)
            assert g.export_rows() == exp[rnd][1], (mode, rnd)
        seen[mode] = (g.stats()["seg_fails"], g.get_option("hard_marked"), g.get_option("hard_fills"), g.get_option("hard_refail"))
        g.close()
    assert seen[0][0] > 0 and seen[0][1:] == (0, 0, 0), seen
    assert seen[2][0] == seen[0][0] and seen[2][1] > 0 and seen[2][2] >= seen[2][3], seen   # counting changes nothing
    assert seen[1][1] > 0 and seen[1][2:] == (0, 0), seen                                     # (marked rows never go the short way)


@pytest.mark.parametrize("onewg", [0, 1])
@pytest.mark.parametrize("waves", [3, 4, 5, 8, 9, 17])
def test_segmented_fill_wave_geometries(waves, onewg, oracle):
    """... with every macro-strip width, the waves of a segment as work-groups of their own or as one (17 waves: only the former)."""
    _row_by_row("toy_b_b1000", 1000, 1, oracle, waves=waves, onewg=onewg, seg_rows=128, seg_max=64, warm_pct=200)
    _row_by_row("lowcov_b300", 300, 2, oracle, waves=waves, onewg=onewg, seg_rows=128, seg_max=64, warm_pct=200)


def test_nine_waves_with_strips_of_256_columns(oracle):
    """9 waves x 4 columns per lane at a bandwidth of 1000 ("wave_cols" 4): strips so wide that no wave has two of a DP row."""
    _row_by_row("toy_b_b1000", 1000, 1, oracle, waves=9, wave_cols=4, seg_rows=128, seg_max=64, warm_pct=200)
    _row_by_row("lowcov_b300", 300, 2, oracle, waves=9, wave_cols=4, seg_rows=128, seg_max=64, warm_pct=200)
    _row_by_row("toy_b_b1000", 1000, 1, oracle, waves=9, wave_cols=4, seg_rows=0)


def test_failed_segment_check_repeats_the_row_in_one_piece(oracle):
    """A warm-up that is far too short (a fifth of the bandwidth): the check of the segments' starts must catch it, the row
    is realigned again with its fill in one piece, and the results are the reference's all the same."""
    from repeatresolver_amd.realigner import PWReAligner
    for name, bw in (("toy_b_b1000", 1000), ("deep_b200", 200)):
        rows = split_rows(golden_input(name))
        for window, fold in ((1, 1), (4, 1), (4, 0)):            # (fold: the check's work-groups ride in the traceback's launch, or have their own)
            g = PWReAligner(rows, bandwidth=bw, window=window, seg_rows=128, seg_max=64, warm_pct=20, check_in_trace=fold)
            g.trim_ends()
            h = oracle.create(rows, bw)
            oracle.lib.pwo_trim(h)
            g.realign_round()
            oracle.lib.pwo_realign_round(h)
            assert g.total_score() == oracle.lib.pwo_total_score(h)
            assert g.export_rows() == oracle.export(h)
            st = g.stats()
            assert 0 < st["seg_fails"] <= st["seg_jobs"], st
            assert st["cells_reference"] == oracle.lib.pwo_cells(h)
            assert st["rows_committed"] == sum(1 for k in range(len(rows)) if oracle.lib.pwo_row_length(h, k) > 0)
            oracle.lib.pwo_destroy(h)
            g.close()


@pytest.mark.parametrize("budget,minrows", [(0, 64), (200, 64), (24, 32)])
def test_cells_computed_count_the_warm_up_rows(budget, minrows, oracle):
    """pwr_stats.cells_computed against the plan restated here: segment s of a row owns the rows [x_s, x_{s+1}),
    x_s = floor(L s / S) rounded down to 16, and warms up from the last multiple of 16 whose base lies at least warm_cols
    columns left of base x_s; every row of it costs min(B, W - anf) cells.  S: about seg_rows rows each (seg_budget 0), or the
    batch's budget of segments dealt by length -- a batch of one row (window 1) has it to itself --, none shorter than seg_minrows."""
    from repeatresolver_amd.realigner import PWReAligner
    rows = split_rows(golden_input("toy_b_b1000"))
    bw, H, sr, smax, wp = 1000, 500, 128, 64, 150
    warm_cols = bw * wp // 100 + 2
    g = PWReAligner(rows, bandwidth=bw, window=1, seg_rows=sr, seg_max=smax, warm_pct=wp, warm_adapt=0, seg_budget=budget, seg_minrows=minrows, seg_balance=0)   # (a fixed warm-up, equal own parts: the plan below)
    g.trim_ends()
    h = oracle.create(rows, bw)
    lib = oracle.lib
    lib.pwo_trim(h)
    exp_cells = exp_segs = exp_jobs = 0
    for k in range(40):
        assert lib.pwo_realign_row(h, k) == 0
        g.realign_row(k)
        L = lib.pwo_dbg_L(h)
        if L == 0:
            continue
        W = lib.pwo_dbg_W_at_fill(h)
        way = np.ctypeslib.as_array(lib.pwo_dbg_way(h), (L,)).astype(np.int64)
        cells = np.minimum(bw, W - np.maximum(0, way - H))
        S = max(1, min((L + sr // 2) // sr, smax, L // 128)) if budget == 0 else max(1, min(budget, L // max(16, minrows, 32), smax))
        xs = [(L * s // S) & ~15 for s in range(S)] + [L]
        for s in range(S):
            xb = 0
            if s > 0:
                cand = [c for c in range(0, xs[s], 16) if way[c] <= way[xs[s]] - warm_cols]
                xb = cand[-1] if cand else 0
            exp_cells += int(cells[xb:xs[s + 1]].sum())
        if S > 1:
            exp_jobs += 1
            exp_segs += S
    st = g.stats()
    assert st["seg_fails"] == 0 and st["rows_recomputed"] == 0
    assert (st["seg_jobs"], st["segs"]) == (exp_jobs, exp_segs)
    assert st["cells_computed"] == exp_cells
    assert st["cells_reference"] == lib.pwo_cells(h)
    lib.pwo_destroy(h)
    g.close()


@pytest.mark.parametrize("ptrace", [0, 1, 2])
def test_traceback_kernels_agree(ptrace, oracle):
    """k_trace_wp (one wave), k_trace_par (64 chunks handing over top-down) and k_trace_blk (one wave per 64 rows, no
    chain): the same placements row by row."""
    _row_by_row("toy_b_b1000", 1000, 1, oracle, ptrace=ptrace)
    _row_by_row("lowcov_b300", 300, 3, oracle, ptrace=ptrace)
    _row_by_row("toy_a_b50", 50, 2, oracle, ptrace=ptrace, seg_rows=128)


def test_commit_ahead_when_the_stale_rows_interval_has_moved(oracle):
    """Advisor, round 2: a row can be stale because an earlier commit of the batch opened or emptied columns inside its band
    interval; the interval its next gather takes is then no longer bounded by the columns the old one ended in, and a later
    row must be tested against the interval as it is NOW before it commits ahead.  Low coverage makes rows slide (columns
    open and empty by the dozen), a narrow band makes the margins matter, wide windows make commits ahead frequent; every
    round of every seed is compared with the oracle."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    lib = oracle.lib
    ahead = 0
    for seed in range(12):
        cfg = dg.SimConfig(kind="Distributed", copies=3, coverage=3, difference=0.03, repeat_len=2500, flank=600,
                           length_scale=0.02, min_aligned=40, seed=100 + seed)
        rows = [bytes(r) for r in dg.build_msa(dg.simulate(cfg))]
        bw = (6, 10, 16, 24)[seed % 4]
        g = PWReAligner(rows, bandwidth=bw, window=(16, 64)[seed % 2])
        g.trim_ends()
        h = oracle.create(rows, bw)
        lib.pwo_trim(h)
        for rnd in range(3):
            g.realign_round()
            lib.pwo_realign_round(h)
            assert g.total_score() == lib.pwo_total_score(h), (seed, rnd)
            assert g.export_rows() == oracle.export(h), (seed, rnd)
        st = g.stats()
        ahead += st["rows_ahead"]
        assert st["cells_reference"] == lib.pwo_cells(h), seed
        lib.pwo_destroy(h)
        g.close()
    assert ahead > 0


def test_rows_committed_ahead_survive_a_regrow(oracle):
    """Advisor, round 2: with no spare capacity (slack 0) the column arrays are regrown in the middle of a slab; the rows that
    had been committed ahead of order before that must not be realigned a second time (PW:1695 visits a row once per round)."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    cfg = dg.SimConfig(kind="Tree", copies=6, coverage=12, difference=0.01, repeat_len=6000, flank=1500,
                       length_scale=0.03, min_aligned=60, seed=31)
    rows = [bytes(r) for r in dg.build_msa(dg.simulate(cfg))]
    lib = oracle.lib
    g = PWReAligner(rows, bandwidth=100, window=16, slack=0)
    g.trim_ends()
    h = oracle.create(rows, 100)
    lib.pwo_trim(h)
    live = sum(1 for k in range(len(rows)) if lib.pwo_row_length(h, k) > 0)
    for rnd in range(3):
        g.reset_stats()
        g.realign_round()
        lib.pwo_realign_round(h)
        st = g.stats()
        assert st["rows_committed"] == live, (rnd, st)
        assert g.total_score() == lib.pwo_total_score(h)
        assert g.export_rows() == oracle.export(h)
    assert st["rows_ahead"] > 0
    lib.pwo_destroy(h)
    g.close()


@pytest.mark.parametrize("bw,opts", [(1500, {}), (1600, {}), (1500, {"seg_rows": 0}), (1500, {"seg_rows": 128, "warm_pct": 20}),
                                     (1000, {"waves": 9}), (1000, {"waves": 17}), (700, {"waves": 3}), (1000, {"onewg": 1})],
                         ids=["b1500", "b1600", "b1500_one_piece", "b1500_short_warmup", "b1000_w9", "b1000_w17", "b700_w3", "b1000_onewg"])
def test_rows_that_jump_further_than_the_band(bw, opts, oracle):
    """Found by scripts/dev/stress.py (in the round-2 library too): rows that still have a run of blanks between two of
    their segments wider than the whole band -- consecutive DP rows whose bands do not overlap (PW:285-295 carries the row
    minimum across).  With more macro-strips than the band is wide (bandwidths above 1024: 9 strips of 256 columns) a
    k_fill_v3 wave that was idle used to take over a strip the band had already jumped past.  Wide bands on a short MSA
    spread the rows out, so that the second round has such rows; every realignment of three rounds is compared."""
    from repeatresolver_amd import datagen as dg
    cfg = dg.SimConfig(kind="Tree", copies=5, coverage=14, difference=0.005, repeat_len=478, flank=676, length_scale=0.06,
                       min_aligned=102, seed=30351)
    rows = [bytes(r) for r in dg.build_msa(dg.simulate(cfg))]
    _row_by_row(rows, bw, 3, oracle, window=1, **opts)


def test_rows_picked_ahead_commute_and_commit(oracle):
    """"plan_ahead" (round 4): the speculative rows of a batch are picked by the batch before it -- first the rows among the next
    64 whose band interval keeps a wide gap from every uncommitted row before them.  They jump rows that are not even in their
    batch, and the MSA must still be the reference's after every round; with the option off (rows in order, round 3) the result
    is the same and nothing jumps."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    cfg = dg.SimConfig(kind="Tree", copies=12, coverage=15, difference=0.01, repeat_len=6000, flank=2000,
                       length_scale=0.12, min_aligned=200, seed=23)
    rows = [bytes(r) for r in dg.make_msa(cfg)]
    lib = oracle.lib
    h = oracle.create(rows, 300)
    lib.pwo_trim(h)
    exp = []
    for rnd in range(3):
        lib.pwo_realign_round(h)
        exp.append((lib.pwo_total_score(h), oracle.export(h)))
    cells = lib.pwo_cells(h)
    lib.pwo_destroy(h)
    jumped = {}
    for plan, window in ((1, 3), (0, 3), (1, 6)):
        g = PWReAligner(rows, bandwidth=300, window=window, plan_ahead=plan, plan_slack=200, plan_gate_rel=0)
        g.trim_ends()
        for rnd in range(3):
            g.realign_round()
            assert g.total_score() == exp[rnd][0], (plan, window, rnd)
            assert g.export_rows() == exp[rnd][1], (plan, window, rnd)
        st = g.stats()
        assert st["cells_reference"] == cells
        assert st["rows_committed"] == 3 * sum(1 for r in rows if any(c in b"acgtACGT" for c in r))
        jumped[(plan, window)] = (st["rows_jumped"], st["batches"])
        g.close()
    assert jumped[(0, 3)][0] == 0 and jumped[(1, 3)][0] > 0 and jumped[(1, 6)][0] > 0, jumped


def test_rows_commit_past_a_row_whose_check_failed(oracle):
    """A job whose segment check failed is repeated in the next batch; until then it is an uncommitted row like a stale one,
    and a later row of the batch that commutes with it may commit ahead of it (`fail_stops` 0, experimental; 1, the default: the
    batch ends at the failed job).  Far too short a warm-up makes many checks fail: same MSA as the reference's after every
    round either way, fewer batches with the rows that go past."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    cfg = dg.SimConfig(kind="Tree", copies=12, coverage=15, difference=0.01, repeat_len=6000, flank=2000,
                       length_scale=0.12, min_aligned=200, seed=29)
    rows = [bytes(r) for r in dg.make_msa(cfg)]
    lib = oracle.lib
    h = oracle.create(rows, 300)
    lib.pwo_trim(h)
    exp = []
    for rnd in range(2):
        lib.pwo_realign_round(h)
        exp.append((lib.pwo_total_score(h), oracle.export(h)))
    cells = lib.pwo_cells(h)
    lib.pwo_destroy(h)
    seen = {}
    for stops, plan in ((1, 0), (0, 0), (0, 1)):
        g = PWReAligner(rows, bandwidth=300, window=6, seg_rows=64, seg_max=64, warm_pct=25, warm_adapt=0, plan_ahead=plan,
                        plan_slack=200, plan_gate_rel=0, fail_stops=stops)
        g.trim_ends()
        for rnd in range(2):
            g.realign_round()
            assert g.total_score() == exp[rnd][0], (stops, plan, rnd)
            assert g.export_rows() == exp[rnd][1], (stops, plan, rnd)
        st = g.stats()
        assert st["cells_reference"] == cells
        assert st["seg_fails"] > 0, st
        seen[(stops, plan)] = (st["batches"], st["rows_ahead"], st["seg_fails"])
        g.close()
    assert seen[(0, 0)][0] < seen[(1, 0)][0] and seen[(0, 0)][1] > seen[(1, 0)][1], seen


def test_a_jump_that_does_not_hold_is_reported_not_swallowed(oracle):
    """The exactness of a jump rests on a gap that columns opened or emptied in between could in principle close; every row that
    was jumped checks at its gather that it has held (PWR_ERR_ORDER otherwise).  With the safety margins taken away (gap 0, no
    limit on the event rate) on MSAs whose columns open and empty by the dozen, a run either ends with that error or with the
    reference's MSA -- never with another MSA and no error."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner, PwrError
    lib = oracle.lib
    outcomes = {"exact": 0, "reported": 0}
    jumps = 0
    for seed in range(8):
        cfg = dg.SimConfig(kind="Distributed", copies=3, coverage=3, difference=0.03, repeat_len=2500, flank=600,
                           length_scale=0.02, min_aligned=40, seed=100 + seed)
        rows = [bytes(r) for r in dg.build_msa(dg.simulate(cfg))]
        bw = (6, 10, 16, 24)[seed % 4]
        g = PWReAligner(rows, bandwidth=bw, window=(4, 8)[seed % 2], plan_slack=0, plan_evrate_x100=100000000, plan_gate_rel=0)
        g.trim_ends()
        h = oracle.create(rows, bw)
        lib.pwo_trim(h)
        try:
            for rnd in range(3):
                g.realign_round()
                lib.pwo_realign_round(h)
                assert g.total_score() == lib.pwo_total_score(h), (seed, rnd)
                assert g.export_rows() == oracle.export(h), (seed, rnd)
            outcomes["exact"] += 1
            jumps += g.stats()["rows_jumped"]
        except PwrError as e:
            assert e.code == -10, e                                  # PWR_ERR_ORDER: said loudly
            outcomes["reported"] += 1
        lib.pwo_destroy(h)
        g.close()
    assert outcomes["exact"] + outcomes["reported"] == 8 and (jumps > 0 or outcomes["reported"] > 0), (outcomes, jumps)
