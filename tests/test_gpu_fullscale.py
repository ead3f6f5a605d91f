"""-m gpu: the HIP path at BASELINE.json's full sizes and on the remaining configs.

configs[1] (Tree_1perc_30000kb, 13 510 rows x 136 477 columns) and configs[2] (Distributed, 200 copies, 60x:
40 195 rows x 149 140 columns) are far too large to follow to convergence with the CPU oracle inside a test, but the
oracle does ~30 full-size realignments per second: a PREFIX of round 1 is checked row by row (Way, entry column, new
placement) and then through speculative batches (pwr_realign_rows), comparing the placement of every realigned row,
the width, the total score and the reference's cell count.  configs[3]: Window.py sections of a realigned MSA, each
section on the GPU against the oracle on the same section.  configs[4]: a transposon-sized MSA to convergence
through the CLI against a fixture made with the compiled reference (tests/golden/transposon_like.json)."""
import hashlib
import json
import os

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _prefix_parity(workload, n_single, n_batched, oracle, fill_epoch=None, others=()):
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    rows = [bytes(r) for r in dg.make_msa(workload)] if isinstance(workload, str) else workload
    lib = oracle.lib
    g = PWReAligner(rows, bandwidth=1000)
    g.trim_ends()
    h = oracle.create(rows, 1000)
    T = len(rows)
    del rows
    lib.pwo_trim(h)
    assert g.dims() == (lib.pwo_rows(h), lib.pwo_width(h))
    assert g.total_score() == lib.pwo_total_score(h)
    # (1) one realignment at a time: every kernel's output against the oracle
    for k in range(n_single):
        assert lib.pwo_realign_row(h, k) == 0
        g.realign_row(k)
        L = lib.pwo_dbg_L(h)
        if L == 0:
            continue
        d = g.debug_last_job()
        assert d["L"] == L, k
        assert d["W"] == lib.pwo_dbg_W_at_fill(h), k
        assert d["way"] == [lib.pwo_dbg_way(h)[x] for x in range(L)], k
        assert d["entry"] == lib.pwo_dbg_entry(h), k
        assert d["newcol"] == [(lib.pwo_dbg_newcol(h)[x] << 1) | lib.pwo_dbg_newins(h)[x] for x in range(L)], k
    # (2) speculative batches, optionally across the wrap of the launch counter behind the mailbox tags
    if fill_epoch is not None:
        g.set_option("fill_epoch", fill_epoch)
    g.realign_rows(n_single, n_batched)
    for k in range(n_single, n_single + n_batched):
        assert lib.pwo_realign_row(h, k) == 0
    lib.pwo_compact(h)
    assert g.dims() == (T, lib.pwo_width(h))
    for k in list(range(n_single + n_batched)) + list(others):
        assert g.debug_row_columns(k) == oracle.row_columns(h, k), k
    st = g.stats()
    assert st["cells_reference"] == lib.pwo_cells(h)
    assert st["rows_committed"] == sum(1 for k in range(n_single + n_batched) if lib.pwo_row_length(h, k) > 0)
    assert g.total_score() == lib.pwo_total_score(h)            # recount of every column's tallies, PW:864-892
    lib.pwo_destroy(h)
    g.close()


def test_config2_tree_default_prefix_of_round_one(oracle):
    """BASELINE.json configs[1]: the benchmark MSA itself."""
    _prefix_parity("tree_default", 24, 136, oracle, others=(500, 5000, 13509))


def test_config2_pipeline_input_prefix_of_round_one(oracle):
    """The benchmark's default input: the same data set's reads aligned into the template by the GPU InitialAligner and
    stacked by Building_MSA (repeatresolver_amd/pipeline.py) -- what the reference pipeline feeds PW_ReAligner.  The
    InitialAligner's own parity is tests/test_gpu_initial_aligner.py; here three of its 13 594 placements are compared
    with the CPU restatement at full size, then the realigner runs the prefix check on its MSA."""
    import ctypes
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.initial_aligner import InitialAligner
    from repeatresolver_amd.pipeline import initial_msa
    from conftest import ROOT
    cfg = dg.CONFIGS["tree_default"]
    rows, info = initial_msa(cfg)
    assert info["rows"] == info["reads"] - info["rejected_by_cutoff"] == len(rows) and all(len(r) == len(rows[0]) for r in rows[::997])
    seq, _f, _s, _c, cut, _ = dg.simulate_dataset(cfg)
    templ = dg.ASCII[seq].tobytes()
    reads = [dg.ASCII[r].tobytes() for r in cut if r is not None and len(r) >= cfg.min_aligned]
    sample = [reads[j] for j in (0, len(reads) // 2, len(reads) - 1)]
    g = InitialAligner(templ)
    place, dist = g.align(sample)
    g.close()
    ora = ctypes.CDLL(os.path.join(ROOT, "oracle", "libiaoracle.so"))
    ora.iao_align.restype = ctypes.c_long
    ora.iao_align.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                              ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    for j, r in enumerate(sample):
        al = (ctypes.c_int * len(r))()
        codes = ctypes.create_string_buffer(len(r) * len(templ))
        assert ora.iao_align(r, len(r), templ, len(templ), al, None, codes) == int(dist[j])
        assert list(al) == list(place[j])
        # the row of the MSA holds exactly this read, in order
    assert info["rejected_by_cutoff"] == 0
    for j in range(0, len(reads), 1):                   # every row holds exactly its read, in order (Building_MSA, IA:602-655)
        assert rows[j].replace(b"-", b"") == reads[j], j
    _prefix_parity(rows, 16, 96, oracle, others=(700, 13000))


def test_config3_distributed_stress_prefix_and_epoch_wrap(oracle):
    """BASELINE.json configs[2]; one round of it is more than 2^15 batches, so the 15-bit launch counter of
    k_fill_v3's mailbox tags wraps inside a round: start it just below the wrap."""
    _prefix_parity("distributed_stress", 8, 112, oracle, fill_epoch=(1 << 15) - 30, others=(20000, 40194))


def _sha_rows(rows):
    h = hashlib.sha256()
    for r in rows:
        h.update(r)
        h.update(b"\n")
    return h.hexdigest()


def _rounds_fixture():
    with open(os.path.join(GOLDEN, "tree_default_rounds.json")) as f:
        fx = json.load(f)
    assert fx["rounds"], "tests/golden/tree_default_rounds.json holds no round yet (oracle/gen_fullscale.py)"
    return fx


def test_config2_every_round_against_the_reference():
    """BASELINE.json configs[1], "run to convergence, bit-exact check": tests/golden/tree_default_rounds.json holds, for every
    round the unmodified reference ran on the full-size MSA (13 510 x 136 477; oracle/gen_fullscale.py, 35-45 minutes of one
    CPU core per round), its score line and the sha256 of the file it had just rewritten (PW:1741).  The GPU path is followed
    through the same rounds: total score and exported text of every round, and the stop rule (PW:1742) when the fixture
    reaches the reference's last, non-improving round."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    fx = _rounds_fixture()
    rows = [bytes(r) for r in dg.make_msa("tree_default")]
    assert _sha_rows(rows) == fx["input_sha256"], "the seeded generator drifted"
    g = PWReAligner(rows, bandwidth=fx["bandwidth"])
    del rows
    g.trim_ends()
    best = g.total_score()
    for r in fx["rounds"]:
        g.realign_round()
        tot = g.total_score()
        assert tot == r["score"], (r["round"], tot, r["score"])
        assert (tot < best) == r["improved"], r["round"]
        if not r["improved"]:
            break                                              # the reference stops here and writes nothing (PW:1742)
        best = tot
        assert g.dims() == (r["rows"], r["columns"]), r["round"]
        assert _sha_rows(g.export_rows()) == r["output_sha256"], r["round"]
    st = g.stats()
    assert st["stalls"] == 0 and st["rows_wide"] == 0
    g.close()


def test_config2_cli_rounds_against_the_reference(tmp_path):
    """The same through the drop-in binary: `PW_ReAligner in -o out -r 2` -- score lines and the bytes of the file after the
    second round, both as the reference produced them."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import run_file
    fx = _rounds_fixture()
    n = min(2, len([r for r in fx["rounds"] if r["improved"]]))
    ip, op = str(tmp_path / "in.msa"), str(tmp_path / "out.msa")
    dg.write_msa(ip, dg.make_msa("tree_default"))
    rc, lines = run_file(ip, op, bandwidth=fx["bandwidth"], max_rounds=n)
    assert rc == 0
    got = [l for l in lines if l.startswith("OverallScore")]
    assert got[1:1 + n] == [r["score_line"] for r in fx["rounds"][:n]], got
    h = hashlib.sha256()
    with open(op, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    assert h.hexdigest() == fx["rounds"][n - 1]["output_sha256"]


def test_bench_input_is_the_references_own_msa_and_round_one(tmp_path):
    """The MSA `bench.py` runs by default (--input pipeline), pinned by the REFERENCE end to end
    (tests/golden/pipeline_tree_default_round1.json, oracle/gen_pipeline_fixture.py): the reads `pipeline.initial_msa` keeps were
    written as FASTA and aligned by the reference's own InitialAligner (an hour of six cores), its MSA then realigned for one
    round by the reference's PW_ReAligner (40 minutes).  Here: (1) the GPU InitialAligner + Building_MSA reproduce that MSA
    byte for byte at full size -- 13 594 reads, 4 * 10^12 cells: SURVEY N2 against the reference itself; (2) the drop-in binary
    reproduces the score lines and the bytes of the file the reference rewrote after round 1 (PW:1741)."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.pipeline import initial_msa
    from repeatresolver_amd.realigner import run_file, write_msa
    with open(os.path.join(GOLDEN, "pipeline_tree_default_round1.json")) as f:
        fx = json.load(f)
    rows, info = initial_msa(dg.CONFIGS["tree_default"])
    assert (info["reads"], info["bases"], info["template"]) == (fx["reads"], fx["bases"], fx["template"]), "the seeded generator drifted"
    assert (len(rows), len(rows[0])) == (fx["msa_rows"], fx["msa_columns"])
    assert _sha_rows(rows) == fx["msa_sha256"]
    ip, op = str(tmp_path / "in.msa"), str(tmp_path / "out.msa")
    write_msa(ip, rows)
    del rows
    rc, lines = run_file(ip, op, bandwidth=fx["bandwidth"], max_rounds=1)
    assert rc == 0
    r1 = fx["round1"]
    assert [l for l in lines if l.startswith("Rows ")] == [fx["rows_line"]]
    got = [l for l in lines if l.startswith("OverallScore")]
    assert got[:2] == [fx["initial_score_line"], r1["score_line"]], got
    assert r1["improved"]
    h = hashlib.sha256()
    with open(op, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    assert h.hexdigest() == r1["output_sha256"]
    assert os.path.getsize(op) == r1["rows"] * (r1["columns"] + 1)


def test_config3_one_full_round_against_the_port():
    """BASELINE.json configs[2] (Distributed, 200 copies, 60x: 40 195 rows -- more than the reference's Max_Seq_Anzahl 18000,
    PW:17, so the reference cannot hold it and the CPU port, itself pinned to the reference by tests/golden/*.in.gz, is the only
    possible checker): one FULL round, score, width, cell count and sha256 of the exported text against
    tests/golden/distributed_stress_round1.json (PORT-made, oracle/gen_pipeline_fixture.py --config3, 40 CPU-minutes)."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    with open(os.path.join(GOLDEN, "distributed_stress_round1.json")) as f:
        fx = json.load(f)
    rows = [bytes(r) for r in dg.make_msa("distributed_stress")]
    assert (len(rows), len(rows[0])) == (fx["input_rows"], fx["input_columns"])
    assert _sha_rows(rows) == fx["input_sha256"], "the seeded generator drifted"
    g = PWReAligner(rows, bandwidth=fx["bandwidth"])
    del rows
    g.trim_ends()
    assert g.dims() == (fx["input_rows"], fx["columns_after_trim"])
    assert g.total_score() == fx["initial_score"]
    g.realign_round()
    r1 = fx["round1"]
    assert g.total_score() == r1["score"]
    assert g.dims() == (r1["rows"], r1["columns"])
    st = g.stats()
    assert st["cells_reference"] == r1["cells"]
    assert st["stalls"] == 0 and st["rows_committed"] > 40000
    assert _sha_rows(g.export_rows()) == r1["output_sha256"]
    g.close()


def test_config4_sections_of_the_benchmark_msa(oracle):
    """BASELINE.json configs[3] at its real size: the benchmark MSA after one realignment round (checked against the
    reference's digest of that round) is cut at its Window.py boundaries (parts = 6, Window.py:41-60); all six sections are
    realigned side by side on the GPU for a slab of rows, and two of them are followed by the oracle row for row."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    from repeatresolver_amd.window import slice_sections, window_boundaries
    fx = _rounds_fixture()
    rows = [bytes(r) for r in dg.make_msa("tree_default")]
    g = PWReAligner(rows, bandwidth=1000)
    del rows
    g.trim_ends()
    g.realign_round()
    real = g.export_rows()
    g.close()
    assert _sha_rows(real) == fx["rounds"][0]["output_sha256"]
    bounds = window_boundaries(real, parts=6)
    assert len(bounds) == 7 and bounds == sorted(bounds) and bounds[0] > 0 and bounds[-1] < len(real[0])
    secs = slice_sections(real, bounds)
    del real
    T = len(secs[0])
    n = 100
    ctxs = []
    for sec in secs:                                       # six contexts side by side on one GPU, one stream each
        c = PWReAligner(sec, bandwidth=1000)
        c.trim_ends()
        c.total_score()
        ctxs.append(c)
    import threading
    ths = [threading.Thread(target=c.realign_rows, args=(0, n)) for c in ctxs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for p in (1, 4):                                       # the oracle on the same section, the same rows
        h = oracle.create(secs[p], 1000)
        oracle.lib.pwo_trim(h)
        for k in range(n):
            assert oracle.lib.pwo_realign_row(h, k) == 0
        oracle.lib.pwo_compact(h)
        assert ctxs[p].dims() == (T, oracle.lib.pwo_width(h)), p
        for k in list(range(n)) + [T - 1]:
            assert ctxs[p].debug_row_columns(k) == oracle.row_columns(h, k), (p, k)
        assert ctxs[p].total_score() == oracle.lib.pwo_total_score(h), p
        st = ctxs[p].stats()
        assert st["cells_reference"] == oracle.lib.pwo_cells(h), p
        oracle.lib.pwo_destroy(h)
    for p, c in enumerate(ctxs):
        st = c.stats()
        assert st["rows_committed"] == sum(1 for k in range(n) if any(ch in b"ACGT" for ch in secs[p][k])), p
        assert st["stalls"] == 0, p
        c.close()


def test_config4_window_sections_of_a_realigned_msa(oracle):
    """BASELINE.json configs[3] at a size the oracle can follow: Window.py boundaries (parts = 6) of a realigned MSA,
    the six sections realigned side by side on the GPU, each equal to the oracle on the same section."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    from repeatresolver_amd.sharding import realign_sections
    from repeatresolver_amd.window import merge_sections, slice_sections, window_boundaries
    from test_window_sharding import _oracle_worker
    cfg = dg.SimConfig(kind="Tree", copies=12, coverage=15, difference=0.01, repeat_len=6000, flank=2000,
                       length_scale=0.3, min_aligned=300, seed=21)
    rows = [bytes(r) for r in dg.make_msa(cfg)]
    g = PWReAligner(rows, bandwidth=1000)
    g.trim_ends()
    g.realign_round()
    real = g.export_rows()
    g.close()
    h = oracle.create(rows, 1000)
    oracle.lib.pwo_trim(h)
    oracle.lib.pwo_realign_round(h)
    oracle.lib.pwo_total_score(h)                      # compacts, as the reference does before it writes (PW:1741)
    assert real == oracle.export(h)
    oracle.lib.pwo_destroy(h)
    bounds = window_boundaries(real, parts=6)          # Window.py:41-60
    assert len(bounds) == 7 and bounds == sorted(bounds) and bounds[0] > 0 and bounds[-1] < len(real[0])
    secs = slice_sections(real, bounds)
    got = realign_sections(secs, bandwidth=1000, max_rounds=1, concurrent=6)
    for p, sec in enumerate(secs):
        exp, _ = _oracle_worker(sec, 1000, 0, 1)
        assert got[p] == exp, p
    merged = merge_sections(got)
    assert len(merged) == len(rows) and all(len(r) == len(merged[0]) for r in merged)


def test_config5_transposon_like_to_convergence(tmp_path):
    """BASELINE.json configs[4] stand-in (the Drosophila files are not available offline): a transposon-sized MSA run
    to convergence through the drop-in CLI.  Expected score lines and output digest were produced by the compiled,
    unmodified reference on the same input (oracle/gen_golden.py --transposon; 7 rounds, minutes of CPU)."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import run_file
    with open(os.path.join(GOLDEN, "transposon_like.json")) as f:
        fx = json.load(f)
    ip, op = str(tmp_path / "in.msa"), str(tmp_path / "out.msa")
    dg.write_msa(ip, dg.make_msa("transposon_like"))
    assert hashlib.sha256(open(ip, "rb").read()).hexdigest() == fx["input_sha256"], "the seeded generator drifted"
    rc, lines = run_file(ip, op, bandwidth=fx["bandwidth"])
    assert rc == fx["exit_code"]
    assert [l for l in lines if l.startswith(("OverallScore", "Rows ", "bandwidth"))] == fx["stdout"]
    assert hashlib.sha256(open(op, "rb").read()).hexdigest() == fx["output_sha256"]


def _hip_rank_main(rank, world, port, q):
    """One rank of the multi-GPU path with the HIP worker (both ranks on device 0: this box has one GPU)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import golden_input, split_rows
    from repeatresolver_amd.sharding import realign_sections
    from repeatresolver_amd.window import slice_sections
    rows = split_rows(golden_input("toy_b_b1000"))
    W = len(rows[0])
    secs = slice_sections(rows, [0, W // 4, W // 2, 3 * W // 4, W])
    out = realign_sections(secs, bandwidth=300, max_rounds=2, device=0)      # default worker: libpwr.so through the C ABI
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sections_sharded_over_two_ranks_hip_worker(oracle):
    """The N > 1 path with the product worker under a process group: sections dealt to 2 ranks, each realigns its
    sections on the GPU (two contexts side by side), all-gather of the text; every section equals the oracle's."""
    import multiprocessing as mp
    import socket
    from conftest import golden_input, split_rows
    from repeatresolver_amd.window import slice_sections
    from test_window_sharding import _oracle_worker
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hip_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0] == got[1]
    rows = split_rows(golden_input("toy_b_b1000"))
    W = len(rows[0])
    secs = slice_sections(rows, [0, W // 4, W // 2, 3 * W // 4, W])
    for p, sec in enumerate(secs):
        exp, _ = _oracle_worker(sec, 300, 0, 2)
        assert got[0][p] == exp, p


def _nccl_single_rank_main(port, q):
    """The RCCL code path itself (`nccl` backend, CUDA tensors in the all-gather), as far as one GPU allows: a process
    group of ONE rank."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from conftest import golden_input, split_rows
    from repeatresolver_amd.sharding import _allgather_bytes, realign_sections
    from repeatresolver_amd.window import slice_sections
    rows = split_rows(golden_input("toy_b_b1000"))
    W = len(rows[0])
    secs = slice_sections(rows, [0, W // 2, W])
    out = realign_sections(secs, bandwidth=300, max_rounds=1)
    blobs = _allgather_bytes(b"xGMI" * 1000, torch.device("cuda", 0))          # the collective with device tensors
    q.put((out, blobs == [b"xGMI" * 1000], dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_path_runs_with_one_rank(oracle):
    """No multi-GPU node is available to these tests, and RCCL refuses two ranks on one device; what CAN run is the `nccl`
    branch of sharding.py under a one-rank RCCL process group: init, the all-gather of sizes and payloads on CUDA tensors,
    barrier, teardown.  (The N > 1 behaviour is covered with gloo: two ranks, one GPU.)"""
    import multiprocessing as mp
    import socket
    from conftest import golden_input, split_rows
    from repeatresolver_amd.window import slice_sections
    from test_window_sharding import _oracle_worker
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_single_rank_main, args=(port, q))
    p.start()
    out, gathered, backend = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and backend == "nccl" and gathered
    rows = split_rows(golden_input("toy_b_b1000"))
    W = len(rows[0])
    for sec, got in zip(slice_sections(rows, [0, W // 2, W]), out):
        exp, _ = _oracle_worker(sec, 300, 0, 1)
        assert got == exp


@pytest.mark.parametrize("split", ["sections", "rows"])
def test_bench_two_ranks_rehearsal(split):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank), rehearsed on one GPU
    with gloo: ONE MSA cut into its Window.py sections, sections dealt to the ranks, strong scaling, per-rank times
    (default) -- or, --split rows, the whole MSA on both ranks and every batch of the round split between them."""
    import subprocess
    import sys
    from conftest import ROOT
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))                                  # a free port: two suites may share a box
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "tree_medium", "--backend", "gloo", "--one-device", "--split", split]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2
    assert d["value"] > 0 and len(d["per_rank"]) == 2 and all(r["cells"] > 0 for r in d["per_rank"])
    assert ("sections" if split == "sections" else "replica") in d["config"]["workload"]
    if split == "rows":                                        # every replica committed every row of the two steps
        assert d["per_rank"][0]["cells"] == d["per_rank"][1]["cells"] == d["roofline"]["cells_reference"]
