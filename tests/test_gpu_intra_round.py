"""-m gpu: one realignment round of ONE MSA split over several ranks (include/pwr.h pwr_split_*, repeatresolver_amd/
intra_round.py; SURVEY 8e "within one MSA"): every rank is a replica of the whole state, fills and traces its share of
every speculative batch, the new placements are all-gathered, every rank commits all of them in the reference's order.
The MSA on every replica must be the one the reference reaches (the CPU oracle, pinned by tests/golden) -- bit for bit,
whatever the number of ranks.  This box has one GPU: two ranks share it under gloo (RCCL refuses two ranks on one device);
the `nccl` branch runs with a one-rank group."""
import os
import socket

import pytest

from conftest import golden_input, split_rows

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _workload(name):
    if name == "churn":                                    # low coverage, narrow band: columns open and empty by the dozen, commits go ahead
        from repeatresolver_amd import datagen as dg
        cfg = dg.SimConfig(kind="Tree", copies=6, coverage=12, difference=0.01, repeat_len=6000, flank=1500,
                           length_scale=0.03, min_aligned=60, seed=31)
        return [bytes(r) for r in dg.build_msa(dg.simulate(cfg))], 100, {"window": 16, "slack": 0}
    return split_rows(golden_input(name)), 1000, {"window": 4}


def _oracle_rounds(oracle, rows, bw, rounds):
    h = oracle.create(rows, bw)
    oracle.lib.pwo_trim(h)
    out = []
    for _ in range(rounds):
        oracle.lib.pwo_realign_round(h)
        out.append((oracle.lib.pwo_total_score(h), oracle.export(h), oracle.lib.pwo_cells(h)))
    oracle.lib.pwo_destroy(h)
    return out


@pytest.mark.parametrize("name", ["toy_b_b1000", "churn"])
def test_split_round_with_one_replica(name, oracle):
    """world = 1: the staged form of a batch (front half, export, import of nothing, commit) run by the per-batch loop."""
    from repeatresolver_amd.intra_round import SplitRound
    from repeatresolver_amd.realigner import PWReAligner
    rows, bw, opts = _workload(name)
    g = PWReAligner(rows, bandwidth=bw, **opts)
    g.trim_ends()
    sr = SplitRound(g, device=0)
    exp = _oracle_rounds(oracle, rows, bw, 2)
    for rnd in range(2):
        sr.realign_round()
        assert g.total_score() == exp[rnd][0]
        assert g.export_rows() == exp[rnd][1]
    assert g.stats()["cells_reference"] == exp[1][2]
    # ... and the context goes on as an ordinary one afterwards (every job its own again)
    g.realign_round()
    third = _oracle_rounds(oracle, rows, bw, 3)[2]
    assert g.export_rows() == third[1]
    assert sr.batches > 0
    g.close()


def _split_rank_main(rank, world, port, backend, name, rounds, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from repeatresolver_amd.intra_round import SplitRound
    from repeatresolver_amd.realigner import PWReAligner
    rows, bw, opts = _workload(name)
    g = PWReAligner(rows, bandwidth=bw, device=0, **opts)      # (every rank on device 0: this box has one GPU)
    g.trim_ends()
    sr = SplitRound(g, device=0)
    res = []
    for _ in range(rounds):
        sr.realign_round()
        res.append((g.total_score(), g.export_rows()))
    st = g.stats()
    q.put((rank, res, st, sr.batches, sr.bytes_gathered))
    dist.barrier()
    dist.destroy_process_group()
    g.close()


def _run_ranks(world, backend, name, rounds):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_rank_main, args=(r, world, port, backend, name, rounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=900)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("name", ["toy_b_b1000", "churn"])
def test_split_round_over_two_ranks(name, oracle):
    """Two replicas (gloo, one GPU): each fills and traces every other job of every batch; after every round both hold the
    reference's MSA.  "churn" regrows the column arrays in mid-slab (slack 0) and commits rows ahead of stale ones."""
    rows, bw, _ = _workload(name)
    rounds = 2
    exp = _oracle_rounds(oracle, rows, bw, rounds)
    got = _run_ranks(2, "gloo", name, rounds)
    for rank in (0, 1):
        res, st, batches, nbytes = got[rank]
        for rnd in range(rounds):
            assert res[rnd][0] == exp[rnd][0], (rank, rnd)
            assert res[rnd][1] == exp[rnd][1], (rank, rnd)
        assert st["cells_reference"] == exp[-1][2], rank          # every replica commits every row
        assert st["rows_committed"] == got[0][1]["rows_committed"]
        assert batches == got[0][2] and nbytes > 0
    # the fills were shared: each replica computed a part of the cells, together at least what the reference fills
    c0, c1 = got[0][1]["cells_computed"], got[1][1]["cells_computed"]
    assert c0 > 0 and c1 > 0 and c0 + c1 >= exp[-1][2]


def test_split_round_rccl_branch_with_one_rank(oracle):
    """The `nccl` branch (all_gather_into_tensor on device buffers, RCCL) under a one-rank group -- as far as one GPU goes."""
    rows, bw, _ = _workload("toy_b_b1000")
    exp = _oracle_rounds(oracle, rows, bw, 1)
    got = _run_ranks(1, "nccl", "toy_b_b1000", 1)
    res, st, batches, nbytes = got[0]
    assert res[0][0] == exp[0][0] and res[0][1] == exp[0][1]
    assert st["cells_reference"] == exp[0][2] and nbytes > 0


def _fullscale_rank_main(rank, world, port, n, q):
    import hashlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.intra_round import SplitRound
    from repeatresolver_amd.realigner import PWReAligner
    rows = [bytes(r) for r in dg.make_msa("tree_default")]

    def digest(g):
        h = hashlib.sha256()
        for r in g.export_rows():
            h.update(r)
        return h.hexdigest(), g.total_score(), g.dims()

    res = {}
    if rank == 0:                                              # the same rows on one context, no split
        g = PWReAligner(rows, bandwidth=1000, device=0, window=4)
        g.trim_ends()
        g.realign_rows(0, n)
        res["plain"] = digest(g)
        res["plain_cells"] = g.stats()["cells_reference"]
        g.close()
    g = PWReAligner(rows, bandwidth=1000, device=0, window=4)
    del rows
    g.trim_ends()
    SplitRound(g, device=0).realign_rows(0, n)
    res["split"] = digest(g)
    st = g.stats()
    res["split_cells"] = st["cells_reference"]
    res["computed"] = st["cells_computed"]
    res["stalls"] = st["stalls"]
    g.close()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_split_round_at_benchmark_scale():
    """The benchmark MSA (13 510 rows x 136 477 columns), the first rows of round 1: two replicas that split every batch
    end with the MSA one context reaches alone (which test_gpu_fullscale.py follows row for row with the oracle)."""
    import multiprocessing as mp
    n = 96
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fullscale_rank_main, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0]["split"] == got[1]["split"] == got[0]["plain"]
    assert got[0]["split_cells"] == got[1]["split_cells"] == got[0]["plain_cells"]
    assert got[0]["computed"] > 0 and got[1]["computed"] > 0
