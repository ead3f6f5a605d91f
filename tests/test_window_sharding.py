"""Section split (Window.py restatement) and the N>1 sharding path on gloo, world_size 2, CPU.
The worker injected here is the CPU oracle (test infrastructure); on GPUs the default worker is the
HIP path, exercised by tests/test_gpu_parity.py::test_sections_on_gpu."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from conftest import golden_input, split_rows


def _py2_window(rows, coverage=0.90, parts=6):
    """Literal transcription of the arithmetic of Window.py:41-60 with explicit floor division
    (what Python 2 '/' does on ints) -- the check for window_boundaries."""
    MA = [r.decode("latin1") for r in rows]
    Coverages = [sum([1 for z in range(len(MA)) if MA[z][c] != ' ']) for c in range(0, len(MA[0]), 100)]
    average = sum(Coverages) // len(Coverages)
    start = 0
    while Coverages[start] < coverage * average:
        start += 1
    start *= 100
    ende = len(Coverages) - 1
    while Coverages[ende] < coverage * average:
        ende -= 1
    ende *= 100
    return [start] + [start + (p + 1) * (ende - start) // parts for p in range(parts)]


def test_window_boundaries_match_window_py():
    from repeatresolver_amd.window import merge_sections, slice_sections, window_boundaries
    from conftest import golden_output
    for name in ("toy_a_b1000", "ia_toy_b1000", "toy_b_b1000"):
        rows = split_rows(golden_output(name))          # a realigned MSA (has blank margins)
        for parts in (2, 6):
            b = window_boundaries(rows, 0.90, parts)
            assert b == _py2_window(rows, 0.90, parts)
            assert len(b) == parts + 1 and all(b[i] <= b[i + 1] for i in range(parts))
        secs = slice_sections(rows, b)
        merged = merge_sections(secs)
        assert merged == [r[b[0]:b[-1]] for r in rows]
    raw = split_rows(golden_input("toy_a_b1000"))      # un-realigned: no blanks -> equal-width cuts
    b = window_boundaries(raw, 0.90, 6)
    assert b[0] == 0 and b[-1] == 100 * ((len(raw[0]) - 1) // 100)


def _oracle_worker(rows, bandwidth, device, max_rounds):
    from conftest import Oracle
    o = Oracle()
    h = o.create(rows, bandwidth)
    o.lib.pwo_trim(h)
    best = o.lib.pwo_total_score(h)
    lines = [best]
    out = None
    rounds = 0
    while max_rounds < 0 or rounds < max_rounds:
        o.lib.pwo_realign_round(h)
        rounds += 1
        tot = o.lib.pwo_total_score(h)
        lines.append(tot)
        if tot < best:
            best = tot
            out = o.export(h)
        else:
            break
    if out is None:
        out = o.export(h)
    o.lib.pwo_destroy(h)
    return out, lines


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from repeatresolver_amd.sharding import realign_sections
    from repeatresolver_amd.window import slice_sections
    rows = split_rows(golden_input("toy_a_b1000"))
    W = len(rows[0])
    bounds = [0, W // 3, 2 * W // 3, W]
    secs = slice_sections(rows, bounds)
    out = realign_sections(secs, bandwidth=200, max_rounds=1, worker=_oracle_worker)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sections_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank holds all sections, identical, and equal to realigning each section on its own
    assert got[0] == got[1]
    from repeatresolver_amd.window import slice_sections
    rows = split_rows(golden_input("toy_a_b1000"))
    W = len(rows[0])
    secs = slice_sections(rows, [0, W // 3, 2 * W // 3, W])
    for p, sec in enumerate(secs):
        exp, _ = _oracle_worker(sec, 200, 0, 1)
        assert got[0][p] == exp


class _FakeReplica:
    """Stand-in for a PWReAligner in the per-batch loop of intra_round.SplitRound (no GPU here): records what the loop
    hands to pwr_split_stage / pwr_split_commit and checks the layout the import kernel relies on -- after the all-gather,
    rank r's slots_per_rank records lie at offset r * slots_per_rank * slot_bytes."""
    SLOT, PER_RANK = 48, 2

    def __init__(self, rank, world, T):
        self.rank, self.world, self.T = rank, world, T
        self.batch = 0
        self.left = 0
        self.errors = []

    def split_begin(self, k0, n, rank, world):
        assert (rank, world) == (self.rank, self.world)
        self.left = n

    def split_slot_bytes(self):
        return self.SLOT, self.PER_RANK

    def _record(self, rank, slot):
        return bytes(((rank * 31 + slot * 7 + self.batch * 3 + i) & 0xff) for i in range(self.SLOT))

    def split_stage(self, ptr):
        import ctypes
        data = b"".join(self._record(self.rank, s) for s in range(self.PER_RANK))
        ctypes.memmove(ptr, data, len(data))

    def split_commit(self, ptr):
        import ctypes
        n = self.SLOT * self.PER_RANK
        got = ctypes.string_at(ptr, n * self.world)
        for r in range(self.world):
            exp = b"".join(self._record(r, s) for s in range(self.PER_RANK))
            if got[r * n:(r + 1) * n] != exp:
                self.errors.append((self.batch, r))
        self.batch += 1
        self.left = max(0, self.left - 3)                 # (a batch commits a few rows)
        return self.left


def _split_loop_rank_main(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from repeatresolver_amd.intra_round import SplitRound
    g = _FakeReplica(rank, world, 40)
    sr = SplitRound(g, device="cpu")
    sr.realign_round()
    q.put((rank, g.batch, g.errors, sr.batches, sr.bytes_gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_split_round_loop_and_record_layout_gloo_world2():
    """The host side of a round split over ranks (intra_round.py) with two ranks under gloo: stage / all-gather / commit
    until no row is left, every rank's records where pwr_split_commit expects them."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_loop_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, batches, errors, sr_batches, nbytes in got:
        assert errors == [] and batches == sr_batches == 14        # ceil(40 / 3)
        assert nbytes == 14 * 2 * _FakeReplica.SLOT * _FakeReplica.PER_RANK
