"""Multi-GPU sharding of the PW_ReAligner path: independent MSA sections (the Window.py split) are
dealt round-robin to the ranks of one node, each rank realigns its sections on its own GPU with no
data-path communication, and the realigned sections are all-gathered at the end (RCCL over xGMI
when the process group is `nccl`, gloo in CPU tests).  One process per GPU."""
from __future__ import annotations

import torch
import torch.distributed as dist


def gpu_realign_section(rows, bandwidth, device, max_rounds=-1):
    """Default worker: the HIP path through the C ABI.  Returns (rows_out, score_lines)."""
    from .realigner import PWReAligner
    # (window 2: inside a Window.py section every row overlaps every other, a third speculative job per batch commits next to never and
    # takes SIMDs from the sections running beside this one: six sections on one GPU 425 against 464 ms per bench step)
    g = PWReAligner(rows, bandwidth=bandwidth, device=device, window=2)
    try:
        g.trim_ends()
        best = g.total_score()
        lines = [best]
        out = None
        rounds = 0
        while rounds < 10000 and (max_rounds < 0 or rounds < max_rounds):
            g.realign_round()
            rounds += 1
            tot = g.total_score()
            lines.append(tot)
            if tot < best:
                best = tot
                out = g.export_rows()
            else:
                break
        if out is None:                       # no improving round: the reference writes no file
            out = g.export_rows()
        return out, lines
    finally:
        g.close()


def _allgather_bytes(payload: bytes, device):
    """all_gather of one variable-length byte string per rank."""
    world = dist.get_world_size()
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    mx = max(int(s.item()) for s in sizes)
    buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=device)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    return [bytes(b[:int(s.item())].cpu().numpy().tobytes()) for b, s in zip(bufs, sizes)]


def realign_sections(sections, bandwidth=1000, max_rounds=-1, worker=None, device=None, concurrent=4):
    """sections: list (same on every rank) of sections, each a list of T rows.  Section p is realigned
    by rank p % world.  Returns the realigned sections in order, identical on every rank.

    One realignment keeps only a handful of CUs busy (it is a chain of dependent DP rows), so a rank
    runs up to `concurrent` of its sections side by side, each in its own context / HIP stream (the C
    calls release the GIL)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    use_cuda = dist.is_initialized() and dist.get_backend() == "nccl"
    if use_cuda and torch.cuda.device_count() < 1:
        raise RuntimeError("process group `nccl` (RCCL) but no HIP device is visible to this rank: one process per GPU")
    if device is None:
        device = torch.cuda.current_device() if (use_cuda or (not dist.is_initialized() and torch.cuda.is_available())) else 0
    worker = worker or gpu_realign_section
    mine = {}
    my_ids = list(range(rank, len(sections), world))
    if concurrent > 1 and len(my_ids) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(concurrent, len(my_ids))) as pool:
            futs = {p: pool.submit(worker, sections[p], bandwidth, device, max_rounds) for p in my_ids}
            for p, f in futs.items():
                mine[p] = f.result()[0]
    else:
        for p in my_ids:
            mine[p] = worker(sections[p], bandwidth, device, max_rounds)[0]
    if world == 1:
        return [mine[p] for p in range(len(sections))]
    # one message per rank: sections joined as  <p>\n<T>\n<row>\n...  records
    parts = []
    for p, rows in mine.items():
        parts.append(b"%d %d %d\n" % (p, len(rows), len(rows[0]) if rows else 0) + b"".join(rows))
    tdev = torch.device("cuda", device) if use_cuda else torch.device("cpu")
    blobs = _allgather_bytes(b"".join(parts), tdev)
    result = [None] * len(sections)
    for blob in blobs:
        i = 0
        while i < len(blob):
            nl = blob.index(b"\n", i)
            p, T, W = (int(v) for v in blob[i:nl].split())
            i = nl + 1
            result[p] = [blob[i + r * W:i + (r + 1) * W] for r in range(T)]
            i += T * W
    return result
