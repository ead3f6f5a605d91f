"""One realignment round of ONE MSA split over the GPUs of a node (SURVEY 8e "within one MSA"; north star: "within a
round non-overlapping sequence realignments are partitioned across the GPUs with RCCL ... all-gather").

One process per GPU, each holding a replica of the whole MSA state (built from the same text with the same options).
The k loop of PW_ReAligner.c:1695 advances in speculative batches as on one GPU: the next `window` rows are
gathered from the last committed state; rank r FILLS and TRACES the jobs j of the batch with j % world == r; the new
placements -- one fixed-size record per job: sizeof(JobMeta) + 4 * Lmax + 8 * (Lmax / 32 + 1) bytes, 150 KB at benchmark
scale, of which only the row's own L entries are meaningful -- are all-gathered (RCCL over xGMI when the group is `nccl`); then every rank commits ALL jobs in row order with the same validation, so the replicas stay identical and no
tally ever has to be sent: the "broadcast of the updated column tallies" is each replica applying the same delta.
The result is bit-identical to PWReAligner.realign_rows on one GPU (and to the reference).

EXPERIMENTAL and unmeasured on more than one GPU: every batch is serialised on the host; sections
(sharding.py) are the scaling path.  All device work is libpwr.so (include/pwr.h, pwr_split_*); this module only owns the collective.
A failure on one rank (a HIP error, an allocation that fails in a regrow) is carried to every rank with the next collective:
the ranks all-reduce a status word per batch and raise together instead of leaving the others blocked in the all-gather."""
from __future__ import annotations

import torch
import torch.distributed as dist


class SplitRound:
    """Buffers and the per-batch loop for one PWReAligner replica of a process group."""

    def __init__(self, realigner, group=None, device=None):
        self.g = realigner
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if device is None:
            device = torch.cuda.current_device()
        # ("cpu": host buffers -- only for tests of the loop and of the records' layout with a stand-in for the context)
        self.dev = torch.device("cpu") if device == "cpu" else torch.device("cuda", device)
        self.nccl = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.send = self.recv = None
        self.batches = 0
        self.bytes_gathered = 0

    def _buffers(self):
        slot, per_rank = self.g.split_slot_bytes()
        n = slot * per_rank
        if self.send is None or self.send.numel() != n:
            self.send = torch.zeros(n, dtype=torch.uint8, device=self.dev)
            self.recv = torch.zeros(n * self.world, dtype=torch.uint8, device=self.dev)
            if not self.nccl and self.world > 1:
                pin = self.dev.type == "cuda"
                self.h_send = torch.zeros(n, dtype=torch.uint8, pin_memory=pin)
                self.h_recv = torch.zeros(n * self.world, dtype=torch.uint8, pin_memory=pin)

    def _all_gather(self):
        if self.nccl:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)      # RCCL (over xGMI between the GPUs of a node)
        elif self.world == 1:
            self.recv.copy_(self.send)
        else:
            # rehearsal backends (gloo): through pinned host memory
            self.h_send.copy_(self.send)
            parts = list(self.h_recv.chunk(self.world))
            dist.all_gather(parts, self.h_send, group=self.group)
            self.recv.copy_(self.h_recv)
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)       # the commit runs on the context's own stream
        self.bytes_gathered += self.recv.numel()

    def _raise_together(self, err, why):
        """one status word per rank, max-reduced: a rank that failed does not leave the others waiting in the next collective"""
        if self.world > 1:
            t = torch.tensor([err], dtype=torch.int32, device=self.dev if self.nccl else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            if int(t.item()) and not err:
                raise RuntimeError("split round: another rank of the group failed in this batch")
        if err:
            raise why

    def realign_rows(self, k0, n):
        """Rows k0 .. k0+n-1 of the round, in the reference's order; collective: every rank of the group calls it."""
        self.g.split_begin(k0, n, self.rank, self.world)
        if n == 0:
            return
        self._buffers()
        left = n
        budget = 4 * n + 256                       # every batch commits at least its first row, bar the rare repeat (stall, failed segment check)
        err, why = 0, None
        while True:
            if budget == 0:
                raise RuntimeError("split round: no progress")
            budget -= 1
            if not err:
                try:
                    self.g.split_stage(self.send.data_ptr())
                except Exception as e:             # (raised below, on every rank)
                    err, why = 1, e
            self._raise_together(err, why)         # (also carries a failure of the commit before)
            self._all_gather()
            try:
                left = self.g.split_commit(self.recv.data_ptr())
            except Exception as e:
                err, why = 1, e
            self.batches += 1
            if not err and left <= 0:
                break
        self._raise_together(0, None)              # pairs with the status exchange of a rank whose last commit failed

    def realign_round(self):
        self.realign_rows(0, self.g.T)
