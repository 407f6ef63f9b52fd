"""Multi-GPU plumbing: one process per GPU under ``torch.distributed``
(backend "nccl" = RCCL on ROCm; "gloo" on CPU-only hosts for tests).

Round 1 shards nothing on the data path (the serial sweep is a dependency
chain): ranks run independent replicas and only
synchronise around the timed region.  This module is the whole N>1 surface:
rendezvous, barrier, max-over-ranks of the elapsed time, and the replica
throughput aggregation used by bench.py."""

from __future__ import annotations

import os


class Comm:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = None
        if self.world > 1:
            import torch
            import torch.distributed as dist

            if torch.cuda.is_available():
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group("nccl", device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group("gloo")
            self.dist = dist

    def barrier(self):
        import torch

        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        import torch

        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        import torch

        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def replica_throughput(comm: Comm, units_this_rank: float, elapsed_this_rank: float) -> tuple[float, float]:
    """Whole-job throughput of independent replicas: (sum of units over ranks) /
    (max elapsed over ranks).  Returns (throughput, max_elapsed)."""
    tmax = comm.max_over_ranks(elapsed_this_rank)
    units = comm.sum_over_ranks(units_this_rank)
    return units / tmax, tmax
