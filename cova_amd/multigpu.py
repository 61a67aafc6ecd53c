"""One-process-per-GPU plumbing for the filter stage.

The path shards by stream / GoP range exactly as the reference partitions a video
(gst-plugins/gst-gopsplit/gstgopsplit.cpp:556-603, one tracker per range,
cova-rs/gst-plugins/src/cova/tracker.rs:45): stream s runs on rank s mod world_size and no
data-path collective exists.  torch.distributed is used for rendezvous, barriers and the
MAX-over-ranks of the timed region only (backend "nccl" = RCCL on GPUs, "gloo" in the CPU
tests).
"""
from __future__ import annotations

import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def streams_of_rank(n_streams: int, rank: int, world: int) -> list[int]:
    """Stream ids owned by `rank` (round-robin: stream s -> rank s % world)."""
    return [s for s in range(n_streams) if s % world == rank]


class Group:
    """Thin wrapper: no-op when world == 1 (torch is not even imported then)."""

    def __init__(self, backend: str | None = None):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.torch = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            self.backend = backend
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend)

    def _dev(self):
        return "cuda" if self.dist is not None and self.backend == "nccl" else "cpu"

    def barrier(self):
        if self.dist is not None:
            if self.backend == "nccl":
                self.torch.cuda.synchronize()
            self.dist.barrier()

    def max(self, value: float) -> float:
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value: float) -> float:
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
