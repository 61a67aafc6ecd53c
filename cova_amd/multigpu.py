"""One-process-per-GPU plumbing for the filter stage.

The path shards by stream / GoP range exactly as the reference partitions a video
(gst-plugins/gst-gopsplit/gstgopsplit.cpp:556-603, one tracker per range,
cova-rs/gst-plugins/src/cova/tracker.rs:45): stream s runs on rank s mod world_size and no
data-path collective exists.  What the ranks of a job share is a rendezvous, barriers and the
MAX-over-ranks of the timed region -- all of it on CPU tensors over gloo: RCCL is never
initialised (round 5; rounds 1-4 opened an RCCL communicator for the barrier alone).
"""
from __future__ import annotations

import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def streams_of_rank(n_streams: int, rank: int, world: int) -> list[int]:
    """Stream ids owned by `rank` (round-robin: stream s -> rank s % world)."""
    return [s for s in range(n_streams) if s % world == rank]


def parse_cpulist(text: str) -> list[int]:
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11] (the sysfs cpulist format); [] for an empty / unreadable list."""
    out: list[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        try:
            if "-" in part:
                a, b = part.split("-", 1)
                out.extend(range(int(a), int(b) + 1))
            else:
                out.append(int(part))
        except ValueError:
            return []
    return out


def cpus_for_rank(allowed: list[int], node_cpus: list[int], sharers: int, share_index: int) -> list[int]:
    """The CPUs a rank pins itself to: the CPUs of its GPU's NUMA node that this process may use (`allowed`: its affinity mask,
    i.e. what the cgroup grants), divided evenly among the `sharers` ranks whose GPUs hang off that node (`share_index` = this
    rank's position among them).  With no usable node information (`node_cpus` empty, or disjoint from `allowed`) the allowed
    set itself is divided.  Never returns an empty set."""
    allowed = sorted(set(allowed))
    pool = sorted(set(node_cpus) & set(allowed)) or allowed
    sharers = max(1, sharers)
    share_index = min(max(0, share_index), sharers - 1)
    if len(pool) < sharers:
        return pool                                   # fewer cores than ranks: share them all
    lo = len(pool) * share_index // sharers
    hi = len(pool) * (share_index + 1) // sharers
    return pool[lo:hi]


def gpu_numa(device: int):
    """(numa_node, cpus of that node) of HIP device `device` from sysfs; (-1, []) when unknown.  Calls into libcovahip.so
    (hipDeviceGetPCIBusId): only from a process that is going to use the GPU anyway."""
    import ctypes as C
    from cova_amd import _lib as L
    buf = C.create_string_buffer(32)
    try:
        if L.lib().covahip_device_pci_bus_id(device, buf, len(buf)) != 0:
            return -1, []
        base = os.path.join("/sys/bus/pci/devices", buf.value.decode().lower())
        node = int(open(os.path.join(base, "numa_node")).read().strip())
        cpus = parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
        return node, cpus
    except (OSError, ValueError):
        return -1, []


def pin_to_gpu(device: int, sharers: int = 1, share_index: int = 0) -> dict:
    """Pins this process (and the threads it starts later) to the cores next to GPU `device` (cpus_for_rank).  Returns what it
    did: {"numa_node", "cpus", "pinned"}."""
    node, node_cpus = gpu_numa(device)
    try:
        allowed = sorted(os.sched_getaffinity(0))
        cpus = cpus_for_rank(allowed, node_cpus, sharers, share_index)
        os.sched_setaffinity(0, cpus)
        return {"numa_node": node, "cpus": cpus, "pinned": True}
    except (AttributeError, OSError):
        return {"numa_node": node, "cpus": [], "pinned": False}


class Group:
    """Rendezvous, barrier, MAX / SUM / gather over the ranks of a job -- gloo on CPU tensors, whatever the ranks compute on.
    No-op when world == 1 (torch is not even imported then)."""

    def __init__(self, backend: str | None = None):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.torch = None
        self.backend = "gloo"
        if backend not in (None, "gloo"):
            raise ValueError("the filter stage has no data-path collective: the job's control plane is gloo only")
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            dist.init_process_group("gloo")

    def barrier(self):
        """Host barrier.  The caller synchronises its own device work first (bench.py: ctx.sync())."""
        if self.dist is not None:
            self.dist.barrier()

    def _reduce(self, value: float, op) -> float:
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, value: float) -> float:
        return self._reduce(value, self.dist.ReduceOp.MAX if self.dist else None)

    def sum(self, value: float) -> float:
        return self._reduce(value, self.dist.ReduceOp.SUM if self.dist else None)

    def gather(self, obj) -> list:
        """Every rank's `obj` (picklable), in rank order, on every rank."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
