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


def cpus_for_rank(allowed: list[int], node_cpus: list[int], sharers: int, share_index: int,
                  world: int = 0, rank: int = 0) -> list[int]:
    """The CPUs a rank pins itself to: the CPUs of its GPU's NUMA node that this process may use (`allowed`: its affinity mask,
    i.e. what the cgroup grants), divided evenly among the `sharers` ranks whose GPUs hang off that node (`share_index` = this
    rank's position among them).  With no usable node information (`node_cpus` empty, or disjoint from `allowed`) the allowed
    set itself is divided -- among ALL `world` ranks when the caller says how many there are (ranks on different nodes would
    otherwise take identical slices of it), else among the sharers.  Never returns an empty set."""
    allowed = sorted(set(allowed))
    pool = sorted(set(node_cpus) & set(allowed))
    if not pool:
        pool = allowed
        if world > 0:
            sharers, share_index = world, rank
    sharers = max(1, sharers)
    share_index = min(max(0, share_index), sharers - 1)
    if len(pool) < sharers:
        return pool                                   # fewer cores than ranks: share them all
    lo = len(pool) * share_index // sharers
    hi = len(pool) * (share_index + 1) // sharers
    return pool[lo:hi]


def _numa_of_bus_id(bus_id: str):
    try:
        base = os.path.join("/sys/bus/pci/devices", bus_id.lower())
        node = int(open(os.path.join(base, "numa_node")).read().strip())
        cpus = parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
        return node, cpus
    except (OSError, ValueError):
        return -1, []


def gpu_numa(device: int):
    """(numa_node, cpus of that node) of HIP device `device` from sysfs; (-1, []) when unknown.  Calls into libcovahip.so
    (hipDeviceGetPCIBusId) IN THIS PROCESS: only from a process that has initialised the GPU anyway.  A process that wants to pin
    itself first uses gpu_numa_in_child."""
    import ctypes as C
    from cova_amd import _lib as L
    buf = C.create_string_buffer(32)
    try:
        if L.lib().covahip_device_pci_bus_id(device, buf, len(buf)) != 0:
            return -1, []
        return _numa_of_bus_id(buf.value.decode())
    except OSError:
        return -1, []


def kfd_gpu_bus_ids(root: str = "/sys/class/kfd/kfd/topology/nodes", dev_root: str = "/dev/dri") -> list[str]:
    """PCI addresses ("dddd:bb:dd.f") of the GPUs this process can open, in HIP device order, from the KFD topology in sysfs -- no
    HIP call, no GPU context, no child process: topology nodes with SIMDs are GPUs; the runtime enumerates the ones whose render
    node (/dev/dri/renderD<drm_render_minor>) this process may open, in node order; ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES
    lists of plain indices select from that order as the runtime does.  [] when the topology is unreadable or a list holds
    something else than indices (UUIDs): the caller then falls back (gpu_numa_in_child)."""
    try:
        nodes = sorted((int(n) for n in os.listdir(root) if n.isdigit()))
    except OSError:
        return []
    gpus = []
    for n in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(root, str(n), "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) == 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
            if minor >= 0 and not os.access(os.path.join(dev_root, f"renderD{minor}"), os.R_OK | os.W_OK):
                continue
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            gpus.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}")
        except (OSError, ValueError, KeyError):
            return []
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        text = os.environ.get(var)
        if text is None:
            continue
        try:
            idx = [int(x) for x in text.split(",") if x.strip() != ""]
        except ValueError:
            return []
        gpus = [gpus[i] for i in idx if 0 <= i < len(gpus)]
    return gpus


def gpu_numa_sysfs(device: int, modulo_present: bool = False):
    """(numa_node, cpus of that node, device actually meant) of HIP device `device` from sysfs alone (kfd_gpu_bus_ids); None when
    the topology does not say.  What a rank asks BEFORE it touches the GPU: nothing here starts a runtime thread or opens the
    device, so several ranks can ask at once without adding processes to the card."""
    ids = kfd_gpu_bus_ids()
    if not ids:
        return None
    d = device % len(ids) if modulo_present else device
    if not 0 <= d < len(ids):
        return None
    node, cpus = _numa_of_bus_id(ids[d])
    return node, cpus, d


def gpu_numa_in_child(device: int, modulo_present: bool = False):
    """gpu_numa, but the HIP call runs in a short-lived child process: the caller has not touched the GPU afterwards, so it can
    still pin itself BEFORE the HIP runtime, torch and gloo start their threads (sched_setaffinity binds the calling thread and
    the threads created later, not the ones that exist).  `modulo_present`: device % (GPUs present), the rehearsal's mapping.
    Returns (numa_node, cpus, device actually asked)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, ctypes as C\n"
        f"sys.path.insert(0, {root!r})\n"
        "from cova_amd import _lib as L\n"
        "n = C.c_int(0); L.lib().covahip_device_count(C.byref(n))\n"
        f"d = {int(device)} % max(1, n.value) if {bool(modulo_present)} else {int(device)}\n"
        "b = C.create_string_buffer(32)\n"
        "rc = L.lib().covahip_device_pci_bus_id(d, b, len(b))\n"
        "print('BUS', d, b.value.decode() if rc == 0 else '-')\n")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        for line in r.stdout.splitlines():
            if line.startswith("BUS "):
                _, d, bus = line.split()
                node, cpus = _numa_of_bus_id(bus) if bus != "-" else (-1, [])
                return node, cpus, int(d)
    except (OSError, subprocess.SubprocessError, ValueError):
        pass
    return -1, [], device


def pin_threads(cpus: list[int]) -> int:
    """Binds EVERY thread this process has right now (and, through the calling thread, the ones it starts later) to `cpus`:
    sched_setaffinity(0) alone leaves the threads that already exist -- the HIP / HSA runtime's, gloo's, OpenMP's -- where they
    were.  Returns the number of threads bound."""
    n = 0
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = []
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
            n += 1
        except OSError:
            pass                                      # a thread that has just exited
    os.sched_setaffinity(0, cpus)
    return max(n, 1)


def pin_to_gpu(device: int, sharers: int = 1, share_index: int = 0, node_info=None, world: int = 0, rank: int = 0) -> dict:
    """Pins this process -- all of its current threads and the ones it starts later -- to the cores next to GPU `device`
    (cpus_for_rank).  `node_info` = (numa_node, node_cpus) when the caller already has it (gpu_numa_in_child: no HIP call in
    this process before the pin).  Returns what it did: {"numa_node", "cpus", "pinned", "threads_bound"}."""
    node, node_cpus = node_info if node_info is not None else gpu_numa(device)
    try:
        allowed = sorted(os.sched_getaffinity(0))
        cpus = cpus_for_rank(allowed, node_cpus, sharers, share_index, world, rank)
        n = pin_threads(cpus)
        return {"numa_node": node, "cpus": cpus, "pinned": True, "threads_bound": n}
    except (AttributeError, OSError):
        return {"numa_node": node, "cpus": [], "pinned": False, "threads_bound": 0}


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
