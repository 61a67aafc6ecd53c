"""Host-side helpers of bench.py that decide how the CPU baseline is sized (no GPU, no oracle)."""
import bench


def test_cpu_quota_reads_cgroup_lines():
    assert bench.cpu_quota("1600000 100000\n") == 16          # a GPU box's share: 16 of its 256 hardware threads
    assert bench.cpu_quota("max 100000") is None              # unlimited
    assert bench.cpu_quota("50000 100000") == 1               # half a CPU still runs one thread
    assert bench.cpu_quota("-1 100000") is None               # cgroup v1 spelling of "no quota"
    assert bench.cpu_quota("") is None and bench.cpu_quota("garbage") is None


def test_effective_cores_is_positive_and_not_above_the_hardware():
    import os
    n = bench.effective_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
