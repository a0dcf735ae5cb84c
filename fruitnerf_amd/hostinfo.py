"""Host-side facts a caller needs to size its thread pools (no HIP involved)."""
import os


def usable_cpus() -> int:
    """Host threads this process may actually keep busy: the hardware count capped by the container's CFS quota
    (cgroup v2 cpu.max / v1 cpu.cfs_quota_us).  On the MI355X boxes the quota is 16 CPUs of 256 hardware threads; a
    PyTorch / OpenMP pool sized by os.cpu_count() (128 threads) overruns it, and the kernel then freezes the WHOLE
    process for the rest of each 100 ms period — measured as 70-90 ms stalls of arbitrary host calls (a stream
    synchronize, a small D2H copy, plain Python) on every third 256^3 export pass, and as a 30x slower CPU baseline."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n
