"""Host-core accounting for the CPU legs (TEST INFRASTRUCTURE ONLY).

``os.cpu_count()`` reports the machine (256 logical CPUs on the MI355X box) while the container's
cgroup may grant far fewer (cpu.max = 16 there); running torch with 256 threads on a 16-CPU quota is
~8x slower than with 16.  ``usable_cores()`` is the number the CPU baseline is allowed to use and reports.
"""
import math
import os


def usable_cores():
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(math.floor(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n
