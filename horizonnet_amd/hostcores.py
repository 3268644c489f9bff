"""Host-core accounting for the host halves of the pipeline (Manhattan fits, label rasterisation, room rendering).

``os.cpu_count()`` reports the machine (256 logical CPUs on an MI355X box) while the container's cgroup may grant far
fewer (cpu.max = 16 there), and on a multi-GPU node every rank is its own process: each one sizing its pools for the whole
grant oversubscribes it `world` times over (8 ranks x 16 workers on 16 cores).  ``rank_cores()`` is what ONE rank
may use; ``pin_rank_affinity()`` additionally restricts the process to its own slice of the allowed CPUs so that the ranks'
host threads do not migrate onto each other."""
import math
import os


def usable_cores():
    """CPUs this process may really use: the affinity mask AND the cgroup quota."""
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


def local_world():
    """Ranks sharing this host (torch.distributed.run exports LOCAL_WORLD_SIZE; a bare WORLD_SIZE means one node)."""
    for key in ("LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        v = os.environ.get(key)
        if v and v.isdigit() and int(v) > 0:
            return int(v)
    return 1


def rank_cores(world=None):
    """Host cores ONE rank may use: usable_cores() // ranks on this host, at least 1."""
    return max(1, usable_cores() // max(1, local_world() if world is None else int(world)))


def pin_rank_affinity(local_rank=None, world=None):
    """Restrict this process to its slice of the allowed CPUs (slice size = allowed CPUs // ranks; the cgroup quota, if
    tighter, still applies on top).  Returns the CPU set, or None when nothing was changed (single rank, no affinity API,
    fewer CPUs than ranks)."""
    world = local_world() if world is None else int(world)
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1 or not hasattr(os, "sched_getaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    per = len(allowed) // world
    if per < 1:
        return None
    mine = set(allowed[local_rank * per:(local_rank + 1) * per])
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine
