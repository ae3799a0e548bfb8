"""Host-side thread policy.  Everything the host does per step is tiny index work (collating 32 graphs, drawing masks,
planning segments); torch's intra-op pool defaults to every core of the machine (128 on the GPU hosts, of which a
one-GPU job owns 16), and waking that pool for a 40,000-element op costs more than the op -- measured 34 ms/step with
the default against 4.2 ms/step with one thread (scripts/diag_train_loop.py)."""
from __future__ import annotations

import os

import torch


def host_cores(cap: int = 16) -> int:
    """Cores this process may actually use (affinity mask and cgroup-v2 CPU quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def limit_host_threads(n: int = 1) -> None:
    torch.set_num_threads(max(1, min(n, host_cores())))
    # (CPython's switch interval stays at its default: 500 / 100 / 30 us gave 1.61 / 1.72-1.88 / 1.86 ms per step against 1.59-1.67 with the
    # default in round 2 -- more hand-overs cost more than the waits they save -- and the GMP_SWITCH_INTERVAL_US switch was removed in round 3.)
