"""Streams that really run beside each other.

The ROCm runtime multiplexes a process's HIP streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default; raising it
made the step 2.4x slower, profiles/README.md) and binds a stream to a queue at its first use; packets of one queue run in
order.  A kernel trace of the step showed the consequence: of the executor's four streams two shared the main stream's queue
(the link-prediction head ran behind the main stream's heads instead of beside them) while one queue carried nothing.  Which
queue a stream gets depends on everything the process created before, so this module does not guess: it takes streams from
torch's pool one by one and MEASURES each against the ones already chosen (libgnnmp gmp_streams_share_queue: a 400 us spin
kernel on one stream, an empty kernel on the other) until it holds `want` streams on queues of their own, distinct from the
main stream's.  The result is cached per device: every engine of a process uses the same streams."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Tuple

import torch

from . import _lib as L

_cache: Dict[Tuple[int, int], List["torch.cuda.Stream"]] = {}
last_report: Dict[str, object] = {}


def share_queue(a: int, b: int) -> bool:
    out = C.c_int(0)
    L.check(L.lib().gmp_streams_share_queue(a, b, C.byref(out)), "gmp_streams_share_queue")
    return bool(out.value)


def concurrent_streams(device, want: int = 3, max_candidates: int = 16) -> List["torch.cuda.Stream"]:
    """`want` streams for work beside torch's current stream on `device`, on hardware queues of their own where the runtime
    has that many (otherwise the remaining ones share the queue of an earlier chosen stream, never the main stream's when
    that can be avoided).  GMP_STREAM_CALIBRATION=0: the first `want` pool streams, unmeasured."""
    device = torch.device(device)
    main = torch.cuda.current_stream(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), main.cuda_stream)
    if key in _cache and len(_cache[key]) >= want:
        return _cache[key][:want]
    if os.environ.get("GMP_STREAM_CALIBRATION", "1") == "0":
        _cache[key] = [torch.cuda.Stream(device=device) for _ in range(want)]
        last_report.update(calibrated=False, own_queue=None, tried=want)
        return _cache[key]
    reps = [main.cuda_stream]                 # one representative per hardware queue seen so far
    chosen: List[torch.cuda.Stream] = []
    off_main: List[torch.cuda.Stream] = []   # streams that share a CHOSEN stream's queue (still beside main)
    on_main: List[torch.cuda.Stream] = []
    tried = 0
    while len(chosen) < want and tried < max_candidates:
        s = torch.cuda.Stream(device=device)
        tried += 1
        if share_queue(reps[0], s.cuda_stream):
            on_main.append(s)
        elif any(share_queue(r, s.cuda_stream) for r in reps[1:]):
            off_main.append(s)
        else:
            reps.append(s.cuda_stream)
            chosen.append(s)
    own = len(chosen)
    chosen += (off_main + on_main)[:want - len(chosen)]
    while len(chosen) < want:
        chosen.append(torch.cuda.Stream(device=device))
    _cache[key] = chosen
    last_report.update(calibrated=True, own_queue=own, tried=tried)
    return chosen[:want]
