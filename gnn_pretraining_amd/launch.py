"""Start the N ranks of a one-node data-parallel job as CHILD processes (one process per GPU).

`python bench.py --gpus N` calls this before anything in the parent has touched
the GPU: on this pool a process that has initialised HIP must never be replaced by another program, and
torch.cuda.device_count() -- the only device query made here -- does not initialise it.  The reference has no counterpart
(its only parallelism is job fan-out, run_pretrain.py:57-58); the rank environment is the one torch.distributed.run sets
(RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT), so a script works the same under either.
"""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
import time
from typing import Callable, List, Optional, Sequence, Tuple


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pick_backend(n: int, ndev: int) -> str:
    """RCCL ("nccl") needs a GPU per rank; with fewer the ranks share devices round-robin and exchange over gloo (a rehearsal
    of the N>1 code path, never a scaling measurement -- the result line says which backend ran)."""
    return "nccl" if ndev >= n else "gloo"


def launch_ranks(n: int, cmd: Sequence[str], ndev: Optional[int] = None, log: Callable[[str], None] = lambda m: None,
                 is_result: Callable[[str], bool] = lambda ln: ln.startswith("{"), timeout_s: Optional[float] = None,
                 extra_env: Optional[dict] = None) -> Tuple[int, Optional[str], List[int]]:
    """Run `cmd` as ranks 0..n-1.  Returns (exit code, rank 0's last result line or None, per-rank exit codes).
    Exit code is 0 only when every rank exited 0 AND rank 0 printed a result line.  When one rank dies the others are
    terminated (they would wait in a collective for ever)."""
    if ndev is None:
        import torch
        ndev = torch.cuda.device_count()
    port = free_port()
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("GMP_DIST_BACKEND", pick_backend(n, ndev))
    env.update(extra_env or {})
    log(f"launcher: {n} ranks on {ndev} visible GPU(s), backend {env['GMP_DIST_BACKEND']}, rendezvous 127.0.0.1:{port}")
    procs = [subprocess.Popen(list(cmd), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL) for r in range(n)]
    import threading
    chunks: List[bytes] = []
    reader = threading.Thread(target=lambda: chunks.extend(iter(lambda: procs[0].stdout.read(65536), b"")), daemon=True)
    reader.start()                                 # rank 0's stdout is drained as it comes (a full pipe would block the rank)
    t_end = None if timeout_s is None else time.time() + timeout_s
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or (t_end is not None and time.time() > t_end):
            failed = True
            for p in procs:                       # exact PIDs we started, nothing else
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10
            while any(p.poll() is None for p in procs) and time.time() < t_kill:
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out = b"".join(chunks).decode(errors="replace")
    line = None
    for ln in out.splitlines():
        if is_result(ln):
            line = ln
    rc = 0
    if failed or any(rcs):
        rc = next((c for c in rcs if c), 1)
    elif line is None:
        rc = 1
    if rc:
        log(f"launcher: rank exit codes {rcs}, result line {'missing' if line is None else 'present'}")
    return rc, line, rcs


DEFAULT_TIMEOUT_S = 1800.0


def run_and_relay(n: int, cmd: Sequence[str], log: Callable[[str], None], expect_key: str = "n_gpus", **kw) -> int:
    """launch_ranks + the check the bench contract needs: the relayed line must report n ranks under `expect_key`.
    A job in which EVERY rank hangs (say, in its first collective) exits no rank, so nothing above would ever react: the launch has a
    finite time limit (GMP_LAUNCH_TIMEOUT_S, default 30 minutes; `timeout_s=` overrides), after which the exact child PIDs are terminated
    and the call returns non-zero."""
    if kw.get("timeout_s") is None:
        kw["timeout_s"] = float(os.environ.get("GMP_LAUNCH_TIMEOUT_S", DEFAULT_TIMEOUT_S))
    rc, line, _ = launch_ranks(n, cmd, log=log, is_result=lambda ln: ln.startswith("{") and f'"{expect_key}"' in ln, **kw)
    if rc:
        return rc
    got = json.loads(line).get(expect_key)
    if got != n:
        log(f"launcher: result line reports {expect_key}={got} for {n} ranks")
        return 1
    print(line, flush=True)
    return 0


if __name__ == "__main__":          # python -m gnn_pretraining_amd.launch N script.py args...
    sys.exit(run_and_relay(int(sys.argv[1]), [sys.executable] + sys.argv[2:], log=lambda m: print(m, file=sys.stderr)))
