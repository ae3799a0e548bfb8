"""A stand-in for bench.py's rank body used by tests/test_launch_cpu.py: joins the job the launcher started (gloo), makes one
all-reduce and lets rank 0 print a result line.  argv[1]: "ok" | "fail-rank1" | "lie" (reports n_gpus 1) | "silent" | "hang" (every rank sleeps: nobody exits)."""
import json
import os
import sys

import torch
import torch.distributed as dist

mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if mode == "fail-rank1" and rank == 1:
    sys.exit(3)                       # dies before the rendezvous: the others must not wait for ever
if mode == "hang":
    import time
    time.sleep(600)
dist.init_process_group("gloo")
t = torch.ones(1)
dist.all_reduce(t)
if rank == 0 and mode != "silent":
    print("noise before the line")
    print(json.dumps({"metric": "stub", "n_gpus": 1 if mode == "lie" else int(t.item()), "backend": os.environ.get("GMP_DIST_BACKEND")}), flush=True)
dist.destroy_process_group()
