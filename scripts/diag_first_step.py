"""First step of many freshly built engines (same seed, same inputs) in one process: per-task gradients must be bit-identical from
engine to engine.  The test suite builds dozens of engines per process; a difference here names a first-use / many-streams race.
python scripts/diag_first_step.py [scheme] [engines]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_engine import build

scheme = sys.argv[1] if len(sys.argv) > 1 else "s3"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ref, bad, keep = None, 0, []
for k in range(count):
    om, hm, eng, host, inp, gen, tasks, domains = build(scheme, 147)
    art = eng.draw(inp, gen)
    eng.temperature, eng.grl_lambda = 0.41, 0.006
    eng.step(inp, gen, art=art, order=[t for t in tasks if t != "domain_adv"], apply_update=False)
    torch.cuda.synchronize()
    cur = eng.task_grads.clone()
    flags = eng.sync_flags.cpu().tolist() if eng.use_gates else None
    if ref is None:
        ref = cur
    elif not torch.equal(cur, ref):
        bad += 1
        diff = cur != ref
        names = [f"{task}:{n}" for t, task in enumerate(eng.tasks) for n in eng.names if diff[t, eng.off[n]:eng.off[n] + eng.numel[n]].any()]
        print(f"engine {k}: differs in {names[:10]}; gates {'on' if flags else 'off'} err flag {flags[63] if flags else None}", flush=True)
    keep.append(eng)                      # engines stay alive, as pytest's fixtures and tracebacks keep them
    print(f"engine {k}: gates {eng.use_gates}, streams {[int(s.cuda_stream) if s is not None else 0 for s in eng.task_streams]}", flush=True)
print(f"{scheme}: {bad} of {count - 1} engines differ from the first")
