"""Run-to-run reproducibility of one stacked step: the same inputs and artefacts, dropout off, no update -- every per-task gradient
and loss must come out bit-identical on every repetition (all reductions are ordered).  A difference names a race between streams.
python scripts/diag_step_repeat.py [scheme] [reps] [burst]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_engine import build

scheme = sys.argv[1] if len(sys.argv) > 1 else "s3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
burst = int(sys.argv[3]) if len(sys.argv) > 3 else 1           # steps enqueued back to back before each comparison (the host runs ahead, as in training)
om, hm, eng, host, inp, gen, tasks, domains = build(scheme, 147)
art = eng.draw(inp, gen)
eng.temperature, eng.grl_lambda = 0.41, 0.006
order = [t for t in tasks if t != "domain_adv"]
ref = None
bad = 0
for k in range(reps):
    for _ in range(burst):
        eng.step(inp, gen, art=art, order=order, apply_update=False)
    torch.cuda.synchronize()
    cur = (eng.task_grads.clone(), eng.loss_sums.clone())
    if ref is None:
        ref = cur
        continue
    if not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])):
        bad += 1
        diff = (cur[0] != ref[0])
        names = []
        for t, task in enumerate(eng.tasks):
            for n in eng.names:
                o = eng.off[n]
                if diff[t, o:o + eng.numel[n]].any():
                    names.append(f"{task}:{n}")
        print(f"rep {k}: differs in {names[:12]}{' ...' if len(names) > 12 else ''}; losses equal {torch.equal(cur[1], ref[1])}", flush=True)
print(f"{scheme}: {bad} of {reps - 1} repetitions differ from the first")
