"""One step's kernel timeline per stream from a `rocprofv3 --kernel-trace --output-format csv` trace of bench.py:
python scripts/step_timeline.py <kernel_trace.csv> [step index from the end, default 20]
Steps are cut at the upload kernel (one per step).  Prints every kernel of that step with its stream / queue, start offset and duration (us)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "upload_kernel" in r[2]]
if len(marks) < back + 2:
    sys.exit("not enough steps in the trace")
# a step = from the adamw kernel of the previous step to the adamw kernel of this one
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
a, b = ad[-back - 1], ad[-back]
t0 = rows[a][1]
sel = rows[a + 1:b + 1]
streams = sorted({r[3] for r in sel})
print(f"step of {len(sel)} kernels, {(rows[b][1] - t0) / 1e3:.1f} us from the previous adamw's end to this adamw's end; streams {streams}")
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
busy = defaultdict(float)
for s, e, n, st, q in sel:
    busy[st] += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  s{st} q{q}  {short(n)}")
print("busy us per stream:", {k: round(v, 1) for k, v in busy.items()})
