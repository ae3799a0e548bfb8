"""Does work parked on OTHER hardware queues slow the dispatch of a dependent kernel chain on this one?  main runs a chain of
200 x 10 us spin kernels (single wave each: pure launch-to-launch cadence); K other streams (each on a hardware queue of its own,
streams.concurrent_streams) hold work that waits for the end of the chain, parked either by hipStreamWaitEvent (a barrier packet
in the queue) or by a device-side gate kernel (one sleeping wave).  Everything is enqueued behind a 30 ms spin, so the host is
never the limiter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_pretraining_amd import _lib as L, streams as ST

dev = torch.device("cuda:0")
lib = L.lib()
main = torch.cuda.current_stream(dev)
others = ST.concurrent_streams(dev, 3)
print("streams:", ST.last_report)
flag = torch.zeros(64, dtype=torch.int32, device=dev)
N, US = 200, 10


def run(K, how):
    flag.zero_()
    torch.cuda.synchronize()
    t0, t1, E = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
    L.check(lib.gmp_spin_us(30000, main.cuda_stream), "spin")
    if how == "gate":                       # the gates must be enqueued (and may start) before the chain: they only sleep
        for s in others[:K]:
            L.check(lib.gmp_gate_wait(flag.data_ptr(), 1, 1, None, s.cuda_stream), "gate")
            L.check(lib.gmp_spin_us(1, s.cuda_stream), "spin")
    t0.record(main)
    for _ in range(N):
        L.check(lib.gmp_spin_us(US, main.cuda_stream), "spin")
    t1.record(main)
    if how == "gate":
        L.check(lib.gmp_gate_open(flag.data_ptr(), 1, main.cuda_stream), "open")
    else:
        E.record(main)
        for s in others[:K]:
            s.wait_event(E)
            L.check(lib.gmp_spin_us(1, s.cuda_stream), "spin")
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / N * 1e3


for how in ("event", "gate"):
    for K in (0, 1, 2, 3):
        r = [run(K, how) for _ in range(3)]
        print(f"{how:6s} {K} parked streams: {min(r):6.2f} us per 10-us kernel of the chain (3 runs: {[round(x, 2) for x in r]})")


# What does a wait on an event that has LONG completed cost the waiting stream?  The same chain with one / three such waits in
# front of every kernel (events recorded on another stream and completed before the chain starts).
def run_waits(nwait, late=False):
    torch.cuda.synchronize()
    done = []
    if late:      # the events complete AFTER the host has enqueued the waits (20 ms spin) but before the chain runs (30 ms spin)
        L.check(lib.gmp_spin_us(20000, others[0].cuda_stream), "spin")
    for _ in range(3):
        e = torch.cuda.Event()
        L.check(lib.gmp_spin_us(1, others[0].cuda_stream), "spin")
        e.record(others[0])
        done.append(e)
    if not late:
        torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.check(lib.gmp_spin_us(30000, main.cuda_stream), "spin")
    t0.record(main)
    for _ in range(N):
        for e in done[:nwait]:
            main.wait_event(e)
        L.check(lib.gmp_spin_us(US, main.cuda_stream), "spin")
    t1.record(main)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / N * 1e3


for late in (False, True):
    for nwait in (0, 1, 3):
        r = [run_waits(nwait, late) for _ in range(3)]
        print(f"{nwait} waits before every kernel on events complete {'when the chain RUNS (not yet when enqueued)' if late else 'when ENQUEUED'}: "
              f"{min(r):6.2f} us per 10-us kernel (3 runs: {[round(x, 2) for x in r]})")


# ... and what does an event RECORD between two kernels cost the recording stream?
def run_records(nrec):
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.check(lib.gmp_spin_us(30000, main.cuda_stream), "spin")
    t0.record(main)
    evs = []
    for _ in range(N):
        for _ in range(nrec):
            e = torch.cuda.Event(); e.record(main); evs.append(e)
        L.check(lib.gmp_spin_us(US, main.cuda_stream), "spin")
    t1.record(main)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / N * 1e3


for nrec in (0, 1, 2):
    r = [run_records(nrec) for _ in range(3)]
    print(f"{nrec} event records before every kernel: {min(r):6.2f} us per 10-us kernel (3 runs: {[round(x, 2) for x in r]})")
