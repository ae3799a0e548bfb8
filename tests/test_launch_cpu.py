"""The rank launcher behind `python bench.py --gpus N` (gnn_pretraining_amd/launch.py), on CPU with gloo: N child
processes get the torch.distributed.run environment, rank 0's result line is relayed, a failing rank or a line that does
not report N ranks makes the launch fail."""
import json
import os
import sys

from gnn_pretraining_amd import launch

STUB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "rank_stub.py")


def run(mode, n=2, **kw):
    logs = []
    rc, line, rcs = launch.launch_ranks(n, [sys.executable, STUB, mode], ndev=0, log=logs.append,
                                        is_result=lambda ln: ln.startswith("{") and '"n_gpus"' in ln, timeout_s=120, **kw)
    return rc, line, rcs, logs


def test_two_ranks_rendezvous_and_rank0_line_is_relayed():
    rc, line, rcs, logs = run("ok")
    assert rc == 0 and rcs == [0, 0], logs
    out = json.loads(line)
    assert out["n_gpus"] == 2                    # the all-reduce saw both ranks
    assert out["backend"] == "gloo"              # no GPU per rank -> gloo rehearsal, and the line says so


def test_three_ranks():
    rc, line, rcs, _ = run("ok", n=3)
    assert rc == 0 and json.loads(line)["n_gpus"] == 3


def test_a_failing_rank_fails_the_launch_and_the_others_are_stopped():
    rc, line, rcs, logs = run("fail-rank1")
    assert rc != 0 and rcs[1] == 3
    assert any("exit codes" in m for m in logs)


def test_missing_result_line_is_a_failure():
    rc, line, _, _ = run("silent")
    assert rc == 1 and line is None


def test_a_line_that_reports_the_wrong_rank_count_is_refused(capsys):
    rc = launch.run_and_relay(2, [sys.executable, STUB, "lie"], log=lambda m: None, ndev=0, timeout_s=120)
    assert rc == 1
    assert capsys.readouterr().out == ""         # nothing relayed: never an n_gpus:1 line for a 2-rank launch


def test_a_job_in_which_every_rank_hangs_is_ended_by_the_default_time_limit(monkeypatch):
    """No rank exits, so only the time limit can end the launch: run_and_relay has a finite default (GMP_LAUNCH_TIMEOUT_S)."""
    import time
    assert launch.DEFAULT_TIMEOUT_S < float("inf")
    monkeypatch.setenv("GMP_LAUNCH_TIMEOUT_S", "3")
    t0 = time.time()
    rc = launch.run_and_relay(2, [sys.executable, STUB, "hang"], log=lambda m: None, ndev=0)
    assert rc != 0 and time.time() - t0 < 60


def test_backend_choice():
    assert launch.pick_backend(8, 8) == "nccl" and launch.pick_backend(2, 1) == "gloo" and launch.pick_backend(2, 0) == "gloo"
