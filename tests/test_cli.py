"""CLI surface: flags and exit codes of the reference's drivers."""
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*args):
    return subprocess.run([sys.executable, *args], cwd=ROOT, capture_output=True, text=True)


def test_run_pretrain_requires_a_mode():
    r = run("run_pretrain.py")
    assert r.returncode == 1 and "Please specify" in r.stdout


def test_run_pretrain_reports_child_failure_with_exit_1():
    r = run("run_pretrain.py", "--exp_name", "nope", "--seed", "1")     # unknown scheme: the child raises KeyError
    assert r.returncode == 1
    assert "Experiment failed: Exit code" in r.stdout and "nope" in r.stdout


def test_run_finetune_flag_surface():
    r = run("run_finetune.py")
    assert r.returncode == 1
    r = run("run_finetune.py", "--domain_name", "Cora_NC", "--finetune_strategy", "bogus", "--pretrained_scheme", "b1", "--seed", "1")
    assert r.returncode == 2                                               # argparse choices, as in the reference
