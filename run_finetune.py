#!/usr/bin/env python
"""Fine-tuning CLI with the reference's flags (run_finetune.py:129-154):
--sweep | --domain_sweep D | --domain_name D --finetune_strategy {full_finetune,linear_probe} --pretrained_scheme {b1..s5} --seed N."""
import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from itertools import product
from pathlib import Path

DOMAINS = ["ENZYMES", "PTC_MR", "Cora_NC", "CiteSeer_NC", "Cora_LP", "CiteSeer_LP"]
STRATEGIES = ["full_finetune", "linear_probe"]
SCHEMES = ["b1", "b2", "b3", "b4", "s1", "s2", "s3", "s4", "s5"]
SEEDS = [42, 84, 126]


def run_one(job):
    domain, strategy, scheme, seed, extra = job
    env = dict(os.environ, PYTHONPATH=str(Path.cwd()))
    cmd = [sys.executable, "-m", "gnn_pretraining_amd.finetune.finetune", "--domain_name", domain, "--finetune_strategy", strategy,
           "--pretrained_scheme", scheme, "--seed", str(seed)] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    if r.returncode:
        print(f"x Failed: {domain} {strategy} {scheme} (seed={seed})")
        return False, job, f"Exit code {r.returncode}: {r.stderr}"
    print(f"ok Completed: {domain} {strategy} {scheme} (seed={seed}) {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ''}")
    return True, job, None


def main() -> None:
    ap = argparse.ArgumentParser(description="Run finetuning experiments")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--domain_sweep", type=str)
    ap.add_argument("--domain_name", type=str)
    ap.add_argument("--finetune_strategy", type=str, choices=STRATEGIES)
    ap.add_argument("--pretrained_scheme", type=str, choices=SCHEMES)
    ap.add_argument("--seed", type=int)
    args, extra = ap.parse_known_args()
    extra = [e for e in extra if e != "--"]
    if args.sweep or args.domain_sweep:
        doms = [args.domain_sweep] if args.domain_sweep else DOMAINS
        jobs = [(d, st, sc, s, extra) for d, st, sc, s in product(doms, STRATEGIES, SCHEMES, SEEDS)]
        with ThreadPoolExecutor(max_workers=1) as ex:
            results = list(ex.map(run_one, jobs))
        bad = [r for r in results if not r[0]]
        print(f"ok {len(results) - len(bad)}  failed {len(bad)}")
        for _, job, msg in bad:
            print(f"  - {job[:4]}: {msg}")
    elif args.domain_name:
        ok, _, msg = run_one((args.domain_name, args.finetune_strategy, args.pretrained_scheme, args.seed, extra))
        if not ok:
            print(f"Experiment failed: {msg}")
            sys.exit(1)
    else:
        print("Please specify --sweep, --domain_sweep or --domain_name/--finetune_strategy/--pretrained_scheme/--seed")
        sys.exit(1)


if __name__ == "__main__":
    main()
