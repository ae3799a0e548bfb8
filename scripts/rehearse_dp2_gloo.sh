set -e
export GMP_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 30 --warmup 5 --no-roofline 2>&1 | tail -2
