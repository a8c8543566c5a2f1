for cfg in "2 4" "3 6"; do set -- $cfg
SCALCE_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 6 --warmup 2 --reads 20000000 --cpu-sample 0 --group $1 --inflight $2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('world2', sys.argv[1:], 'hbm', d['config']['hbm_used_gb'], 'ms', d['ms_per_step'])" $cfg
done
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --reads 20000000 --cpu-sample 0 --group 3 --inflight 6 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('world1 3 6 hbm', d['config']['hbm_used_gb'])"
