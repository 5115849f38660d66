#!/bin/bash
# bench.py's multi-rank path on the one-GPU box: 2 and 4 ranks share GPU 0, collectives over gloo on host copies
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rehearse
for N in 2 4; do
  for X in slabs gather; do
    timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 2 --warmup 1 --grid 128 --rehearse-on-one-gpu --exchange $X > gpurun_out/rehearse/n${N}_$X.json 2> gpurun_out/rehearse/n${N}_$X.err || { tail -20 gpurun_out/rehearse/n${N}_$X.err; exit 1; }
    python3 -c "
import json; r=json.loads(open('gpurun_out/rehearse/n${N}_$X.json').read().strip().splitlines()[-1]); print('N=$N $X:', r['n_gpus'], r['scaling'], r['config']['parallelism'], 'nnu_this_rank', r['config']['nnu_this_rank'], 'ndir_this_rank', r['config']['ndir_this_rank'], 'value %.3e' % r['value'])"
  done
done
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 3 --steps 2 --warmup 1 --grid 128 --rehearse-on-one-gpu > gpurun_out/rehearse/n3.json 2> gpurun_out/rehearse/n3.err || { tail -20 gpurun_out/rehearse/n3.err; exit 1; }
python3 -c "
import json; r=json.loads(open('gpurun_out/rehearse/n3.json').read().strip().splitlines()[-1]); print('N=3:', r['config']['parallelism'], 'nnu_this_rank', r['config']['nnu_this_rank'], 'ndir_this_rank', r['config']['ndir_this_rank'])"
