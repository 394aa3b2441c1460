#!/bin/bash
# 1-rank RCCL runs of bench.py with variations (tools only)
P=29620
for ARGS in "" "--planner-opt host_spin=0" "--collective torch" "--collective torch --planner-opt host_spin=0"; do
  P=$((P+1))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --steps 100 --warmup 5 --samples 2048 --no-cpu-baseline --no-ilt $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$ARGS |', round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_avg_ms'].items()})"
done
python bench.py --steps 100 --warmup 5 --samples 2048 --no-cpu-baseline --no-ilt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no group |', round(d['ms_per_step'],4))"
