#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5o; mkdir -p $O
S0=$SECONDS
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 > $O/line_1rank.json 2> $O/err_1rank.txt; echo "rc=$? wall=$((SECONDS-S0)) s"
python -c "
import json; d=json.loads(open('$O/line_1rank.json').read().strip().splitlines()[-1]); c=d['config']
print(round(d['value'],1), round(d['ms_per_step'],4), c['collective'], c['native_collective'], c['ranks_seen']['torch_world'], c['ranks_seen']['library_comm_world'], c['collective_timing'], c['supervisor']['attempts'][0]['ranks'], 'other' , {k:round(v['value'],1) for k,v in (d.get('other_configs') or {}).items()})
print({k:round(v['avg_ms'],4) for k,v in d['kernels_avg_ms'].items()})"
grep -v "amdgpu.ids" $O/err_1rank.txt | tail -5
