#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5n; mkdir -p $O
S0=$SECONDS; timeout -k 10 400 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? wall=$((SECONDS-S0)) s"; python -c "
import json; d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])
print(round(d['value'],1), round(d['ms_per_step'],4), d['roofline']['traffic'] is not None)
for k,v in d['other_configs'].items(): print(k, v['name'], round(v['value'],1), round(v['ms_per_step'],3), v['rollout_body'], round(v['roofline_step_frac'],3))"
