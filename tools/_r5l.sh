#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_full.log; tail -3 $O/pytest_full.log
timeout -k 10 300 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value'],1), round(d['ms_per_step'],4), 'gru frac', round(r['frac'],3), 'traffic', r['traffic'], '| rollout', round(r['also']['frac'],3), r['also']['traffic'], '| step', round(r['step']['frac'],3), '| ilt', round(d['roofline_ilt']['frac'],3), d['roofline_ilt']['traffic'], '| cpu', round(d['cpu_baseline']['value'],3), d['cpu_baseline']['cores'], '| commit', d['config']['commit'])
print(str(r['traffic_source'])[:200])"
python -c "import __graft_entry__ as g; g.smoke()"
