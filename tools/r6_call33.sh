#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6end; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
/usr/bin/time -v python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; grep "Elapsed" $O/bench_default.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6end/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac'], d['parity']['ok'], d['roofline']['traffic_stale'])
for k in ('resnet18','r2plus1d','s3dg'): print(k, d[f'{k}_clips_per_s'], d[f'{k}_whole_step_frac'], d[f'{k}_dominant_kernel_frac'])
print('dp', d['dp_path_at_one_rank']['clips_per_s'], {a:o['dp_path_at_one_rank']['clips_per_s'] for a,o in d['other_workloads'].items()})
PY
