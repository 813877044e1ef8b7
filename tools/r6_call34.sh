#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6end; mkdir -p $O
S=$(date +%s)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench wall seconds: $(( $(date +%s) - S ))"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6end/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac'], d['parity']['ok'], d['roofline']['traffic_stale'])
for k in ('resnet18','r2plus1d','s3dg'): print(k, d[f'{k}_clips_per_s'], d[f'{k}_whole_step_frac'], d[f'{k}_dominant_kernel_frac'])
print('dp', d['dp_path_at_one_rank']['clips_per_s'], {a:o['dp_path_at_one_rank']['clips_per_s'] for a,o in d['other_workloads'].items()})
PY
