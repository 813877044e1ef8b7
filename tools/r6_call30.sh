#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6fin; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
soak() {  # tag arch steps extra env...
  local tag=$1 a=$2 n=$3 extra=$4; shift 4
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps $n --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$a $tag steps $n', d['value'], d['ms_per_step'], d.get('step_issue_mode'), 'final_loss', d['final_loss'])" >> $O/soak.txt
}
soak "eager" resnet18 400 "--graph off"
soak "lanes" resnet18 400 "--graph on"
soak "lanes dp" resnet18 400 "--force-dp"
soak "eager dp" resnet18 400 "--force-dp --graph off"
soak "eager" s3dg 300 "--graph off"
soak "lanes" s3dg 300 ""
soak "lanes dp" s3dg 300 "--force-dp"
soak "eager" r2plus1d-vcop 150 "--graph off"
soak "lanes" r2plus1d-vcop 150 "--graph on"
soak "eager" c3d 100 "--graph off"
soak "lanes" c3d 100 "--graph on"
cat $O/soak.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6fin/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac'], d['parity']['ok'], d['roofline']['traffic_stale'])
for k in ('resnet18','r2plus1d','s3dg'): print(k, d[f'{k}_clips_per_s'], d[f'{k}_whole_step_frac'], d[f'{k}_dominant_kernel_frac'])
print('dp', d['dp_path_at_one_rank']['clips_per_s'], {a:o['dp_path_at_one_rank']['clips_per_s'] for a,o in d['other_workloads'].items()})
PY
