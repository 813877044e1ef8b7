#!/bin/bash
# closing run: the GPU suite, smoke(), the default bench line, and the data-parallel path at one RCCL rank with / without DDP's buffer broadcast
cd "$(dirname "$0")/.."
O=gpurun_out/r6p; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
for a in s3dg resnet18; do
  for rep in 1 2; do
    python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 --force-dp 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$a dp', d['value'], d['ms_per_step'], d.get('step_issue_mode'), d.get('comm_ms'))" >> $O/dp.txt
  done
done
cat $O/dp.txt
