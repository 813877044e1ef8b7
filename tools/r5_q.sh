#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5g; mkdir -p $O
timeout 1500 python -m pytest tests/test_rccl_gpu.py tests/test_graph_step_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -5 $O/tests.log
timeout 900 python tools/grad_census.py --batch 4 --hw 32 --queue 64 > $O/census_small.json 2> $O/census_small.err; echo "census small rc $?"; tail -3 $O/census_small.err
timeout 1500 python tools/grad_census.py > $O/grad_census_c3d.json 2> $O/census.err; echo "census rc $?"; tail -3 $O/census.err
for arch in s3dg resnet18; do
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --no-other-workloads --graph on > $O/${arch}_lanes.json 2> $O/${arch}_lanes.err
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --no-other-workloads --force-dp --graph on > $O/${arch}_dp_lanes.json 2> $O/${arch}_dp_lanes.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5g/*lanes.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        sm=d.get("steps_ms",{})
        sh=sm.get("segment_host_p50") or {}
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("step_issue_mode"), "idle", sm.get("host_issue_idle_gpu_p50"), "ngraphs", sum(1 for k in sh if k.startswith("graph")), d.get("comm_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
