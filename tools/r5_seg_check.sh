#!/bin/bash
# round 5: segmented replay of the data-parallel step — tests, then same-box A/B against the one-graph N=1 step and the eager DP step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_rccl_gpu.py tests/test_graph_step_gpu.py -x -q -m gpu > gpurun_out/r5a/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r5a/tests.log
tail -5 gpurun_out/r5a/tests.log
for arch in s3dg resnet18; do
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r5a/${arch}_plain.json 2> gpurun_out/r5a/${arch}_plain.err
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --force-dp > gpurun_out/r5a/${arch}_dp_seg.json 2> gpurun_out/r5a/${arch}_dp_seg.err
  RSP_NO_SEGMENTS=1 timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --force-dp > gpurun_out/r5a/${arch}_dp_eager.json 2> gpurun_out/r5a/${arch}_dp_eager.err
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --force-dp --graph on > gpurun_out/r5a/${arch}_dp_segon.json 2> gpurun_out/r5a/${arch}_dp_segon.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5a/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("step_issue_mode"), d.get("steps_ms",{}).get("host_submit_p50"), d.get("comm_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
