#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5e; mkdir -p $O
timeout 1500 python -m pytest tests/test_rccl_gpu.py tests/test_graph_step_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -5 $O/tests.log
for arch in s3dg resnet18 r2plus1d-vcop c3d; do
  for piece in 4 8 16; do
    RSP_BWD_PIECE=$piece timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --no-other-workloads --graph on > $O/${arch}_lanes_p$piece.json 2> $O/${arch}_lanes_p$piece.err
  done
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --no-other-workloads --graph off > $O/${arch}_eager.json 2> $O/${arch}_eager.err
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --no-other-workloads --force-dp --graph on > $O/${arch}_dp_lanes.json 2> $O/${arch}_dp_lanes.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5e/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        sm=d.get("steps_ms",{})
        sh=sm.get("segment_host_p50") or {}
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("step_issue_mode"), "submit", sm.get("host_submit_p50"), "idle", sm.get("host_issue_idle_gpu_p50"), "ngraphs", sum(1 for k in sh if k.startswith("graph")), d.get("comm_ms"), d.get("hbm_kernels",{}).get("groups",{}).get("clip_gather",{}).get("tb_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
