#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5b
timeout 1500 python -m pytest tests/test_rccl_gpu.py tests/test_graph_step_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "rccl or bucket or graph or policy or clip_gather or wgrad" > gpurun_out/r5b/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r5b/tests.log
tail -5 gpurun_out/r5b/tests.log
timeout 300 python tools/chain_gap_probe.py > gpurun_out/r5b/chain_gap.txt 2>&1; cat gpurun_out/r5b/chain_gap.txt
for arch in s3dg; do
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r5b/${arch}_plain.json 2> gpurun_out/r5b/${arch}_plain.err
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --force-dp > gpurun_out/r5b/${arch}_dp_seg.json 2> gpurun_out/r5b/${arch}_dp_seg.err
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --force-dp --graph on > gpurun_out/r5b/${arch}_dp_segon.json 2> gpurun_out/r5b/${arch}_dp_segon.err
done
timeout 600 python bench.py --arch c3d --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads > gpurun_out/r5b/c3d_plain.json 2> gpurun_out/r5b/c3d_plain.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5b/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        sm=d.get("steps_ms",{})
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("step_issue_mode"), "submit", sm.get("host_submit_p50"), "idle", sm.get("host_issue_idle_gpu_p50"), sm.get("segment_host_p50"), d.get("comm_ms"), d.get("hbm_kernels",{}).get("groups",{}).get("clip_gather"))
    except Exception as e:
        print(f, "ERR", e)
PY
