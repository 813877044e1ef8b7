#!/bin/bash
# round 5 closing run: the whole GPU suite, soaks in the new issue modes, training sanity
cd "$(dirname "$0")/.."
O=gpurun_out/r5j; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -6 $O/tests.log
timeout 900 python bench.py --arch s3dg --steps 800 --warmup 10 --no-cpu-baseline --no-other-workloads > $O/soak_s3dg_lanes.json 2> $O/soak_s3dg_lanes.err
timeout 900 python bench.py --arch s3dg --steps 800 --warmup 10 --no-cpu-baseline --no-other-workloads --force-dp > $O/soak_s3dg_dp_lanes.json 2> $O/soak_s3dg_dp_lanes.err
timeout 900 python bench.py --arch resnet18 --steps 800 --warmup 10 --no-cpu-baseline --no-other-workloads --force-dp > $O/soak_resnet18_dp_lanes.json 2> $O/soak_resnet18_dp_lanes.err
timeout 900 python bench.py --arch c3d --steps 300 --warmup 10 --no-cpu-baseline --no-other-workloads --force-dp > $O/soak_c3d_dp.json 2> $O/soak_c3d_dp.err
timeout 900 python tools/train_sanity.py > $O/train_sanity.txt 2>&1; tail -7 $O/train_sanity.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5j/soak*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        sm=d.get("steps_ms",{})
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["steps"], d.get("step_issue_mode"), "loss", d["final_loss"], "p50", sm.get("p50"), "max", sm.get("max"), "idle", sm.get("host_issue_idle_gpu_p50"), d.get("comm_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
