#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5i; mkdir -p $O
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -6 $O/tests.log
for arch in resnet18 s3dg r2plus1d-vcop; do
  timeout 600 python tools/layer_table.py --arch $arch --top 80 > $O/layers_${arch}_new.txt 2>&1
  RSP_NO_DIRECT=1 RSP_NO_MULTI_SPLIT=1 timeout 600 python tools/layer_table.py --arch $arch --top 80 > $O/layers_${arch}_old.txt 2>&1
  head -2 $O/layers_${arch}_new.txt | tail -1; head -2 $O/layers_${arch}_old.txt | tail -1
done
for arch in resnet18 s3dg r2plus1d-vcop; do
  timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --no-other-workloads > $O/${arch}_new.json 2> $O/${arch}_new.err
  RSP_NO_DIRECT=1 RSP_NO_MULTI_SPLIT=1 timeout 600 python bench.py --arch $arch --steps 30 --warmup 10 --no-cpu-baseline --no-other-workloads > $O/${arch}_old.json 2> $O/${arch}_old.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5i/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("step_issue_mode"))
    except Exception as e:
        print(f, "ERR", e)
PY
