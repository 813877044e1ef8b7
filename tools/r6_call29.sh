#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6race3; mkdir -p $O
for cfg in "X=1" "RSP_NO_QOVERLAP=1"; do
  echo "== $cfg (60 steps, per-parameter checksums)" >> $O/race.txt
  env $cfg python3 tools/graph_vs_eager_fullsize.py resnet18 32 112 60 2>&1 | grep -v "amdgpu.ids\|graph mode" | tail -4 | cut -c1-420 >> $O/race.txt
done
echo "== 420 steps" >> $O/race.txt
python3 tools/graph_vs_eager_fullsize.py resnet18 32 112 420 2>&1 | grep -v "amdgpu.ids\|graph mode" | tail -4 | cut -c1-420 >> $O/race.txt
echo "== r2plus1d 60" >> $O/race.txt
python3 tools/graph_vs_eager_fullsize.py r2plus1d-vcop 32 112 60 2>&1 | grep -v "amdgpu.ids\|graph mode" | tail -4 | cut -c1-420 >> $O/race.txt
cat $O/race.txt
python3 -m pytest tests/test_kernels_gpu.py -q -x -k "dgrad or conv" 2>&1 | tail -3
