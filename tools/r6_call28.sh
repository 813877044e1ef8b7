#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6race2; mkdir -p $O
for cfg in "X=1" "RSP_TASK_RUN_AHEAD=0" "RSP_NO_KOVERLAP=1" "RSP_NO_POOL_FUSION=1" "RSP_NO_MULTI_SPLIT=1" "RSP_WGRAD_ASIDE_GFLOP=0 RSP_WGRAD_MID_GFLOP=0" "RSP_BWD_PIECE=0" "RSP_NO_QOVERLAP=1"; do
  echo "== $cfg" >> $O/race.txt
  env $cfg python3 tools/graph_vs_eager_fullsize.py resnet18 32 112 60 2>&1 | grep -v "amdgpu.ids\|graph mode" | tail -5 | cut -c1-420 >> $O/race.txt
done
echo "== s3dg" >> $O/race.txt
python3 tools/graph_vs_eager_fullsize.py s3dg 16 224 60 2>&1 | grep -v "amdgpu.ids\|graph mode" | tail -5 | cut -c1-420 >> $O/race.txt
echo "== c3d" >> $O/race.txt
python3 tools/graph_vs_eager_fullsize.py c3d 32 112 40 2>&1 | grep -v "amdgpu.ids\|graph mode" | tail -5 | cut -c1-420 >> $O/race.txt
cat $O/race.txt
