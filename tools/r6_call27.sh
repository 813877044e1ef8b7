#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6race; mkdir -p $O
for cfg in "RSP_BWD_PIECE=0" "RSP_BWD_TAIL_CUT_GFLOP=0" "RSP_NO_QOVERLAP=1" "RSP_GRAPH_MODE=segments"; do
  echo "== $cfg" >> $O/race.txt
  env $cfg python3 tools/graph_vs_eager_fullsize.py resnet18 32 112 420 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/race.txt
done
cat $O/race.txt
