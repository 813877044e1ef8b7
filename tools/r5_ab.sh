#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5x; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
bash tools/ab_lib.sh tools/librspnet_hip_wg64.so s3dg resnet18 r2plus1d-vcop > $O/ab_wg64.txt 2>&1; cat $O/ab_wg64.txt | cut -c1-62
