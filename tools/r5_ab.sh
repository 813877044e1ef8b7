#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5o; mkdir -p $O
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_full_size_gpu.py tests/test_full_size_parity_gpu.py tests/test_teacher_forced_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
bash tools/ab_lib.sh tools/librspnet_hip_nolong.so r2plus1d-vcop s3dg resnet18 > $O/ab_nolong.txt 2>&1; cat $O/ab_nolong.txt | cut -c1-60
