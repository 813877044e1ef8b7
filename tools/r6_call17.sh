#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6q; mkdir -p $O
python3 -m pytest tests/test_rccl_gpu.py tests/test_two_rank_gpu.py tests/test_graph_step_gpu.py tests/test_pretrain_gpu.py -x -q > $O/gpu_dp_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_dp_tests.log
tail -8 $O/gpu_dp_tests.log
