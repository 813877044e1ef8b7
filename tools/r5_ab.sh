#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6a; mkdir -p $O
timeout 3000 python -m pytest tests/test_step_gpu.py tests/test_rccl_gpu.py tests/test_two_rank_gpu.py tests/test_bench_contract_gpu.py tests/test_abi.py -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; grep -E "tile plan|passed|failed|FAILED" $O/tests.log | cut -c1-260
