#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5k; mkdir -p $O
timeout 2400 python -m pytest tests/test_two_rank_gpu.py tests/test_step_gpu.py tests/test_teacher_forced_gpu.py tests/test_rccl_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "two_rank or step or teacher or rccl or clip_gather" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -4 $O/tests.log
bash tools/profile_round.sh r5p
