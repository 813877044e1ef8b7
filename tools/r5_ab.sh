#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5v; mkdir -p $O
for v in A B A B; do
  if [ "$v" = A ]; then unset RSP_NO_DIRECT; else export RSP_NO_DIRECT=1; fi
  python3 tools/geom_bench.py small > $O/small_$v$RANDOM.txt 2>&1
done
unset RSP_NO_DIRECT
tail -n 13 $O/small_A*.txt | cut -c1-90; tail -n 13 $O/small_B*.txt | cut -c1-90
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_full_size_gpu.py tests/test_full_size_parity_gpu.py tests/test_teacher_forced_gpu.py tests/test_step_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
