#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6r; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -6 $O/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
