#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5f; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -8 $O/tests.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
head -c 1500 $O/bench_default.json
