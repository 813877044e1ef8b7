#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5l; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
