#!/bin/bash
# GPU box: whole-step sweep of environment settings:  bash tools/env_sweep.sh "arch1 arch2" "VAR=a VAR=b OTHER=c ..."   (each setting
# alone against the default, alternating with it; A=1,B=2 sets two variables together)
set -u
ARCHS=($1); SETS=($2)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"
run() { python3 bench.py --arch $1 --no-cpu-baseline --no-other-workloads --eager-steps 0 --steps 40 --warmup 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 ${2:-default}', d['value'], d['ms_per_step'], d['config']['step_issue'][:12])"; }
for a in "${ARCHS[@]}"; do
  run $a
  for s in "${SETS[@]}"; do
    env ${s//,/ } bash -c "$(declare -f run); run $a '$s'"
    run $a
  done
done
