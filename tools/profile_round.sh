#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>      e.g. r1c
# Three separate rocprofv3 passes over the default bench (C3D 16x112x112, B=32): kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE counters each in their own run (MI355X_MICROARCH.md: one --pmc pass per counter group, never with sys traces).
# Raw outputs land in gpurun_out/; tools/summarize_profiles.py <round> <tag> turns them into profiles/<round>/.
R=$GRAFT_REPO_ROOT; TAG=$1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_fetch_$TAG $R/gpurun_out/pmc_write_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof_$TAG.json 2>/dev/null
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R; ls gpurun_out/prof_$TAG/*/ gpurun_out/pmc_fetch_$TAG/*/ gpurun_out/pmc_write_$TAG/*/ | head -20
# keep only the small files (the kernel trace of 7 steps is a few MB; counter CSVs can be large)
find gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG -name "*.csv" -size +30M -delete
tail -1 gpurun_out/bench_under_rocprof_$TAG.json
# MFMA utilisation: matrix-pipe busy cycles vs elapsed shader clocks, own pass (SQ + GRBM slots only)
cd /tmp
rm -rf $R/gpurun_out/pmc_mfma_$TAG
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R; find gpurun_out/pmc_mfma_$TAG -name "*.csv" -size +30M -delete; ls gpurun_out/pmc_mfma_$TAG/*/
