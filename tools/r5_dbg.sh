#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5w; mkdir -p $O
python3 tools/grad_dump.py /tmp/g_on.pt c3d 4 32 64 2>&1 | tail -1
RSP_NARROW_MAX_TILES=0 python3 tools/grad_dump.py /tmp/g_off.pt c3d 4 32 64 2>&1 | tail -1
python3 tools/grad_dump.py --compare /tmp/g_on.pt /tmp/g_off.pt > $O/cmp.txt 2>&1; cat $O/cmp.txt
