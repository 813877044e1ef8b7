#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5z; mkdir -p $O
echo "== narrow off"; RSP_NARROW_MAX_TILES=0 timeout 900 python tools/grad_report.py 2>&1 | grep -E "^(c3d  |resnet18|s3dg|r2plus)" 
echo "== narrow off, multi split off"; RSP_NARROW_MAX_TILES=0 RSP_NO_MULTI_SPLIT=1 timeout 900 python tools/grad_report.py 2>&1 | grep -E "^(c3d  |resnet18|s3dg|r2plus)"
echo "== direct on"; RSP_DIRECT_MAX_TILES=448 timeout 900 python tools/grad_report.py 2>&1 | grep -E "^(c3d  |resnet18|s3dg|r2plus)"
