#!/bin/bash
# sweep the tail K-split factor per C3D layer (ablation build); prints fwd ms per S
cd $GRAFT_REPO_ROOT
for S in 1 2 3 4 5 6 7 8 9 10 12 14 16; do
  echo -n "S=$S  "; RSP_SPLIT=$S python tools/conv_bench.py --what fwd --tune 0 2>&1 | grep -E "conv3a|conv3b|conv4a|conv4b|conv5a" | awk '{print $1, $6}' | tr '\n' ' '; echo
done
echo "== whole split (full=0)"
for S in 2 3 4 5 6 8; do
  echo -n "S=$S  "; RSP_FULL=0 RSP_SPLIT=$S python tools/conv_bench.py --what fwd --tune 0 2>&1 | grep -E "conv4a|conv4b|conv5a" | awk '{print $1, $6}' | tr '\n' ' '; echo
done
