#!/bin/bash
# The costs behind lf_free's deal of bins and cell chunks (lfmcmc.hip: ensure_deal), swept on ONE box in ONE call:
#   LF_DEAL_H = how far behind the younger half of the virtual workgroups is counted, LF_DEAL_B = cost of a flux bin
#   (a cell chunk costs 3).  Prints the period of back-to-back evaluations at 128, 256 and 64 rows per setting.
#   gpurun -- 'bash tools/deal_sweep.sh'
for hb in "8 8" "0 8" "4 8" "10 8" "12 8" "16 8" "8 6" "12 10"; do set -- $hb; echo "H=$1 B=$2"
  for rows in 128 256 64; do LF_DEAL_H=$1 LF_DEAL_B=$2 python3 tools/call_period.py --levels 0 --rows $rows 2>/dev/null | grep profiling; done
done
echo "arithmetic deal (LF_NO_DEAL=1)"; for rows in 128 256 64; do LF_NO_DEAL=1 python3 tools/call_period.py --levels 0 --rows $rows 2>/dev/null | grep profiling; done
