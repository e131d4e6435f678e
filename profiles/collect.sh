#!/bin/bash
# Profiles of one bench.py invocation, as committed under profiles/.  Run on the GPU box:
#   bash profiles/collect.sh <tag>        (writes gpurun_out/prof_<tag>/...)
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, each in its own run (--pmc with
# --kernel-trace only), FETCH_SIZE and WRITE_SIZE separately (TCC slots).
set -u
tag=${1:-run}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 20 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py $args > $out/trace.log 2>&1 || echo "trace failed"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$n -o p -- python3 bench.py $args > $out/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
ls $out
