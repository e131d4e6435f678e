#!/bin/bash
# Profiles of one bench.py invocation, as committed under profiles/.  Run on the GPU box:
#   bash profiles/collect.sh <tag>        (writes gpurun_out/prof_<tag>/...)
# Pass 1: kernel trace + stats.  Further passes: PMC counters, each in its own run (--pmc with
# --kernel-trace only), FETCH_SIZE and WRITE_SIZE separately (TCC slots).
# PASSES="trace fetch write sq grbm lds" selects passes (default: all).
set -u
tag=${1:-run}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 20 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_ARGS:-}"
passes=${PASSES:-"trace fetch write sq grbm lds"}
for p in $passes; do
  case $p in
    trace) rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py $args > $out/trace.log 2>&1 || echo "trace failed"; continue;;
    fetch) c="FETCH_SIZE";;
    write) c="WRITE_SIZE";;
    sq) c="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY";;
    grbm) c="GRBM_GUI_ACTIVE";;
    lds) c="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA";;
  esac
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$n -o p -- python3 bench.py $args > $out/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
ls $out
