#!/bin/bash
# Everything the round's committed profiles come from, in one GPU session:
#   bash profiles/collect_round.sh r02      (on the GPU box; writes gpurun_out/round_<tag>/ and gpurun_out/prof_*)
set -u
r=${1:-rNN}
out=gpurun_out/round_$r
mkdir -p $out
export TMPDIR=/tmp
# the driver's own command first (full line: cpu baseline, extras)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || echo "default bench failed"
bash profiles/collect.sh free_1000000_w256 > $out/collect_w256.log 2>&1
BENCH_ARGS="--walkers 512" PASSES="trace fetch write sq" bash profiles/collect.sh free_1000000_w512 > $out/collect_w512.log 2>&1
BENCH_ARGS="--variant zevol" PASSES="trace fetch write sq" bash profiles/collect.sh zevol_1000000_w256 > $out/collect_zevol.log 2>&1
BENCH_ARGS="--nsrc 100000" PASSES="trace fetch write" bash profiles/collect.sh free_100000_w256 > $out/collect_n1e5.log 2>&1
BENCH_ARGS="--no-cells" PASSES="trace fetch write sq" bash profiles/collect.sh free_1000000_w256_nocells > $out/collect_nocells.log 2>&1
# the same runs without the profiler (the numbers quoted next to the profiles)
b="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
python3 bench.py $b --walkers 512 > $out/bench_w512.json 2>/dev/null
python3 bench.py $b --walkers 1024 > $out/bench_w1024.json 2>/dev/null
python3 bench.py $b --nsrc 100000 > $out/bench_n1e5.json 2>/dev/null
python3 bench.py $b --nsrc 100000 --walkers 1024 > $out/bench_n1e5_w1024.json 2>/dev/null
python3 bench.py $b --nsrc 1000 --walkers 32 > $out/bench_n1e3.json 2>/dev/null
python3 bench.py $b --variant zevol > $out/bench_zevol.json 2>/dev/null
python3 bench.py $b --variant zevol --nsrc 800000 --walkers 512 > $out/bench_zevol_800000_w512.json 2>/dev/null   # BASELINE config 5's shape on one GPU
python3 bench.py $b --walkers 128 > $out/bench_w128.json 2>/dev/null     # config 4's share of one GPU: 1024 walkers over 8
python3 bench.py $b --walkers 64 > $out/bench_w64.json 2>/dev/null
python3 bench.py $b --variant zevol --no-cells > $out/bench_zevol_nocells.json 2>/dev/null
python3 bench.py $b --variant fixcomp > $out/bench_fixcomp.json 2>/dev/null
python3 bench.py $b --no-fuse > $out/bench_nofuse.json 2>/dev/null
python3 bench.py $b --no-cells > $out/bench_nocells.json 2>/dev/null
python3 bench.py $b --no-cells --no-tables > $out/bench_nocells_notables.json 2>/dev/null
python3 bench.py $b --profile-every 1 > $out/bench_events_every_launch.json 2>/dev/null
python3 bench.py $b --profile-level 0 > $out/bench_no_events.json 2>/dev/null
python3 tools/stamps.py > $out/stamps.txt 2>&1
python3 tools/time_parts.py --sets "default;skip_grid=1;cells=0;cells=0,skip_grid=1" > $out/time_parts.txt 2>&1
python3 tools/time_parts.py --rows 256 --sets "default;skip_grid=1" >> $out/time_parts.txt 2>&1
python3 tools/time_parts.py --variant zevol --sets "default;cells=0" >> $out/time_parts.txt 2>&1
python3 tools/call_period.py > $out/call_period.txt 2>&1
ls $out
