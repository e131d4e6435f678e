#!/bin/bash
# Everything the round's committed profiles come from, in one GPU session:
#   bash profiles/collect_round.sh r03      (on the GPU box; writes gpurun_out/round_<tag>/ and gpurun_out/prof_*)
#   BENCH_ONLY=1 bash profiles/collect_round.sh r03      (the bench lines and tools only, no rocprofv3 passes)
set -u
r=${1:-rNN}
out=gpurun_out/round_$r
mkdir -p $out
export TMPDIR=/tmp
# the driver's own command first (full line: cpu baseline, extras)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || echo "default bench failed"
if [ -z "${BENCH_ONLY:-}" ]; then
# rocprofv3 passes (kernel trace + stats; FETCH_SIZE, WRITE_SIZE, SQ counters each in a run of its own)
PASSES="trace fetch write sq" bash profiles/collect.sh free_1000000_w256 > $out/collect_w256.log 2>&1
BENCH_ARGS="--walkers 512" PASSES="trace fetch write" bash profiles/collect.sh free_1000000_w512 > $out/collect_w512.log 2>&1
BENCH_ARGS="--variant zevol" PASSES="trace fetch write sq" bash profiles/collect.sh zevol_1000000_w256 > $out/collect_zevol.log 2>&1
BENCH_ARGS="--variant fixcomp" PASSES="trace" bash profiles/collect.sh fixcomp_1000000_w256 > $out/collect_fixcomp.log 2>&1
BENCH_ARGS="--no-cells" PASSES="trace fetch write sq" bash profiles/collect.sh free_1000000_w256_nocells > $out/collect_nocells.log 2>&1
BENCH_ARGS="--no-grid-shortcut" PASSES="trace" bash profiles/collect.sh free_1000000_w256_nogridshortcut > $out/collect_nogq.log 2>&1
fi
# the same runs without the profiler (the numbers quoted next to the profiles)
b="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for c in 1 2 3 5; do python3 bench.py $b --config $c > $out/bench_config$c.json 2>/dev/null; done
python3 bench.py $b --walkers 1024 > $out/bench_w1024.json 2>/dev/null
python3 bench.py $b --walkers 2048 > $out/bench_w2048.json 2>/dev/null
python3 bench.py $b --walkers 128 > $out/bench_w128.json 2>/dev/null     # config 4's share of one GPU: 1024 walkers over 8
python3 bench.py $b --variant zevol > $out/bench_zevol.json 2>/dev/null
python3 bench.py $b --variant fixcomp > $out/bench_fixcomp.json 2>/dev/null
python3 bench.py $b --no-fuse > $out/bench_nofuse.json 2>/dev/null
python3 bench.py $b --no-cells > $out/bench_nocells.json 2>/dev/null
python3 bench.py $b --no-grid-shortcut > $out/bench_nogridshortcut.json 2>/dev/null
python3 bench.py $b --profile-level 0 > $out/bench_no_events.json 2>/dev/null
python3 bench.py $b --force-collective > $out/bench_force_collective.json 2>/dev/null
for v in free zevol fixcomp; do python3 tools/stamps_fused.py --variant $v > $out/stamps_fused_$v.txt 2>&1; done
python3 tools/host_burst.py > $out/host_burst.txt 2>&1
for v in free zevol fixcomp; do python3 tools/sampler_ab.py $v >> $out/sampler_ab.txt 2>&1; done
for v in free zevol fixcomp; do python3 tools/call_period.py --variant $v --levels 0,0 >> $out/call_period.txt 2>&1; done
python3 tools/call_period.py --levels 0 --rows 256 >> $out/call_period.txt 2>&1
python3 tools/call_period.py --levels 0 --rows 64 >> $out/call_period.txt 2>&1
python3 tools/call_period.py --levels 0 --nsrc 1000 --rows 16 >> $out/call_period.txt 2>&1
python3 tools/call_period.py --levels 0 --variant zevol --nsrc 800000 --rows 256 >> $out/call_period.txt 2>&1
ls $out
