#!/bin/bash
# Collects the rocprofv3 evidence for `python bench.py` on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats   -> per-kernel average durations
#   2. --pmc FETCH_SIZE         -> L2 fabric read requests   (own pass)
#   3. --pmc WRITE_SIZE         -> L2 fabric write requests  (own pass)
# Outputs land in gpurun_out/prof_$1/ ; summarise with profiles/summarise.py and copy the summary into profiles/$1/.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/bench_stats.json 2> $OUT/bench_stats.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras > $OUT/bench_write.json 2> $OUT/bench_write.err
ls $OUT/*
