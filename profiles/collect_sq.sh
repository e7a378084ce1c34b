#!/bin/bash
# Two SQ counter passes (8 SQ counters each, one rocprofv3 run per pass) over one step of bench.py.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq_a -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras > $OUT/bench_sq_a.json 2> $OUT/bench_sq_a.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_LDS_ATOMIC --output-format csv -d $OUT/sq_b -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras > $OUT/bench_sq_b.json 2> $OUT/bench_sq_b.err
ls $OUT
