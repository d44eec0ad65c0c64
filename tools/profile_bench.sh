#!/bin/bash
# Collects the evidence bench.py's numbers are checked against (run on the GPU box):
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command  -> per-kernel durations
#   2. separate --pmc passes (SQ instruction mix, FETCH_SIZE, WRITE_SIZE) on a shorter run
# and writes CSV summaries + one JSON digest under gpurun_out/<tag>/.  Copy what should be
# judged into profiles/.  Usage: tools/profile_bench.sh <tag> [bench args for pass 1]
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
FULL="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
SHORT="python3 $ROOT/bench.py --spp 64 --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $FULL > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq -- $SHORT > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_sq2 -- $SHORT > $OUT/pmc_sq2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $SHORT > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $SHORT > $OUT/pmc_write.log 2>&1 || exit 1
python3 $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.json
cat $OUT/summary.json
