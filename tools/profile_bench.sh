#!/bin/bash
# Collects the evidence bench.py's numbers are checked against (run on the GPU box):
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command -> per-kernel durations
#      (C2 headline + the C3 line of config.extra);
#   2. separate --pmc passes (SQ instruction mix three times -- the third counts the binary64 instructions --, FETCH_SIZE,
#      WRITE_SIZE) on ONE launch of
#      each workload at its full size (c2: cornell 1024^2 x1024 spp d50, c3: bunny 1024^2 x512 spp),
#      so instruction counts and HBM bytes are per launch of exactly what bench.py times;
# and writes CSVs + <tag>_summary.json + roofline_inputs.json under gpurun_out/<tag>/.  Copy what
# should be judged into profiles/ (tools/summarize_profile.py prints the cp commands).
# Workload tags: c2 c3 c1 c1big (whole frames) and c4s0of8 c5s0of8 (shard 0 of 8 of c4 / c5 at full spp: what
# one GPU of eight renders) -- every line of bench.py's config.extra has its own digest.
# Usage: tools/profile_bench.sh <tag> [workloads, default "c2 c3 c1 c1big c4s0of8 c5s0of8"]
set -u
TAG=${1:-prof}; shift || true
WLS=${*:-c2 c3 c1 c1big c4s0of8 c5s0of8}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32"
SQ3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"
# (a gpurun call is at most 20 minutes: PROF_SKIP_TRACE=1 / PROF_NO_SUMMARY=1 split the collection over two calls whose
# outputs merge under gpurun_out/<tag>/; tools/summarize_profile.py then runs on the merged directory, anywhere)
if [ -z "${PROF_SKIP_TRACE:-}" ]; then
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
  echo "trace done"
fi
for WL in $WLS; do
  case $WL in
    c4s*|c5s*) SH=${WL#c?s}; ARGS="--workload ${WL%%s*} --shard ${SH%of*}/${SH#*of}";;
    *) ARGS="--workload $WL";;
  esac
  ONE="python3 $ROOT/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-extra"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/${WL}_pmc_sq -- $ONE > $OUT/${WL}_pmc_sq.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $OUT/${WL}_pmc_sq2 -- $ONE > $OUT/${WL}_pmc_sq2.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ3 --output-format csv -d $OUT/${WL}_pmc_sq3 -- $ONE > $OUT/${WL}_pmc_sq3.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${WL}_pmc_fetch -- $ONE > $OUT/${WL}_pmc_fetch.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${WL}_pmc_write -- $ONE > $OUT/${WL}_pmc_write.log 2>&1 || exit 1
  echo "$WL pmc done"
done
[ -n "${PROF_NO_SUMMARY:-}" ] && exit 0
python3 $ROOT/tools/summarize_profile.py $OUT $TAG $WLS > $OUT/${TAG}_summary.json
cat $OUT/${TAG}_summary.json
