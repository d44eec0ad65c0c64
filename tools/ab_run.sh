#!/bin/bash
# On the GPU box: C2 (5 steps) and one C4 shard with each A/B library, interleaved twice (A B A B) so that drift shows.
# Usage: tools/ab_run.sh [-w "bench args"] <name> <name> ...   (names as given to tools/ab_build.sh; "base" = librtmi.so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL="--no-extra --no-cpu-baseline --steps 5"
if [ "$1" = "-w" ]; then WL="$2"; shift 2; fi
mkdir -p $ROOT/gpurun_out/ab
for ROUND in 1 2; do
  for N in "$@"; do
    LIB=$ROOT/ray-tracing-cuda_amd/lib/librtmi_$N.so
    [ "$N" = base ] && LIB=$ROOT/ray-tracing-cuda_amd/lib/librtmi.so
    RTMI_LIB_PATH=$LIB python3 $ROOT/bench.py $WL 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-12s round $ROUND  kernel_ms %.2f  (min %.2f)  %.0f Mrays/s' % ('$N', d['config']['kernel_ms'], d['config']['kernel_ms'], d['value']))"
  done
done
