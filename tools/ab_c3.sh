#!/bin/bash
# On the GPU box: C3 medians of nine with each A/B library, alternated (A B ... A B ...) so that drift shows.
# Usage: tools/ab_c3.sh [-r rounds] <name> <name> ...   (names as given to tools/ab_build.sh; "base" = librtmi.so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=2
if [ "$1" = "-r" ]; then ROUNDS=$2; shift 2; fi
for ROUND in $(seq $ROUNDS); do
  for N in "$@"; do
    LIB=$ROOT/ray-tracing-cuda_amd/lib/librtmi_$N.so
    [ "$N" = base ] && LIB=$ROOT/ray-tracing-cuda_amd/lib/librtmi.so
    RTMI_LIB_PATH=$LIB python3 $ROOT/tools/gpu_c3_stats.py 9 || exit 1
  done
done
