#!/bin/bash
# On the GPU box: one bench workload under several (library, environment) variants, interleaved ROUNDS times so that
# drift shows.  Usage: tools/ab_env.sh [-r rounds] [-w "bench args"] name=lib[,ENV=val,...] ...
#   lib: "base" = librtmi.so, anything else = librtmi_<lib>.so (tools/ab_build.sh, or a copy kept by hand)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL="--no-extra --no-cpu-baseline --steps 5"
ROUNDS=2
while [ "${1:0:1}" = "-" ]; do
  case "$1" in
    -w) WL="$2"; shift 2;;
    -r) ROUNDS="$2"; shift 2;;
    *) echo "unknown option $1"; exit 2;;
  esac
done
for ROUND in $(seq 1 $ROUNDS); do
  for V in "$@"; do
    NAME=${V%%=*}; REST=${V#*=}
    IFS=, read -ra PARTS <<< "$REST"
    LIBN=${PARTS[0]}
    LIB=$ROOT/ray-tracing-cuda_amd/lib/librtmi_$LIBN.so
    [ "$LIBN" = base ] && LIB=$ROOT/ray-tracing-cuda_amd/lib/librtmi.so
    ENVS=("${PARTS[@]:1}")
    env RTMI_LIB_PATH=$LIB "${ENVS[@]}" python3 $ROOT/bench.py $WL 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-14s round $ROUND  kernel_ms %.2f  %.0f Mrays/s' % ('$NAME', d['config']['kernel_ms'], d['value']), flush=True)"
  done
done
