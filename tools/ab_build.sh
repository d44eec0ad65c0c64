#!/bin/bash
# A/B builds of librtmi.so for same-device comparisons (devices differ by several per cent: two variants are only
# comparable inside ONE gpurun call).  Usage: tools/ab_build.sh <name> "<extra hipcc flags>" [more name/flag pairs]
# -> ray-tracing-cuda_amd/lib/librtmi_<name>.so; then on the GPU box: tools/ab_run.sh <name> <name> ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
while [ $# -ge 2 ]; do
  N=$1; F=$2; shift 2
  make -s -C $ROOT/ray-tracing-cuda_amd/csrc OUT=../lib/ab_$N OBJ=../lib/ab_$N/obj EXTRA="$F" -j4 >/dev/null
  cp $ROOT/ray-tracing-cuda_amd/lib/ab_$N/librtmi.so $ROOT/ray-tracing-cuda_amd/lib/librtmi_$N.so
  echo "built librtmi_$N.so ($F)"
done
