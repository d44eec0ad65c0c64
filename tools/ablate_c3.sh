#!/bin/bash
# Diagnostic (GPU box): time + VALU/SALU counts of the C3 frame for diagnostic builds of librtmi
# (ray-tracing-cuda_amd/lib/librtmi_<name>.so, built with EXTRA=-DRTMI_ABLATE=n; "base" = the product build).
# usage: tools/ablate_c3.sh <out-tag> name [name ...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = base ]; then unset RTMI_LIB_PATH; else export RTMI_LIB_PATH=$ROOT/ray-tracing-cuda_amd/lib/librtmi_$n.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/$n -- python3 $ROOT/tools/gpu_c3.py 512 10 > $OUT/$n.log 2>&1 || exit 1
  python3 - $OUT/$n $n <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); ms = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"]) / 2  # two launches
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]: ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print(sys.argv[2], "kernel_ms %.1f" % ms, {k: "%.2fG" % (v / 1e9) for k, v in sorted(acc.items())})
PY
  grep C3 $OUT/$n.log | tail -1
done
