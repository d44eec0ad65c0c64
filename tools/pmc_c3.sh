#!/bin/bash
# Instruction mix of the mesh-scene frame (run on the GPU box). usage: tools/pmc_c3.sh <out-dir-under-gpurun_out> <spp>
OUT=gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/a -- python3 tools/gpu_c3.py $2 10 > $OUT/a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/b -- python3 tools/gpu_c3.py $2 10 > $OUT/b.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "render_kernel" in k: print(k, dict(v))
PY
grep C3 $OUT/a.log
