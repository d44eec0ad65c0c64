#!/bin/bash
# Experiment (GPU box): can DistributedMain's RCCL path run with two ranks on ONE GPU?  (RCCL may
# refuse duplicate devices.)  Runs build/scenes/spheres as ranks 0 and 1 of a 2-rank job, both in
# pixel-tile mode and in the reference's sample-split mode, and compares with the 1-rank frame.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
EXE=$ROOT/ray-tracing-cuda_amd/build/scenes/spheres
OUT=$ROOT/gpurun_out/two_ranks; rm -rf $OUT; mkdir -p $OUT/one $OUT/r0 $OUT/r1 $OUT/s0 $OUT/s1
export RT_HEIGHT=64 RT_WIDTH=96 RT_SPP=4 HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=WARN
(cd $OUT/one && RT_DUMP=$OUT/one/frame.bin timeout -k 5 120 $EXE > log.txt 2>&1) || { echo "1-rank run failed"; tail -3 $OUT/one/log.txt; exit 1; }
run2() {  # $1 = dir prefix, $2 = extra env
  local port=$((20000 + RANDOM % 20000))
  for r in 0 1; do
    (cd $OUT/$1$r && env $2 RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port RT_RUN_ID=t$$_$1 RT_DUMP=$OUT/$1$r/frame.bin timeout -k 5 120 $EXE > log.txt 2>&1) &
    pids[$r]=$!
  done
  local rc=0
  for r in 0 1; do wait ${pids[$r]} || rc=1; done
  return $rc
}
run2 r "RT_UNUSED=1"; echo "tile mode rc=$?"; tail -2 $OUT/r0/log.txt
run2 s "RT_DIST_MODE=spp"; echo "spp mode rc=$?"; tail -2 $OUT/s0/log.txt
python3 - <<PY
import numpy as np, os
one = np.fromfile("$OUT/one/frame.bin", dtype=np.float32)
for d in ("r0", "s0"):
    p = "$OUT/%s/frame.bin" % d
    if os.path.exists(p):
        f = np.fromfile(p, dtype=np.float32)
        print(d, "frame equals the 1-rank frame:", bool(f.size == one.size and np.array_equal(f, one)))
    else:
        print(d, "no frame")
PY
