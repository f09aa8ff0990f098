#!/bin/bash
# SQ PMC passes (rocprofv3 --pmc only, one counter group per run) for the rollout kernels: 512 candidates on the dense tier
# (two workgroups per CU) vs the same batch at full capacity only (one per CU).  gpurun -- 'bash tools/sq_counters.sh'
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq; rm -rf $O; mkdir -p $O
for tier in B A; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    if [ $tier = A ]; then T="--tier A"; else T=""; fi
    rocprofv3 --pmc $grp --kernel-trace -d $O/t${tier}_$i -o p --output-format csv -- python3 $R/bench.py --samples 512 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $T > /dev/null 2> $O/t${tier}_$i.err
  done
done
echo done
