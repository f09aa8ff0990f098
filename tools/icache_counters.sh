#!/bin/bash
# instruction-cache PMC pass of the C2 bench (run on the GPU box: gpurun -- 'bash tools/icache_counters.sh [lib.so]'): hits / misses / requests of the
# shader instruction cache and the wave-level fetch / wait counters, one --pmc group per run (never combined with trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/icache; rm -rf $O; mkdir -p $O
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $O/$tag -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/$tag.err
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "rollout_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(acc.items()):
        print(f"{k:32s} per launch {v / max(n, 1):.4g}  (launches {n})")
PY
