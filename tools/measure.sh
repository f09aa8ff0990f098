#!/bin/bash
# full measurement pass for profiles/: bench (default), rocprofv3 kernel stats, PMC traffic passes
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/meas; rm -rf $O; mkdir -p $O
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/stats_run.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
python3 $R/bench.py --workload humanoid --steps 10 --warmup 3 > $O/bench_humanoid.json 2> $O/bench_humanoid.err
find $O -name "*.csv" | head -20
tail -c 600 $O/bench_default.json
