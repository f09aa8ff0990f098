#!/bin/bash
# full measurement pass for profiles/ (run on the GPU box: gpurun -- 'bash tools/measure.sh'): bench lines, rocprofv3 kernel
# stats of the same command, separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (never combined with trace domains)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/meas; rm -rf $O; mkdir -p $O
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench default done"
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/stats_run.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_write.err
echo "rocprof quadruped done"
python3 $R/bench.py --samples 512 --no-cpu-baseline --no-secondary --steps 30 > $O/bench_quadruped_512.json 2> $O/bench_512.err
python3 $R/bench.py --tier A --samples 512 --no-cpu-baseline --no-secondary --steps 30 > $O/bench_quadruped_512_tierA.json 2>> $O/bench_512.err
python3 $R/bench.py --samples 1024 --no-cpu-baseline --no-secondary --steps 20 > $O/bench_quadruped_1024.json 2>> $O/bench_512.err
rocprofv3 --kernel-trace --stats -d $O/stats512 -o s --output-format csv -- python3 $R/bench.py --samples 512 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2> $O/stats512.err
echo "dense tier done"
python3 $R/bench.py --workload humanoid --steps 10 --warmup 3 --no-secondary > $O/bench_humanoid.json 2> $O/bench_humanoid.err
python3 $R/bench.py --tier A --workload humanoid --steps 6 --warmup 2 --no-secondary --no-cpu-baseline > $O/bench_humanoid_tierA.json 2>> $O/bench_humanoid.err
rocprofv3 --kernel-trace --stats -d $O/stats_humanoid -o s --output-format csv -- python3 $R/bench.py --workload humanoid --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2> $O/stats_humanoid.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_humanoid -o p --output-format csv -- python3 $R/bench.py --workload humanoid --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_fetch_humanoid.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_humanoid -o p --output-format csv -- python3 $R/bench.py --workload humanoid --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_write_humanoid.err
echo "humanoid done"
python3 $R/bench.py --workload hand --samples 256 --steps 20 --warmup 3 --no-secondary > $O/bench_hand_256.json 2> $O/bench_hand.err
python3 $R/bench.py --workload hand --samples 2048 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > $O/bench_hand_2048.json 2>> $O/bench_hand.err
rocprofv3 --kernel-trace --stats -d $O/stats_hand -o s --output-format csv -- python3 $R/bench.py --workload hand --samples 256 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2> $O/stats_hand.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_hand -o p --output-format csv -- python3 $R/bench.py --workload hand --samples 256 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_fetch_hand.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_hand -o p --output-format csv -- python3 $R/bench.py --workload hand --samples 256 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_write_hand.err
python3 $R/tools/time_task.py fingers 60 > $O/fingers_time.log 2>&1
python3 $R/tools/time_task.py fingers 60 grasp=True >> $O/fingers_time.log 2>&1
python3 $R/tools/time_task.py fingers 256 grasp=True >> $O/fingers_time.log 2>&1
echo "fingers done"
find $O -name "*.csv" | head -40
tail -c 400 $O/bench_default.json
