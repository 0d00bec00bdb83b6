#!/bin/bash
# All profiling evidence of round 2 in one GPU call: kernel stats of the default bench, of the train step, and the PMC passes.
# Summaries land in gpurun_out/r02/ (copied into profiles/ afterwards).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats -d $O/train -o train -- python $R/bench.py --mode train --no-cpu-baseline --steps 30 --warmup 6 > $O/train_under_rocprof.json 2> $O/train.err
rocprofv3 --kernel-trace -d $O/trace -o trace -- python $R/tools/pmc_workload.py > /dev/null 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o fetch -- python $R/tools/pmc_workload.py > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o write -- python $R/tools/pmc_workload.py > /dev/null 2> $O/write.err
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/prof_summary.py $(find $O/bench -name "*.db" | head -1) > $O/r02_bench_default_kernel_stats.txt
python tools/prof_summary.py $(find $O/train -name "*.db" | head -1) > $O/r02_train_step_kernel_stats.txt
python tools/pmc_report.py $(find $O/fetch -name "*.db" | head -1) $(find $O/write -name "*.db" | head -1) $(find $O/trace -name "*.db" | head -1) $O r02
cp $(find $O/bench -name "*kernel_stats.csv" | head -1) $O/r02_bench_default_kernel_stats.csv 2>/dev/null; cp $(find $O/train -name "*kernel_stats.csv" | head -1) $O/r02_train_step_kernel_stats.csv 2>/dev/null
rm -rf $O/bench $O/train $O/trace $O/fetch $O/write 2>/dev/null
ls $O
