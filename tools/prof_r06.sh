#!/bin/bash
# Profiling evidence of round 5 in one GPU call: kernel stats of the default bench and of the train step, the PMC passes
# (dominant kernels + irregular ops) and the whole-train-step traffic.  Summaries land in gpurun_out/r06/ (copied into profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- python $R/bench.py --no-cpu-baseline --no-configs --traffic profiles --seconds 2 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats -d $O/train -o train -- python $R/bench.py --mode train --no-cpu-baseline --steps 30 --warmup 6 --traffic profiles > $O/train_under_rocprof.json 2> $O/train.err
cd $R
python tools/prof_summary.py $(find $O/bench -name "*.db" | head -1) > $O/r06_bench_default_kernel_stats.txt
python tools/prof_summary.py $(find $O/train -name "*.db" | head -1) > $O/r06_train_step_kernel_stats.txt
if [ "$1" == "full" ]; then
  cd /tmp
  rocprofv3 --kernel-trace -d $O/trace -o trace -- python $R/tools/pmc_workload.py > /dev/null 2> $O/trace.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o fetch -- python $R/tools/pmc_workload.py > /dev/null 2> $O/fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o write -- python $R/tools/pmc_workload.py > /dev/null 2> $O/write.err
  cd $R
  python tools/pmc_report.py $(find $O/fetch -name "*.db" | head -1) $(find $O/write -name "*.db" | head -1) $(find $O/trace -name "*.db" | head -1) $O r06
  python tools/pmc_train_total.py $O/r06_pmc_train_total.json > $O/r06_pmc_train_total.txt 2>&1
  python bench.py --evidence r06 > $O/r06_bench_default.json 2> $O/bench_default.err
  cp profiles/r06_insitu_cost_volume.txt $O/
  python bench.py --mode train > $O/r06_bench_train.json 2> $O/bench_train.err
fi
rm -rf $O/bench $O/train $O/trace $O/fetch $O/write 2>/dev/null
ls $O
