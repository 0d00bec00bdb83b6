cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ptrain; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/train -o train -- python $R/bench.py --mode train --no-cpu-baseline --steps 30 --warmup 6 --traffic off > $O/train.json 2> $O/train.err
cd $R
python tools/prof_summary.py $(find $O/train -name "*.db" | head -1) > $O/train_kernel_stats.txt
rm -rf $O/train
head -60 $O/train_kernel_stats.txt | cut -c1-150
