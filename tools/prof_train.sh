# rocprofv3 kernel stats of the training step (tools/bench_train.py); MIOpen find-mode pre-warmed by a first run
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/bench_train.py --iters 2 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python tools/bench_train.py --iters 6 > gpurun_out/prof_train.log 2>&1
tail -1 gpurun_out/prof_train.log
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
