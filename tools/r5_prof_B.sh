# kernel trace of the config-B frame (eager launches) + the K-Net first-layer alternatives
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5prof; mkdir -p $O
python tools/bench_knet_l0.py > $O/knet_l0.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_B -- python bench.py --config B --steps 8 --warmup 2 --no-cpu-baseline --no-graph --no-other-configs > $O/prof_B.log 2>&1
cp $(find $O/prof_B -name "*kernel_stats.csv" | head -1) $O/bench_B_kernel_stats.csv; rm -rf $O/prof_B
grep -v amdgpu.ids $O/knet_l0.txt
