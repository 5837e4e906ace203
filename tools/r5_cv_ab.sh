# round 5: A/B of the sampling kernel's build knobs on one chip (product = run table on, atomics off)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5cv; mkdir -p $O; rm -f $O/ab.txt
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_parity_configs.py -q -k "cost or identity or golden" 2>&1 | tail -2 >> $O/ab.txt
for rep in 1 2; do
for lib in libnrgbd_hip.so libnrgbd_exp_cv_notable.so libnrgbd_exp_cv_atomic.so libnrgbd_exp_cv_both.so; do
  for c in B S; do echo "== $lib $c" >> $O/ab.txt; python tools/bench_costvol.py --config $c --iters 300 --only costvol+ --lib $lib 2>&1 | grep -i "costvol" >> $O/ab.txt; done
done; done
python tools/cv_trace.py --config B 2>&1 | grep -v amdgpu.ids > $O/ab_trace_B.txt
cat $O/ab.txt; head -14 $O/ab_trace_B.txt
