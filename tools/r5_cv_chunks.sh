cd $GRAFT_REPO_ROOT
O=gpurun_out/r5cv; mkdir -p $O; rm -f $O/chunks.txt
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_parity_configs.py -q -k "cost or identity or golden" 2>&1 | tail -2 >> $O/chunks.txt
for ch in 0 2 4; do
  echo "== B NRGBD_QUAD_CHUNKS=$ch" >> $O/chunks.txt
  NRGBD_QUAD_CHUNKS=$ch python tools/bench_costvol.py --config B --iters 300 --dev --only costvol 2>&1 | grep -i "costvol" >> $O/chunks.txt
done
for c in S K H; do echo "== $c chunks default" >> $O/chunks.txt; python tools/bench_costvol.py --config $c --iters 300 --only costvol 2>&1 | grep -i "costvol" >> $O/chunks.txt; done
NRGBD_QUAD_CHUNKS=2 python tools/cv_trace.py --config B 2>&1 | grep -v amdgpu.ids > $O/chunks_trace2.txt
cat $O/chunks.txt; grep -i "workgroups;\|residency\|wall clock" $O/chunks_trace2.txt
