# round 5: one step of the sampling-kernel work = its parity tests + timing at B / S / K / H + the developer build's phase clocks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5cv; mkdir -p $O
TAG=${1:-step}
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_parity_configs.py -q -k "cost or identity or golden" 2>&1 | tail -8 > $O/${TAG}_tests.txt
for c in B S K H; do python tools/bench_costvol.py --config $c --iters 200 --only costvol+ 2>&1 | grep -i costvol | sed "s/^/$c /" ; done > $O/${TAG}_time.txt
python tools/cv_trace.py --config B 2>&1 | grep -v amdgpu.ids > $O/${TAG}_trace_B.txt
python tools/cv_trace.py --config S 2>&1 | grep -v amdgpu.ids > $O/${TAG}_trace_S.txt
cat $O/${TAG}_tests.txt $O/${TAG}_time.txt $O/${TAG}_trace_B.txt
