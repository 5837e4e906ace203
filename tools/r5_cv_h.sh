cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_parity_configs.py -q -k "cost or identity or golden" 2>&1 | tail -1
for c in H S K B; do python tools/bench_costvol.py --config $c --iters 200 --only costvol+ 2>&1 | grep -i costvol | sed "s/^/$c /" ; done
bash tools/pmc_traffic.sh r5h H > /dev/null 2>&1; grep -A3 '"H"' gpurun_out/r5h/costvol_traffic.json
