cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_parity_configs.py tests/test_gpu_train.py -q -k "cost or identity or golden" 2>&1 | tail -2
bash tools/r5_cv_ab.sh libnrgbd_exp_cv_rmw.so | grep -A1 " B\| S\| H"
bash tools/pmc_traffic.sh r5lds B > /dev/null 2>&1; grep -A3 '"B"' gpurun_out/r5lds/costvol_traffic.json
