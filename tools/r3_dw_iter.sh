# DW kernel iteration check: correctness of the dw tests + layer timing (product lib) + ablation (dev lib)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_knet.py -k "dw" -q 2>&1 | tail -2
timeout 300 python tools/bench_wino.py --config B 2>&1 | grep "wino-pc\|wino-dw"
timeout 300 python tools/abl_dw.py ${ABL:-0 1 2} 2>&1 | grep -v amdgpu
