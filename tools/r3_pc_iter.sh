cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_knet.py tests/test_gpu_cnn.py tests/test_gpu_rnet.py tests/test_gpu_train.py -q 2>&1 | tail -3
timeout 300 python tools/bench_wino.py --config B --cnn 2>&1 | grep -v amdgpu | tail -14
for c in B S; do echo "$c: $(timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-200)"; done
