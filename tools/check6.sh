cd $GRAFT_REPO_ROOT
O=gpurun_out/c6; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_train.py -m gpu -q -x -s -k "native_vs_vendor" 2>&1 | grep -E "parity|assert|Error|passed|failed" | head -8 > $O/pytest.txt
timeout 300 python tools/bench_train.py --graph --iters 8 2>&1 | tail -2 > $O/g_native.txt
NRGBD_TRAIN_CONV=vendor timeout 300 python tools/bench_train.py --graph --iters 8 2>&1 | tail -2 > $O/g_vendor.txt
cat $O/pytest.txt $O/g_native.txt $O/g_vendor.txt
