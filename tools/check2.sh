# kernel-level check: new Winograd kernel tests + A/B timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_knet.py tests/test_gpu_cnn.py -m gpu -q -x -k "wino" -s 2>&1 | grep -E "parity|passed|failed|Error|error|assert" | head -60 > $O/pytest.txt
timeout 300 python tools/bench_wino.py --config B --cnn > $O/bench_wino_B.txt 2>&1
timeout 120 python tools/bench_wino.py --config S > $O/bench_wino_S.txt 2>&1
cat $O/pytest.txt; cat $O/bench_wino_B.txt $O/bench_wino_S.txt
