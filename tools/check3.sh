cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_knet.py tests/test_gpu_cnn.py -m gpu -q -x -k "wino" 2>&1 | tail -2 > $O/abl14.txt
for a in 64; do echo "== NRGBD_WINO_ABL=$a"; NRGBD_WINO_ABL=$a timeout 120 python tools/bench_wino.py --dev --config B 2>&1 | grep -E "wino-pc|clock64|publish"; done >> $O/abl14.txt
timeout 120 python tools/bench_wino.py --config B --cnn 2>&1 | grep -E "wino-pc|conv2d" >> $O/abl14.txt
cat $O/abl14.txt
