# full check after wiring the Winograd kernel into the K-Net and the feature CNN
cd $GRAFT_REPO_ROOT
O=gpurun_out/c4; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt
python bench.py --no-cpu-baseline 2>$O/bench.err | tail -1 | cut -c1-330 > $O/bench_B.txt
for c in S K; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-330 > $O/bench_$c.txt; NRGBD_CNN=mfma python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-330 > $O/bench_${c}_mfma.txt; done
timeout 200 python tools/bench_cnn.py --config B 2>&1 | tail -3 > $O/cnn_B.txt
timeout 200 python tools/bench_cnn.py --config S 2>&1 | tail -2 > $O/cnn_S.txt
cat $O/pytest.txt $O/bench_B.txt $O/bench_S.txt $O/bench_S_mfma.txt $O/bench_K.txt $O/bench_K_mfma.txt $O/cnn_B.txt $O/cnn_S.txt
