cd $GRAFT_REPO_ROOT
O=gpurun_out/c6; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt
python bench.py --no-cpu-baseline 2>$O/bench.err | tail -1 | cut -c95-330 > $O/bench_B.txt
NRGBD_CNN_SMALL=vendor python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-330 > $O/bench_B_vendor_small.txt
python bench.py --config S --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-330 > $O/bench_S.txt
NRGBD_CNN_SMALL=vendor python bench.py --config S --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-330 > $O/bench_S_vendor_small.txt
cat $O/pytest.txt $O/bench_B.txt $O/bench_B_vendor_small.txt $O/bench_S.txt $O/bench_S_vendor_small.txt
