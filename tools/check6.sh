cd $GRAFT_REPO_ROOT
O=gpurun_out/c6; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt
python bench.py --mode train 2>$O/bench.err | tail -1 | cut -c1-330 > $O/bench_train.txt
python bench.py --no-cpu-baseline 2>>$O/bench.err | tail -1 | cut -c95-330 > $O/bench_B.txt
cat $O/pytest.txt $O/bench_train.txt $O/bench_B.txt
