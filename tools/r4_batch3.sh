cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist.py -q -x -s 2>&1 | grep -v amdgpu.ids | tail -25 > $O/tests.txt
python bench.py --mode train --no-cpu-baseline 2>$O/train4.err | tail -1 > $O/bench_train_accum4.json
python bench.py --mode train --accum 1 --no-cpu-baseline 2>$O/train1.err | tail -1 > $O/bench_train_accum1.json
python bench.py --no-cpu-baseline --no-live-traffic 2>$O/bench.err | tail -1 > $O/bench_B.json
cat $O/tests.txt; cut -c1-900 $O/bench_train_accum4.json; echo; cut -c1-400 $O/bench_train_accum1.json; echo; tail -3 $O/train4.err; python - <<'P'
import json
d=json.load(open('gpurun_out/r4b3/bench_B.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_mfma']['kernel_ms'], d['roofline_mfma']['frac'])
P
