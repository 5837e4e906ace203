# round 3, GPU call 2: wino_dw.hip correctness + timing, re-run of the tests that failed in call 1, frame time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_knet.py -k "dw" -q -s 2>&1 | grep -v "amdgpu.ids" > $O/dw_tests.txt; echo "dw tests rc=$?" >> $O/dw_tests.txt
grep "parity\|passed\|failed\|Error\|rc=" $O/dw_tests.txt | tail -30
timeout 300 python tools/bench_wino.py --config B > $O/bench_wino_B.txt 2>&1; cat $O/bench_wino_B.txt | grep -v amdgpu
timeout 300 python tools/bench_wino.py --config S > $O/bench_wino_S.txt 2>&1; cat $O/bench_wino_S.txt | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py -q 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_B.json; cut -c1-330 $O/bench_B.json
NRGBD_KNET=wino2 timeout 600 python bench.py --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_B_gen2.json; cut -c90-330 $O/bench_B_gen2.json
NRGBD_KNET_DW=16,64 timeout 600 python bench.py --no-cpu-baseline 2>>$O/bench.err | tail -1 > $O/bench_B_dw16.json; cut -c90-330 $O/bench_B_dw16.json
