cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_net.py tests/test_gpu_parity_configs.py -q -s 2>&1 | grep "spp_concat\|passed\|failed\|Error" | tail -5
for c in B S; do echo "$c: $(timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-200)"; echo "$c unfused: $(NRGBD_SPP=torch timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-200)"; done
