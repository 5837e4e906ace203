cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_knet.py tests/test_gpu_net.py -q 2>&1 | tail -3
for c in B S; do echo "$c fused: $(timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-200)"; echo "$c separate: $(NRGBD_KNET_BN=separate timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-200)"; done
