cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_rnet.py tests/test_gpu_cnn.py -q 2>&1 | tail -4
for c in B S K H; do echo "$c: $(timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-200)"; done
echo "A/B direct R-Net layers:"; for c in B S; do echo "$c: $(NRGBD_RNET_WINO=0 timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-200)"; done
echo "A/B vendor R-Net:"; for c in S K; do echo "$c: $(NRGBD_RNET=vendor timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | cut -c95-200)"; done
