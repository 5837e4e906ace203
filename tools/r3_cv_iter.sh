# costvol generation 4 check: parity tests + timing of generations 3 / 4 (and the 2-WG/CU big-patch form on the dev library)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_parity_configs.py -q -k "costvol or generation" 2>&1 | tail -3
for c in B S K H; do for g in quad quad4; do echo "config $c gen $g: $(timeout 120 python tools/bench_costvol.py --config $c --only costvol --gen $g 2>/dev/null | tr '\n' ' ')"; done; done
echo "dev, 188-texel patch at 2 WG/CU:"; NRGBD_QUAD4=2 timeout 120 python tools/bench_costvol.py --dev --config B --only costvol --gen quad4 2>/dev/null
NRGBD_QUAD4=2 timeout 120 python tools/bench_costvol.py --dev --config H --only costvol --gen quad4 2>/dev/null
