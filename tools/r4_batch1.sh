# round 4, GPU batch 1: serpentine channel-block order A/B, in-frame gap experiment, design probe, frame time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_knet.py -q -x 2>&1 | grep -v amdgpu.ids | tail -5 > $O/knet_tests.txt
for c in B S H; do
  echo "== config $c: product (serpentine)" >> $O/serp_ab.txt
  python tools/bench_wino.py --config $c --iters 20 --only wino-dw 2>&1 | grep -v amdgpu.ids >> $O/serp_ab.txt
  echo "== config $c: NRGBD_DW_SERP=0" >> $O/serp_ab.txt
  NRGBD_DEV_LIB=$GRAFT_REPO_ROOT/neuralrgbd_amd/csrc/libnrgbd_exp_noserp.so python tools/bench_wino.py --config $c --iters 20 --only wino-dw --dev 2>&1 | grep -v amdgpu.ids >> $O/serp_ab.txt
done
./tools/probes/wino_design_probe > $O/design_probe.txt 2>&1
timeout 600 python tools/inframe_gap.py r4b1 > $O/inframe_gap.log 2>&1
python bench.py --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 > $O/bench_B.json
bash tools/pmc_wino.sh r4b1/pmc_wino B wino-dw > /dev/null 2>&1
rm -rf $O/pmc_wino/pmc?
cat $O/knet_tests.txt $O/serp_ab.txt $O/design_probe.txt; tail -30 $O/inframe_gap.log; cut -c1-300 $O/bench_B.json; cat $O/pmc_wino/summary.txt
