# round 5: A/B of the sampling kernel's build variants on one chip:  bash tools/r5_cv_ab.sh libA.so libB.so ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5cv; mkdir -p $O; rm -f $O/ab.txt
for rep in 1 2; do
for lib in libnrgbd_hip.so "$@"; do
  for c in B S H; do echo "== $lib $c" >> $O/ab.txt; python tools/bench_costvol.py --config $c --iters 100 --only costvol+ --lib $lib 2>&1 | grep -i "costvol" >> $O/ab.txt; done
done; done
cat $O/ab.txt
