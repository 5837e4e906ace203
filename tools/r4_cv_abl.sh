cd $GRAFT_REPO_ROOT
O=gpurun_out/r4cv; mkdir -p $O; rm -f $O/abl.txt
for a in 0 1 2 64 16 32; do
  echo "== NRGBD_ABLATE=$a" >> $O/abl.txt
  NRGBD_ABLATE=$a python tools/bench_costvol.py --config B --iters 300 --dev --only costvol 2>&1 | grep -i "costvol" >> $O/abl.txt
done
cat $O/abl.txt
