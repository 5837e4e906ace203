cd $GRAFT_REPO_ROOT
O=gpurun_out/r4cv; mkdir -p $O
for ch in 0 1 2 4; do
  echo "== NRGBD_QUAD_CHUNKS=$ch" >> $O/chunks.txt
  NRGBD_QUAD_CHUNKS=$ch python tools/bench_costvol.py --config B --iters 300 --dev 2>&1 | grep -i "costvol" >> $O/chunks.txt
done
for c in S K H; do echo "== $c chunks default" >> $O/chunks.txt; python tools/bench_costvol.py --config $c --iters 300 --dev 2>&1 | grep -i "costvol" >> $O/chunks.txt; done
cat $O/chunks.txt
