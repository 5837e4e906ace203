# round 5: where the fused sampling kernel's time goes (candidate ranges, per-phase clocks of the developer build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5cv; mkdir -p $O; rm -f $O/diag.txt
for sl in 0:64 0:4 0:8 8:16 16:32 32:64 0:16 16:64; do
  echo "== candidates $sl" >> $O/diag.txt
  python tools/bench_costvol.py --config B --iters 200 --only costvol+ --dslice $sl 2>&1 | grep -i "costvol" >> $O/diag.txt
done
echo "== one view" >> $O/diag.txt
python tools/bench_costvol.py --config B --iters 200 --only costvol+ --views 1 2>&1 | grep -i costvol >> $O/diag.txt
python tools/cv_trace.py --config B > $O/trace_B.txt 2>&1
python tools/cv_trace.py --config S > $O/trace_S.txt 2>&1
cat $O/diag.txt $O/trace_B.txt
