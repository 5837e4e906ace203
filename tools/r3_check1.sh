# round 3, GPU call 1: parity work (pose inverse, true-grid tests, export bytes, d_candi_new) + bench B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $O/pytest_full.txt
tail -25 $O/pytest_full.txt > $O/pytest.txt
grep "\[parity\] config [BH]\|pose\|export u16" $O/pytest_full.txt > $O/parity_lines.txt
python bench.py 2>$O/bench.err | tail -1 > $O/bench_B.json
tail -8 $O/pytest.txt; cat $O/parity_lines.txt | tail -30; cut -c1-600 $O/bench_B.json
python -c "
import json; d=json.load(open('$O/bench_B.json')); print(json.dumps(d.get('parity'),indent=0)[:1500])"
