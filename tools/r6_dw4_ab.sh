# same-chip A/B of wino_dw4.hip build variants: product library vs neuralrgbd_amd/csrc/libnrgbd_exp_<name>.so
cd $GRAFT_REPO_ROOT
python tools/bench_dw4.py B 2>&1 | grep "dw4" | tail -3
for v in "$@"; do echo "variant $v"; NRGBD_EXP_LIB=neuralrgbd_amd/csrc/libnrgbd_exp_$v.so python tools/bench_dw4.py B 2>&1 | grep "dw4" | tail -3; done
