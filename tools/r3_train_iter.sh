cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -q 2>&1 | tail -3
for m in vendor native; do echo "$m: $(NRGBD_TRAIN_CONV=$m timeout 300 python bench.py --mode train --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c100-230)"; done
