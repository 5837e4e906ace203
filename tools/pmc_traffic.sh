#!/bin/bash
# HBM-side traffic of the fused sampling kernel ON THE BENCH'S OWN WINDOWS: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
# separate passes (MI355X_MICROARCH.md §HBM: they do not fit one pass; never combined with trace domains) around
# `bench.py --no-graph` (eager launches, so every costvol dispatch of the frames is visible), for each config given.
#   bash tools/pmc_traffic.sh <outdir under gpurun_out> [configs...]      e.g.  bash tools/pmc_traffic.sh r2_traffic B S K H
# Result: gpurun_out/<outdir>/costvol_traffic.json — copy it to profiles/r6_costvol_traffic.json; bench.py reports it as
# roofline.traffic (per launch, FETCH_SIZE doubled as the guide prescribes for gfx950 16-B/lane streaming reads).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r2_traffic}
shift || true
CFGS=${@:-B}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  CMD="python $R/bench.py --config $c --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-live-traffic"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$c/fetch -o p -- $CMD > $OUT/$c.fetch.log 2>&1 || true
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$c/write -o p -- $CMD > $OUT/$c.write.log 2>&1 || true
done
python $R/tools/pmc_traffic_json.py $OUT $CFGS > $OUT/costvol_traffic.json
cat $OUT/costvol_traffic.json
