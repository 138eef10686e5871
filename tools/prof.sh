#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of the driver's bench command (--steps 20 --warmup 5), then two
# PMC passes (HBM bytes) of the same command (the fixed-h leg shortened to 10 steps: it is not the region the roofline describes).
# --agg-tile 0: the aggregated leg runs other grids (256^2, 2048^2) and would mix their launches into the per-kernel averages.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${PROF_TAG:-prof_r1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --agg-tile 0 --steps 20 --warmup 5 > $OUT/bench_under_rocprof.log 2>&1
echo "trace rc=$?" >> $OUT/bench_under_rocprof.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --agg-tile 0 --steps 20 --warmup 5 --fixed-steps 10 > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?" >> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --agg-tile 0 --steps 20 --warmup 5 --fixed-steps 10 > $OUT/pmc_write.log 2>&1
echo "write rc=$?" >> $OUT/pmc_write.log
find $OUT -name "*.csv" | head -20
