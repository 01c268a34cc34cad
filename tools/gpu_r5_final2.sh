#!/bin/bash
# after the last kernel change of round 5: the split-plane / RAFT tests, the PMC traffic passes at the final sources, a full default bench
export COMMIT=${COMMIT:-unknown} COMMIT_TIME=${COMMIT_TIME:-0} RAFT_DTYPE=f16x3
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
R=$(pwd); cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --single-pass --window-streams 1 --raft-streams 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch --output-format csv -- $CMD > $R/gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write --output-format csv -- $CMD > $R/gpurun_out/pmc_write.log 2>&1; echo "pmc write exit $?"
cd $R
python tools/rocprof_summary.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r5zz_hbm_traffic_720p.json "$COMMIT" "$COMMIT_TIME" f16x3 | head -6
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +8M -delete 2>/dev/null
cp gpurun_out/r5zz_hbm_traffic_720p.json profiles/        # (so that the bench below already finds the profile of these sources)
timeout 800 python bench.py > gpurun_out/r5zz_bench_720p.json 2> gpurun_out/r5zz.err
python -c "
import json; d=json.load(open('gpurun_out/r5zz_bench_720p.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d['roofline']['traffic'], d['roofline']['avg_launch_us']); print(d['stress']['value'], d['parity']['psnr_db'], d['parity']['max_abs'], d['configs']['c2_432x240x80']['value']); print({k: round(v['ms'],1) for k,v in d['kernels'].items() if v['ms']>10})"
