#!/bin/bash
# One GPU session: parity suite, headline bench (720p), rocprofv3 kernel trace of the same command, HBM-traffic PMC
# passes.  Everything lands under gpurun_out/; the summaries worth keeping are copied to profiles/ by hand.
TAG=${1:-r1}
mkdir -p gpurun_out
R=$(pwd)
python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0)); import os; print('cores', len(os.sched_getaffinity(0)))" > gpurun_out/env.txt 2>&1
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  tail -30 gpurun_out/pytest_gpu.log
fi
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --steps 2 --warmup 1 $BENCH_ARGS > gpurun_out/bench_720.json 2> gpurun_out/bench_720.err
echo "720p exit $?"; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_720.json'))
    print({k:d[k] for k in ('value','ms_per_step','roofline','stages_ms','cpu_baseline')})
    for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms']):
        print(f"{k:26s} n={v['launches']:6d} ms={v['ms']:9.2f} avg_us={v['avg_us']:8.1f} TF={v['tflops']:8.1f} GB/s={v['gbs']:8.1f}")
except Exception as e:
    print('bench parse failed', e)
PY
tail -5 gpurun_out/bench_720.err
if [ -z "$SKIP_PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  CMD="python $R/bench.py --eager --steps 1 --warmup 0 --no-cpu-baseline --no-profile $BENCH_ARGS"
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace --output-format csv -- $CMD > $R/gpurun_out/prof_trace.log 2>&1
  echo "trace exit $?"
  CMD1="python $R/bench.py --eager --steps 1 --warmup 0 --no-cpu-baseline --no-profile $BENCH_ARGS"
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch --output-format csv -- $CMD1 > $R/gpurun_out/pmc_fetch.log 2>&1
  echo "pmc fetch exit $?"
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write --output-format csv -- $CMD1 > $R/gpurun_out/pmc_write.log 2>&1
  echo "pmc write exit $?"
  cd $R
  python tools/rocprof_summary.py stats gpurun_out/prof_trace gpurun_out/${TAG}_rocprof_kernel_stats_720p.md | head -40
  python tools/rocprof_summary.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/${TAG}_hbm_traffic_720p.json
  # keep the merge-back small: raw traces can be hundreds of MB
  find gpurun_out/prof_trace gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +8M -delete
  du -sh gpurun_out
fi
