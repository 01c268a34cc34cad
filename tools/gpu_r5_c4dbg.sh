#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_streaming_c4_variants.txt
: > $O
for v in "PP_DIAG_LANES=1" "PP_SG_STAGES=0,2" ; do
  echo "== $v" >> $O
  env $v timeout 500 python tools/bench_streaming.py --steps 1 --debug-stages --only-pipelined 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps({k:v for k,v in d.items() if k in ('pipelined_replay_to_replay','pipelined_vs_eager','pipelined_single_graph')})); print([ (r['rank'], r['completed_flows'], r['composited_bytes_differ']) for r in d['pipelined_stage_deviation_vs_eager']])" >> $O
done
cat $O
