#!/bin/bash
# round 6: capture-time hazard check of every multi-stream submission form (tools/check_hazards.py), small size then BASELINE config 4
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_hazards.txt
: > $O
run() { echo "== $*" >> $O; timeout 900 python tools/check_hazards.py "$@" 2>&1 | grep -E "HAZARDS|Error|error|Traceback" | tail -3 >> $O; }
timeout 900 python -m pytest tests/test_hazard_gpu.py -x -q -s 2>&1 | tail -5 >> $O
run eager 40 10 240 432
run clip 40 10 240 432
run stream 40 10 240 432 0,2
run stream 40 10 240 432 0,1,2
run multi 40 10 240 432
PP_HAZARD_STACKS=1 run stream 320 80 720 1280 0,1,2
cp $O gpurun_out/r6_hazards_full.txt; python - <<PY
import json,re
for line in open("gpurun_out/r6_hazards.txt"):
    if line.startswith("HAZARDS "):
        r=json.loads(line[8:]); print({k:(v if not isinstance(v,(list,dict)) else (len(v) if k=="graphs" else v[:4])) for k,v in r.items()})
    else: print(line.rstrip()[:400])
PY
