#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_r6_halo8.sh > gpurun_out/r6_halo8_stdout.txt 2>&1
O=gpurun_out/r6_hazards2.txt
timeout 600 python -m pytest tests/test_hazard_gpu.py -x -q -s 2>&1 | grep -E "HAZARD|passed|failed|skipped|Error" | tail -6 > $O
run() { echo "== $*" >> $O; timeout 900 python tools/check_hazards.py "$@" 2>&1 | grep -E "HAZARDS|Error|error|Traceback" | tail -3 >> $O; }
run clip 40 10 240 432
run multi 40 10 240 432
PP_HAZARD_STACKS=1 run stream 320 80 720 1280 0,1,2
cat gpurun_out/r6_halo8.txt | cut -c1-180
python - <<PY
import json
for line in open("gpurun_out/r6_hazards2.txt"):
    if line.startswith("HAZARDS "):
        r=json.loads(line[8:]); print({k:(v if not isinstance(v,(list,dict)) else (len(v) if k=="graphs" else [m[:700] for m in v[:5]])) for k,v in r.items()})
    else: print(line.rstrip()[:400])
PY
