#!/bin/bash
timeout 600 python -m pytest tests/test_split_plane_gpu.py -m gpu -q -p no:cacheprovider -s -k "on_the_fly" 2>&1 | grep -E "passed|failed|Error" | tail -5
for d in 0 16 0; do PP_OTF_DBG=$d python tools/bench_otf.py --pairs 79 --reps 5 2>&1 | grep OTF_; done
python tools/bench_otf.py --pairs 79 --reps 5 --flow zoom 2>&1 | grep OTF_
