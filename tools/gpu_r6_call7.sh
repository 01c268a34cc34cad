#!/bin/bash
O=gpurun_out/r6_call7.txt; : > $O
python -m pytest tests/test_ops_gpu.py tests/test_stress_gpu.py -m gpu -x -q -k "attention or documented_limits" 2>&1 | tail -3 >> $O
python tools/check_hazards.py eager 40 40 240 432 2>&1 | grep HAZARDS | python -c "import sys,json; d=json.loads(sys.stdin.read()[8:]); print('INTRA', d['intra'], 'CONV_INPLACE', d['conv_inplace'][:3], {k:d[k] for k in d if k.endswith('_count')})" >> $O 2>&1
python tools/bench_linear.py 2>/dev/null | grep LINEAR >> $O
python tools/bench_split.py --only "convc1|enc_1x1" 2>&1 | tail -20 >> $O
cat $O
