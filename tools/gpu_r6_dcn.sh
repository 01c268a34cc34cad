#!/bin/bash
# round 6: deformable kernel with all corner reads of a tap in flight -- parity tests, then same-box A/B against the previous kernel
O=gpurun_out/r6_dcn.txt; : > $O
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "deform or dcn" 2>&1 | tail -3 >> $O
python -m pytest tests/test_stress_gpu.py -m gpu -x -q -k "fallback_counters or generator_window_720p_stress or flow_completion_chunk_720p_stress" 2>&1 | tail -3 >> $O
for r in 1 2; do
  echo "== new kernel (round $r)" >> $O; python tools/bench_dcn.py 2>/dev/null | grep "impl 90" >> $O
  echo "== previous kernel (round $r)" >> $O; PP_LIB_PATH=build/dcn_old/libpropainter_hip.so python tools/bench_dcn.py 2>/dev/null | grep "impl 90" >> $O
done
cat $O
