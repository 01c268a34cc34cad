#!/bin/bash
# round 6: variant library whose halo kernels fill their whole LDS allocation with fp16 NaNs before staging anything -- does any fragment read depend on what
# the previous tenant of the LDS left behind?  Parity tests of the layers and modules that go through the halo kernel, against the oracle.
O=gpurun_out/r6_poison.txt; : > $O
export PP_LIB_PATH=build/poison/libpropainter_hip.so
python -m pytest tests/test_ops_gpu.py -m gpu -q -k "conv" 2>&1 | tail -4 >> $O
python -m pytest tests/test_split_plane_gpu.py tests/test_halo8_gpu.py -m gpu -q -x 2>&1 | tail -3 >> $O
python -m pytest tests/test_modules_gpu.py -m gpu -q -k "generator or flow_completion or raft" 2>&1 | tail -4 >> $O
python -m pytest tests/test_stress_gpu.py -m gpu -q -k "config3_timed" 2>&1 | tail -3 >> $O
cat $O
