#!/bin/bash
# Runs the whole GPU test-suite without stopping at the first failure and leaves a report under gpurun_out/.
mkdir -p gpurun_out
python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.txt 2>&1
python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -80 gpurun_out/pytest_gpu.log
