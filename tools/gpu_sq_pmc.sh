#!/bin/bash
# Where do the waves of the halo kernel spend their cycles?  SQ wave-state counters of tools/kbench launches (C++, no torch: seconds),
# two rocprofv3 --pmc passes (8 SQ counter slots each) over the shipped kernel (impl 70) and the software-pipelined one (impl 116):
#   SQ_WAVE_CYCLES = wave residency, SQ_WAIT_ANY = parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stalls (dependencies / busy pipes),
#   SQ_ACTIVE_INST_ANY = issuing; SQ_WAIT_INST_LDS = LDS issue stalls; SQ_ACTIVE_INST_{VALU,LDS,VMEM,SCA,MISC}: issue cycles by class;
#   SQ_VALU_MFMA_BUSY_CYCLES = matrix pipe busy; SQ_INST_CYCLES_VMEM: cycles VMEM instructions take to issue.
R=$(pwd)
K=$R/build/kbench
OUT=$R/gpurun_out/sq_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHAPE="${SHAPE:-conv 16 90 160 3 3 256 256 --act 1}"
rocprofv3 -L > $R/gpurun_out/rocprofv3_counters.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d $OUT/p1 --output-format csv -- $K $SHAPE --impls 70,116 --reps 10 > $OUT/p1.log 2>&1
echo "pass 1 exit $?"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/p2 --output-format csv -- $K $SHAPE --impls 70,116 --reps 10 > $OUT/p2.log 2>&1
echo "pass 2 exit $?"
cd $R
python tools/pmc_summary.py $OUT all | grep -v "^   .*n=   0" > gpurun_out/${TAG:-r3v}_sq_wave_states.txt
cat gpurun_out/${TAG:-r3v}_sq_wave_states.txt | head -80
tail -3 $OUT/p1.log $OUT/p2.log
find $OUT -name "*.csv" -size +4M -delete 2>/dev/null
