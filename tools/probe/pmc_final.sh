#!/bin/bash
# PMC counters of the fused LN+MLP kernels at the end of the round (one rocprofv3 --pmc pass per width; summarised by tools/pmc_csv.py)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/pmc_final/c96 -o p -- python3 tools/mlp_bench.py --C 96 --hw 56 --what fwd,bwd_in --iters 6 > gpurun_out/pmc_final_c96.log 2>&1
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/pmc_final/c192 -o p -- python3 tools/mlp_bench.py --C 192 --hw 28 --what fwd,bwd_in,hpre --iters 6 > gpurun_out/pmc_final_c192.log 2>&1
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/pmc_final/c384 -o p -- python3 tools/mlp_bench.py --C 384 --hw 14 --what fwd,hpre --iters 6 > gpurun_out/pmc_final_c384.log 2>&1
python tools/pmc_csv.py gpurun_out/pmc_final/c96 gpurun_out/pmc_final/c192 gpurun_out/pmc_final/c384 --filter blk_mlp > gpurun_out/pmc_final_summary.txt 2>&1
cat gpurun_out/pmc_final_summary.txt
