cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_d; mkdir -p $O
python -m pytest tests/test_gpu_graph.py -x -q > $O/pytest_graph.log 2>&1; echo "rc=$?" >> $O/pytest_graph.log
for C in 384 192 96; do HW=$((5376/C)); APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_abl.so python tools/blk_trace.py --C $C --hw $HW >> $O/trace.log 2>&1; done
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs --graph 1 > $O/bench_g1.log 2>&1
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs --graph 0 > $O/bench_g0.log 2>&1
python tools/host_profile.py > $O/host_profile.log 2>&1
export TMPDIR=/tmp
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for C in 96 192 384; do HW=$((5376/C)); W=fwd,hpre; [ $C = 96 ] && W=fwd,bwd_in; 
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc/c$C -o p -- python3 tools/mlp_bench.py --C $C --hw $HW --what $W --iters 6 > $O/pmc_c$C.log 2>&1; done
python tools/pmc_csv.py $O/pmc/c96 $O/pmc/c192 $O/pmc/c384 --filter blk_mlp > $O/pmc_summary.txt 2>&1
rm -rf $O/pmc
