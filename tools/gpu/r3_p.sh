cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_p; mkdir -p $O
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_graph.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for CFG in "192 99" "192 0" "192 2" "192 4" "192 8" "0 0" "256 99" "256 2" "256 4" "256 8"; do set -- $CFG; echo "=== APGD_GEMM_BN=$1 APGD_GEMM_PW=$2" >> $O/gemm_pw.log; APGD_GEMM_BN=$1 APGD_GEMM_PW=$2 python tools/gemm_bench.py >> $O/gemm_pw.log 2>&1; done
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs --graph-train 1 > $O/bench_gt1.log 2>&1
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs --graph-train 0 > $O/bench_gt0.log 2>&1
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs --graph-train 1 > $O/bench_gt1b.log 2>&1
export TMPDIR=/tmp
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
for PW in 99 0; do
APGD_GEMM_BN=192 APGD_GEMM_PW=$PW rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc/pw$PW -o p -- python3 tools/gemm_pmc.py 50432 3072 768 > $O/pmc_pw$PW.log 2>&1
python tools/pmc_csv.py $O/pmc/pw$PW --filter gemm_nt > $O/pmc_pw$PW.txt 2>&1
done
APGD_GEMM_BN=256 APGD_GEMM_PW=0 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc/bn256 -o p -- python3 tools/gemm_pmc.py 50432 3072 768 > $O/pmc_bn256.log 2>&1
python tools/pmc_csv.py $O/pmc/bn256 --filter gemm_nt > $O/pmc_bn256.txt 2>&1
rm -rf $O/pmc
timeout 300 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --other-configs cfg3 > $O/oc_cfg3.log 2>&1
timeout 400 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --other-configs cfg4 > $O/oc_cfg4.log 2>&1
APGD_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 3 --warmup 4 --no-cpu-baseline --other-configs cfg5 > $O/oc_cfg5.log 2>&1
