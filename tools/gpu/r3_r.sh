cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_r; mkdir -p $O
timeout 200 python tools/cfg_step.py vit_b 224 256 2 --graph 0 --dump-after 90 > $O/vit_g0.log 2>&1
timeout 200 python tools/cfg_step.py vit_b 224 256 2 --graph 1 --graph-train 0 --dump-after 90 > $O/vit_g1_t0.log 2>&1
APGD_ATTACK_STREAMS=1 timeout 200 python tools/cfg_step.py vit_b 224 256 2 --graph 1 --graph-train 0 --dump-after 90 > $O/vit_g1_t0_s1.log 2>&1
timeout 200 python tools/cfg_step.py vit_b 224 256 2 --graph 1 --graph-train 1 --dump-after 90 > $O/vit_g1_t1.log 2>&1
for rep in 1 2; do
echo "=== w4=0 $rep" >> $O/gemm_w4.log; APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_w4.so APGD_GEMM_W4=0 python tools/gemm_bench.py >> $O/gemm_w4.log 2>&1
echo "=== w4=1 $rep" >> $O/gemm_w4.log; APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_w4.so APGD_GEMM_W4=1 python tools/gemm_bench.py >> $O/gemm_w4.log 2>&1
echo "=== nta $rep" >> $O/gemm_w4.log; APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_nta.so python tools/gemm_bench.py >> $O/gemm_w4.log 2>&1
done
APGD_HIP_LIB=$PWD/revisiting-at_amd/libapgd_w4.so APGD_GEMM_W4=1 python -m pytest tests/test_gpu_gemm.py -x -q > $O/pytest_w4.log 2>&1
