cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_k; mkdir -p $O
export TMPDIR=/tmp
for shp in "50432 3072 768" "50176 1536 384"; do tag=$(echo $shp | tr ' ' '_')
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1_$tag -o p -- python3 tools/gemm_pmc.py $shp > $O/p1_$tag.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2_$tag -o p -- python3 tools/gemm_pmc.py $shp > $O/p2_$tag.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p3_$tag -o p -- python3 tools/gemm_pmc.py $shp > $O/p3_$tag.log 2>&1
python tools/pmc_csv.py $O/p1_$tag $O/p2_$tag $O/p3_$tag > $O/pmc_$tag.txt 2>&1
rm -rf $O/p1_$tag $O/p2_$tag $O/p3_$tag
done
