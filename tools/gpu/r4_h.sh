cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_h; mkdir -p $O
./tools/probe/epi_probe > $O/epi_probe.log 2>&1
./tools/probe/epi_probe >> $O/epi_probe.log 2>&1
