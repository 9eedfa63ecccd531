export AHIP_NO_ARITH_SELFCHECK=1
bash pair_allegro_amd/tools/pmc_mem.sh r06_lx2_41k "--config 5 --ncell 24" > /dev/null 2>&1
bash pair_allegro_amd/tools/pmc_mem.sh r06_lx_41k "--config 6 --ncell 24" > /dev/null 2>&1
echo lx2; tail -14 gpurun_out/pmc_r06_lx2_41k/mem.txt; echo lx; tail -14 gpurun_out/pmc_r06_lx_41k/mem.txt
