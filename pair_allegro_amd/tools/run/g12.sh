export AHIP_NO_ARITH_SELFCHECK=1
bash pair_allegro_amd/tools/pmc_mem.sh r06_lx2_41k "--config 5 --ncell 24" > /dev/null 2>&1
bash pair_allegro_amd/tools/pmc_mem.sh r06_lx_41k "--config 6 --ncell 24" > /dev/null 2>&1
bash pair_allegro_amd/tools/traffic_passes.sh r06_6 "--config 6" > gpurun_out/traffic_r06_6.log 2>&1
echo lx2; cat gpurun_out/pmc_r06_lx2_41k/mem.txt | tail -14; echo lx; cat gpurun_out/pmc_r06_lx_41k/mem.txt | tail -14; grep "^k_" gpurun_out/traffic_r06_6.log
