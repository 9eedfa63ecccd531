mkdir -p gpurun_out/r06v
R="python -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
timeout 900 $R --nproc-per-node 8 --master-port 29771 pair_allegro_amd/tools/rehearse_ranks.py --config 4 --out gpurun_out/r06v/rehearse_config4_8ranks.json > gpurun_out/r06v/r4.log 2>&1
timeout 600 $R --nproc-per-node 4 --master-port 29773 pair_allegro_amd/tools/rehearse_ranks.py --config 3 --out gpurun_out/r06v/rehearse_config3_4ranks.json > gpurun_out/r06v/r3.log 2>&1
timeout 900 $R --nproc-per-node 8 --master-port 29775 pair_allegro_amd/tools/rehearse_ranks.py --config 5 --out gpurun_out/r06v/rehearse_config5_8ranks.json > gpurun_out/r06v/r5.log 2>&1
for f in gpurun_out/r06v/rehearse_*.json; do python -c "
import json,sys; d=json.load(open('$f')); o=d['overlapped']; print('$f', d['ok'], o['max_abs_dF_after_steps'], o['re_neighboring_in_the_library'], o['rebuild_ms_max_over_ranks_shared_gpu'], d['serial']['rebuild_ms_max_over_ranks_shared_gpu'])"; done
tail -3 gpurun_out/r06v/r4.log | cut -c1-300
