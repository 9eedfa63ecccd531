export AHIP_NO_ARITH_SELFCHECK=1
bash pair_allegro_amd/tools/ab4.sh pair_allegro_amd/abl_base.so pair_allegro_amd/abl_er1.so pair_allegro_amd/abl_er2.so pair_allegro_amd/abl_base.so pair_allegro_amd/abl_er1.so pair_allegro_amd/abl_er2.so
ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/abl_er2.so timeout 300 python -m pytest tests/test_gpu_fused.py -q -m gpu -x -k "golden or parity or soak or multi_rank" 2>&1 | tail -2
