AHIP_NO_ARITH_SELFCHECK=1 AHIP_FUSED_PROF=1 AHIP_FUSED_CLK=1 timeout 300 python bench.py --config 4 --ncell 30 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "fused prof|fused clk" | tail -3
