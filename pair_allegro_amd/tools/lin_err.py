import os, sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from pair_allegro_amd import capi
for arith in ['f32', 'b3']:
    os.environ['AHIP_FUSED_ARITH'] = arith
    lib = capi.Lib()
    for K, N in [(8, 64), (32, 32), (32, 64), (64, 32), (64, 64), (96, 64), (64, 96), (64, 8)]:
        rng = np.random.RandomState(K * 100 + N)
        W = rng.normal(size=(K, N)).astype(np.float32).astype(np.float64)
        x = rng.normal(size=(32, K)).astype(np.float32)
        out = lib.debug_fused_linear(W, x)
        ref = x.astype(np.float64) @ W
        ref32 = (x @ W.astype(np.float32))
        print(arith, K, N, 'max err vs f64: %.3e  (numpy f32 matmul: %.3e)  scale %.2f' % (np.abs(out - ref).max(), np.abs(ref32 - ref).max(), np.abs(ref).max()))
