#!/bin/bash
# Device assembly + static census of the headline instance of k_fused (f16x2, 4 waves, 2 layers, MLP depth 2) alone: ~25 s instead of the whole object.
# usage (repo root): [AHIP_NW=8] pair_allegro_amd/tools/asm_k_fused.sh [extra hipcc flags]   -> /tmp/k_fused_headline.s
cd "$(dirname "$0")/../csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function -DAHIP_FUSED_PART=2 -DAHIP_ASM_ONLY -DAHIP_ASM_NW=${AHIP_NW:-4} \
  -mllvm -amdgpu-use-amdgpu-trackers=1 -mllvm -disable-machine-licm -mllvm -disable-postra-machine-licm "$@" \
  --offload-device-only -S fused.hip -o /tmp/k_fused_headline.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Function Name: _ZN4ahip7k_fused" | grep -E "VGPRs:|ScratchSize|LDS Size" | sed 's/.*remark: *//'
python3 ../tools/isa_census.py /tmp/k_fused_headline.s k_fusedILi${AHIP_NW:-4}ELb0ELi3ELb1ELi2ELi2
