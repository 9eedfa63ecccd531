#!/bin/bash
# Assembly-level fault bisection (round 4: how the store-data hazard of fused_common.h: bstore was found).
#   1. tools/asm_variant.sh asm <tag> [extra hipcc options]      -> /tmp/ahip_asm/<tag>.s   device assembly of the bf16-split k_fused instances
#   2. python tools/asm_xform.py in.s <mode> out.s               -> an edited copy (s_nop / s_waitcnt inserted per instruction class or kernel region)
#   3. tools/asm_variant.sh lib <file.s> <tag>                   -> pair_allegro_amd/var/liballegro_hip_<tag>.so (assembled, bundled, linked with the normal objects)
#   4. gpurun -- python pair_allegro_amd/tools/dbg_arith_variants.py     runs the failing case on every library in var/
set -e
HERE=$(cd "$(dirname "$0")" && pwd); CS=$HERE/../csrc; LL=/opt/rocm/lib/llvm/bin; W=/tmp/ahip_asm; mkdir -p $W
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize -DAHIP_FUSED_PART=1"
case "$1" in
asm) TAG=$2; shift 2
  (cd $CS && /opt/rocm/bin/hipcc $FLAGS "$@" --cuda-device-only -S fused.hip -o $W/$TAG.s) ;;
lib) S=$2; TAG=$3; D=$W/$TAG.d; mkdir -p $D $HERE/../var
  $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $S -o $D/dev.o
  $LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $D/dev.out $D/dev.o
  $LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$D/dev.out -output=$D/dev.hipfb
  (cd $CS && /opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $D/dev.hipfb -c fused.hip -o $D/fused_bf.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../var/liballegro_hip_$TAG.so allegro_hip.o prims.o neigh.o edges.o gemm.o fused.o $D/fused_bf.o fused_lx.o fused_lx2.o comm.o model_io.o -ldl)
  ls -la $HERE/../var/liballegro_hip_$TAG.so ;;
*) sed -n 2,7p "$0" ;;
esac
