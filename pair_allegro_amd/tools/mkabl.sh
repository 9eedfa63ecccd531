#!/bin/bash
# Builds A/B variants of liballegro_hip.so that differ in ONE object: pair_allegro_amd/abl_<name>.so (git-ignored; they travel to the GPU box).
# usage (repo root, after `make -C pair_allegro_amd/csrc`):
#   pair_allegro_amd/tools/mkabl.sh fused   "name:extra hipcc flags" ...     -> fused.o rebuilt with the Makefile's options + the extra flags
#   pair_allegro_amd/tools/mkabl.sh fused_lx2 "name:extra hipcc flags" ...
# then on the GPU box: bash pair_allegro_amd/tools/ab4.sh pair_allegro_amd/liballegro_hip.so pair_allegro_amd/abl_<name>.so   (ab5.sh for config 5)
obj=$1; shift
cd "$(dirname "$0")/../csrc" || exit 1
COMMON="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function"
case $obj in
  fused_h)   SPEC="-DAHIP_FUSED_PART=2 -mllvm -amdgpu-use-amdgpu-trackers=1 -mllvm -disable-machine-licm -mllvm -disable-postra-machine-licm"; SRC=fused.hip ;;
  fused)     SPEC="-DAHIP_FUSED_PART=0 -mllvm -amdgpu-use-amdgpu-trackers=1 -mllvm -disable-machine-licm -mllvm -disable-postra-machine-licm" ;;
  fused_lx2) SPEC="-mllvm -pragma-unroll-threshold=1000000 -mllvm -disable-machine-licm -mllvm -disable-postra-machine-licm -mllvm -amdgpu-use-amdgpu-trackers=1" ;;
  fused_lx)  SPEC="-mllvm -pragma-unroll-threshold=1000000 -mllvm -disable-machine-licm -mllvm -disable-postra-machine-licm" ;;
  *) echo "unknown object $obj"; exit 1 ;;
esac
[ -n "$ABL_SPEC" ] && SPEC="$ABL_SPEC"      # ABL_SPEC=... replaces the per-object options of the Makefile (e.g. to drop an -mllvm flag)
ALL="allegro_hip.o prims.o neigh.o edges.o gemm.o fused.o fused_bf.o fused_h.o fused_lx.o fused_lx2.o comm.o model_io.o"
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  (
    /opt/rocm/bin/hipcc $COMMON $SPEC $flags -c ${SRC:-$obj.hip} -o /tmp/abl_${name}_$obj.o 2> /tmp/abl_${name}.err || { echo "build $name failed"; head -5 /tmp/abl_${name}.err; exit 1; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../abl_$name.so ${ALL/$obj.o//tmp/abl_${name}_$obj.o} -ldl && echo "built $name"
  ) &
done
wait
