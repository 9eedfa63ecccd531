#!/bin/bash
# Install the HIP `pair_style allegro` into a LAMMPS source tree (counterpart of the reference's
# patch_lammps.sh, which adds find_package(Torch); here the only dependency is liballegro_hip.so).
#   ./patch_lammps.sh /path/to/lammps [/path/to/allegro-hip-repo]
set -euo pipefail
lammps_dir=${1:?usage: patch_lammps.sh <lammps dir> [repo dir]}
repo_dir=${2:-$(cd "$(dirname "$0")/../.." && pwd)}
[ -f "$lammps_dir/cmake/CMakeLists.txt" ] || { echo "$lammps_dir does not look like a LAMMPS tree"; exit 1; }
cp "$repo_dir"/pair_allegro_amd/lammps/pair_allegro_hip.{h,cpp} "$lammps_dir/src/"
cp "$repo_dir"/pair_allegro_amd/lammps/compute_allegro_hip.{h,cpp} "$lammps_dir/src/"      # compute allegro, compute allegro/atom
cp "$repo_dir"/include/allegro_hip.h "$lammps_dir/src/"
if [ -d "$lammps_dir/src/KOKKOS" ]; then      # pair_style allegro/kk (device-resident coupling, HIP backend of the KOKKOS package)
  cp "$repo_dir"/pair_allegro_amd/lammps/pair_allegro_hip_kokkos.{h,cpp} "$lammps_dir/src/KOKKOS/"
fi
cat >> "$lammps_dir/cmake/CMakeLists.txt" <<CMAKE

# --- allegro-hip: MI355X-native pair_style allegro -------------------------------------------
find_library(ALLEGRO_HIP_LIBRARY allegro_hip HINTS "$repo_dir/pair_allegro_amd" REQUIRED)
target_link_libraries(lammps PUBLIC \${ALLEGRO_HIP_LIBRARY})
CMAKE
echo "Done.  Build liballegro_hip.so first (make -C $repo_dir/pair_allegro_amd/csrc), then configure LAMMPS as usual."
