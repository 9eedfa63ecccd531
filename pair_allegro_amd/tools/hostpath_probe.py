"""Times the HOST-pointer entry point (`ahip_compute`: x H2D + types H2D + force evaluation + f D2H + host add),
i.e. the PCIe-inclusive rate of the plain (non-Kokkos) LAMMPS coupling.  python -m pair_allegro_amd.tools.hostpath_probe [ncell]"""
import sys
import tempfile
import time

import numpy as np

from pair_allegro_amd import capi, lmp_like, model_file


def main():
    ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    cfg = model_file.model_S()
    path = tempfile.mkdtemp() + "/hp.ahip"
    model_file.save_ahip(path, cfg, model_file.init_weights(cfg))
    m = capi.Model(path, 0, capi.Library())
    cell, pos, types = lmp_like.diamond_si(ncell)
    rs = lmp_like.build_rank_system(cell, pos, types, cfg["r_max"] + 1.0)
    t0 = time.perf_counter()
    m.neigh_update_csr(rs.nall, rs.ilist, rs.offsets, rs.flat)
    print(f"neigh_update_csr ({rs.offsets[-1]} entries): {(time.perf_counter() - t0) * 1e3:.1f} ms (rebuild steps only)")
    f = np.zeros_like(rs.x)
    mapper = np.array([0], np.int32)
    cm = np.array([[cfg["r_max"]]])
    m.set_option("timing", "1")
    for _ in range(5):
        f[:] = 0
        m.timings()
        t0 = time.perf_counter()
        m.compute(rs.nlocal, rs.nghost, rs.x, rs.type, mapper, cm, f)
        dt = time.perf_counter() - t0
        st = m.timings()
        dev = sum(v for k, v in st.items() if k in ("edge_build", "tile_pack", "model_fused", "model_generic"))
        print(f"ahip_compute host pointers, {rs.nlocal} atoms (+{rs.nghost} ghosts): {dt * 1e3:.2f} ms -> "
              f"{rs.nlocal / dt / 1e6:.2f} M atom-evaluations/s PCIe-inclusive ({m.last_path}); device stages {dev:.2f} ms, "
              f"host + PCIe overhead {dt * 1e3 - dev:.2f} ms = {100 * (dt * 1e3 - dev) / (dt * 1e3):.1f} % of the call")


if __name__ == "__main__":
    main()
