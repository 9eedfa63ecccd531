"""The gfx950 store-data hazard behind the wrong saved rows of rounds 2-3 (csrc/fused_common.h: bstore), as a test: the 40-line
reproducer pair_allegro_amd/tools/store_hazard.hip is built on the GPU box and must report NO poisoned dword for the padded forms -- the two wait states
`bstore` puts behind every buffer_store_dwordx4 -- whatever it reports for the unpadded ones (printed: that is the hardware's business)."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_wait_states_behind_a_wide_buffer_store_are_enough(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "store_hazard")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-o", exe, os.path.join(ROOT, "pair_allegro_amd", "tools", "store_hazard.hip")], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = subprocess.run([exe], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300).stdout.decode()
    print(out)
    rows = re.findall(r"^(.*?)\s+poisoned dwords by data register: (\d+) (\d+) (\d+) (\d+)\s+of (\d+) each", out, flags=re.M)
    assert len(rows) >= 6, out
    padded = [r for r in rows if "2 wait states" in r[0]]
    assert len(padded) == 2                                   # soffset in an SGPR (every saved-row store of the kernels) and soffset = 0
    for r in padded:
        assert int(r[5]) > 1_000_000 and all(int(v) == 0 for v in r[1:5]), r
