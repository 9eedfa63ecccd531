"""The product library loads without a GPU and exports every symbol include/allegro_hip.h declares
(no compute calls here); loading a model without a GPU fails loudly instead of falling back."""
import os
import re

import pytest

from pair_allegro_amd import capi, model_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "allegro_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ahip_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    import __graft_entry__ as g
    if not os.path.exists(capi.DEFAULT_LIB):
        g.build()
    lib = capi.Library()
    names = _declared()
    assert set(names) == set(capi.SYMBOLS), (sorted(set(names) ^ set(capi.SYMBOLS)))
    for n in names:
        assert hasattr(lib.lib, n), n


def test_no_cpu_fallback(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = capi.Library()
    assert lib.device_count() == 0
    cfg = model_file.model_S()
    p = str(tmp_path / "m.ahip")
    model_file.save_ahip(p, cfg, model_file.init_weights(cfg))
    with pytest.raises(capi.AhipError) as e:
        capi.Model(p, 0, lib)
    assert e.value.code == capi.AHIP_ERR_DEVICE and "no CPU fallback" in e.value.msg
