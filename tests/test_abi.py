"""CPU: the C-ABI shared library loads and exports every symbol include/cellvit_amd.h declares
(no compute calls without a GPU); the product path fails loudly without a device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "cellvit_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cv_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from cellvit_amd import build as B
    from cellvit_amd import _lib
    B.build(verbose=False)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cellvit_amd.h but not exported"
    assert set(names) == set(_lib.SYMBOLS), (set(names) ^ set(_lib.SYMBOLS))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cellvit_amd.model import CellViT256
    m = CellViT256(None, 6, 19)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32))
    import ctypes as C
    from cellvit_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.cv_config()
    cfg.arch, cfg.embed_dim, cfg.depth, cfg.num_heads, cfg.mlp_ratio, cfg.patch_size = 0, 384, 12, 6, 4, 16
    rc = lib.cv_create(C.byref(cfg), C.byref(h))
    assert rc == _lib.CV_ERR_HIP and b"no CPU fallback" in lib.cv_last_error()
    pp = C.c_void_p()
    assert lib.cv_pp_create(1, 64, 64, 16, 16, C.byref(pp)) == _lib.CV_ERR_HIP


def test_state_dict_contract_of_the_shim():
    from cellvit_amd.model import CellViT256
    from cellvit_amd.spec import param_specs
    from cellvit_amd.weights import make_state_dict
    m = CellViT256(None, 6, 19)
    assert [k for k in m.state_dict()] == [k for k, _, _ in param_specs(m.cfg)]
    assert str(m.load_state_dict(make_state_dict(m.cfg))) == "<All keys matched successfully>"
    assert m.patch_size == 16 and m.embed_dim == 384 and m.num_nuclei_classes == 6
