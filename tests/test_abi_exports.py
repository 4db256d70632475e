"""The C-ABI shared library builds for gfx950 and exports every symbol include/esmi.h declares.
(No compute calls here: this runs without a GPU.)"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return g.LIB


def header_functions():
    src = open(os.path.join(ROOT, "include", "esmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(esmi_[A-Za-z0-9_]+)\s*\(", src)))


def test_header_lists_expected_entry_points():
    from efficientspeech_amd import _lib
    assert sorted(_lib.EXPORTS) == header_functions()


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built)
    for name in header_functions():
        assert hasattr(lib, name), f"libesmi.so does not export {name}"


def test_backend_is_hip_gfx950(built):
    from efficientspeech_amd import _lib
    lib = _lib.load()
    assert _lib.backend(lib) == "hip:gfx950"
    assert lib.esmi_version() == 100


def test_code_object_targets_gfx950(built):
    blob = open(built, "rb").read()
    assert b"gfx950" in blob and b"mel_decoder_kernel" in blob


def test_product_path_refuses_cpu_tensors(built):
    """No CPU fallback: host tensors are rejected loudly by the product library."""
    import torch
    from efficientspeech_amd import CONFIGS, build_phoneme2mel
    net = build_phoneme2mel(CONFIGS["tiny"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net({"phoneme": torch.ones((1, 8), dtype=torch.int32)})


def test_missing_library_raises(monkeypatch, built):
    from efficientspeech_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libesmi.so")
    with pytest.raises(RuntimeError, match="not built"):
        _lib.load()


def test_state_dict_keys_match_reference_table():
    """Checkpoint compatibility: the module mirror produces exactly the reference's keys/shapes
    (the table in synth.state_dict_spec was asserted against the reference by tools/gen_golden.py)."""
    from efficientspeech_amd import CONFIGS, build_phoneme2mel
    from efficientspeech_amd.config import PARAM_COUNTS
    from efficientspeech_amd.synth import state_dict_spec
    for name, cfg in CONFIGS.items():
        net = build_phoneme2mel(cfg)
        got = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        assert got == [(k, s) for k, s, _ in state_dict_spec(cfg)]
        assert sum(p.numel() for p in net.parameters()) == PARAM_COUNTS[name]
