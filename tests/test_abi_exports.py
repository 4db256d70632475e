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


def exported_esmi_symbols(path):
    """Dynamic symbols of the shared library that start with esmi_ (nm -D --defined-only)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("esmi_")})


@pytest.mark.parametrize("which", ["libesmi.so", "libesmi_fp32mfma.so"])
def test_library_exports_exactly_the_declared_symbols(built, which):
    """Both directions: every declared entry point is exported, and nothing named esmi_* is exported that the header does
    not declare (no undeclared development hooks in the product library)."""
    path = os.path.join(os.path.dirname(built), which)
    lib = ctypes.CDLL(path)
    for name in header_functions():
        assert hasattr(lib, name), f"{which} does not export {name}"
    assert exported_esmi_symbols(path) == header_functions()


def test_library_keeps_no_mutable_launch_state(built):
    """The launch plan travels per call (esmi_encoder_block_shape.plan / the plan argument): no setter, no global."""
    assert "esmi_set_fusion" not in exported_esmi_symbols(built)
    from efficientspeech_amd import _lib
    assert _lib.current_plan() == _lib.FUSE_ALL
    with _lib.launch_plan(5):
        assert _lib.current_plan() == 5
        with _lib.launch_plan(0):
            assert _lib.current_plan() == 0
        assert _lib.current_plan() == 5
    assert _lib.current_plan() == _lib.FUSE_ALL


def test_backend_is_hip_gfx950(built):
    from efficientspeech_amd import _lib
    lib = _lib.load()
    assert _lib.backend(lib) == "hip:gfx950"
    assert lib.esmi_version() == 501


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


def test_load_refuses_a_library_that_is_not_the_hip_build(monkeypatch, built):
    """ESMI_LIB / LIB_PATH pointing at the wave-simulator build of the same ABI (a CPU library) is refused by the product's load()
    and use_library(): the only way to bind the simulator is tests/simlib.py's explicit bind()."""
    from efficientspeech_amd import _lib
    from tests.simlib import sim_lib, SIM_SO
    sim_lib()                                         # (built on demand)
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", SIM_SO)
    with pytest.raises(RuntimeError, match="hip:gfx950|HIP build"):
        _lib.load()
    with pytest.raises(RuntimeError, match="not a HIP build"):
        with _lib.use_library(SIM_SO):
            pass


def test_get_mask_from_lengths_matches_reference_semantics():
    """utils/tools.py:43-51: mask[b, t] = t >= lengths[b]; max_len defaults to max(lengths); True marks padding."""
    import torch
    from efficientspeech_amd import get_mask_from_lengths
    lengths = torch.tensor([5, 1, 3, 0])
    m = get_mask_from_lengths(lengths)
    assert m.dtype == torch.bool and m.shape == (4, 5)
    ref = torch.arange(5)[None, :] >= lengths[:, None]
    assert torch.equal(m, ref)
    m7 = get_mask_from_lengths(lengths, max_len=7)
    assert m7.shape == (4, 7) and torch.equal(m7[:, :5], ref) and bool(m7[:, 5:].all())
    assert torch.equal(get_mask_from_lengths(torch.tensor([2]), 2), torch.zeros((1, 2), dtype=torch.bool))


def test_from_lightning_checkpoint_loads_reference_shaped_dict():
    """A Lightning-shaped checkpoint {'state_dict': {'phoneme2mel.*', 'hifigan.*'}, 'hyper_parameters': ...} (what
    demo.py:122 / synthesize.py:103-119 load): the phoneme2mel.* slice loads strict=True, vocoder keys are ignored, a
    missing or unexpected acoustic-model key raises."""
    import numpy as np
    import torch
    from efficientspeech_amd import CONFIGS
    from efficientspeech_amd.model import from_lightning_checkpoint
    from efficientspeech_amd.synth import synth_state_dict
    for name in ("tiny", "base"):
        cfg = CONFIGS[name]
        sd = synth_state_dict(cfg, 31)
        ckpt = {"state_dict": {"phoneme2mel." + k: torch.from_numpy(v) for k, v in sd.items()},
                "hyper_parameters": {"depth": cfg.depth, "reduction": cfg.reduction}}
        ckpt["state_dict"]["hifigan.conv_pre.weight_g"] = torch.zeros(3)
        ckpt["state_dict"]["hifigan.conv_pre.bias"] = torch.zeros(3)
        net = from_lightning_checkpoint(ckpt, cfg)
        got = net.state_dict()
        assert list(got) == list(sd)
        for k, v in sd.items():
            assert np.array_equal(got[k].numpy(), v), k
        broken = {"state_dict": dict(ckpt["state_dict"])}
        del broken["state_dict"]["phoneme2mel.decoder.mel_linear.bias"]
        with pytest.raises(RuntimeError, match="mel_linear.bias"):
            from_lightning_checkpoint(broken, cfg)
        extra = {"state_dict": dict(ckpt["state_dict"])}
        extra["state_dict"]["phoneme2mel.decoder.nope"] = torch.zeros(1)
        with pytest.raises(RuntimeError, match="nope"):
            from_lightning_checkpoint(extra, cfg)


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


def test_bucket_plan_covers_every_request_once():
    from efficientspeech_amd.scheduler import BucketedSynthesizer
    import numpy as np
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 200, size=500).tolist()
    for g, mb in ((1, 256), (8, 64), (32, 7)):
        plan = BucketedSynthesizer(None, max_batch=mb, granularity=g).plan(lens)
        seen = sorted(i for idx, _ in plan for i in idx)
        assert seen == list(range(500))
        for idx, T in plan:
            ls = [lens[i] for i in idx]
            assert len(idx) <= mb and max(ls) == T and T - min(ls) < g
        waste = BucketedSynthesizer.padding_waste(lens, plan)
        assert 0.0 <= waste < g / (np.mean(lens) + g) + 1e-9 if g > 1 else waste == 0.0
    # one big padded batch, for comparison: what the reference does with a ragged batch
    assert BucketedSynthesizer.padding_waste(lens, [(list(range(500)), max(lens))]) > 0.3


def test_wrapper_attaches_the_vocoder_a_checkpoint_carries(tmp_path):
    """model.py:148 builds the vocoder inside the module, so a reference checkpoint holds `hifigan.*` weights and a
    `hifigan_checkpoint` path from the machine that wrote it: loading it here must end with a vocoder attached (built from the
    weights when the path does not exist) -- not silently with `hifigan = None`.  (Host-side only: no kernels run.)"""
    import warnings
    import torch
    from efficientspeech_amd import CONFIGS, EfficientSpeech, build_phoneme2mel
    from efficientspeech_amd.hifigan import HIFIGAN_CONFIGS, Generator, synth_hifigan_state_dict
    from efficientspeech_amd.synth import synth_state_dict
    sd = {"phoneme2mel." + k: torch.from_numpy(v) for k, v in synth_state_dict(CONFIGS["tiny"], 1234).items()}
    vsd = {k: torch.from_numpy(v) for k, v in synth_hifigan_state_dict(HIFIGAN_CONFIGS["v2"], 7).items()}
    sd.update({"hifigan." + k: v for k, v in vsd.items()})
    ckpt = {"state_dict": sd, "hyper_parameters": {"depth": 2, "reduction": 4, "decoder_kernel_size": 5,
                                                   "hifigan_checkpoint": "hifigan/LJ_V2/generator_v2"}}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model = EfficientSpeech.load_from_checkpoint(ckpt)
    assert any("not found" in str(m.message) for m in w)
    assert isinstance(model.hifigan, Generator) and model.hifigan.h.upsample_initial_channel == 128
    for k, v in model.hifigan.state_dict().items():
        assert torch.equal(v, vsd[k]), k
    plugged = Generator(HIFIGAN_CONFIGS["v2"])
    model2 = EfficientSpeech.load_from_checkpoint(ckpt, hifigan=plugged)       # an explicitly plugged-in module wins and is loaded
    assert model2.hifigan is plugged and torch.equal(plugged.state_dict()["conv_pre.weight"], vsd["conv_pre.weight"])
