"""Test helper: build + bind the CPU wave-simulator build of the kernel sources (tools/wavesim).

Only tests use this.  `use_sim()` temporarily swaps the library handle inside
efficientspeech_amd._lib so the very same host-side modules drive the simulated kernels on
host tensors; the product never does this (it loads libesmi.so or raises).
"""
import contextlib
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_SO = os.path.join(ROOT, "tools", "wavesim", "_build", "libesmi_sim.so")
_CSRC = os.path.join(ROOT, "efficientspeech_amd", "csrc")
_SRCS = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))] + \
        [os.path.join(ROOT, "tools", "wavesim", f) for f in ("wavesim.h", "wavesim.cpp")] + \
        [os.path.join(ROOT, "include", "esmi.h")]
_handle = None


def sim_lib():
    global _handle
    if _handle is None:
        stale = not os.path.exists(SIM_SO) or any(os.path.getmtime(s) > os.path.getmtime(SIM_SO) for s in _SRCS)
        if stale:
            subprocess.check_call([os.path.join(ROOT, "tools", "wavesim", "build.sh")], stdout=subprocess.DEVNULL)
        from efficientspeech_amd import _lib
        _handle = _lib.bind(ctypes.CDLL(SIM_SO))
        assert _handle.esmi_backend() == b"wavesim"
    return _handle


def _sim_runtime(t):
    """Stand-in for efficientspeech_amd.networks._runtime while the simulator is bound: host tensors only, no stream."""
    if t.is_cuda:
        raise RuntimeError("wave-simulator backend only accepts host tensors")
    return sim_lib(), None


@contextlib.contextmanager
def use_sim():
    from efficientspeech_amd import _lib, networks
    old_lib, old_rt = _lib._LIB, networks._runtime
    _lib._LIB, networks._runtime = sim_lib(), _sim_runtime
    try:
        yield _lib._LIB
    finally:
        _lib._LIB, networks._runtime = old_lib, old_rt
