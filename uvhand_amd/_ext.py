"""Loader of ``_msda_torch.so`` — the thin torch C++ extension over the C ABI (``csrc/torch_ext/msda_torch.cpp``): the
same two native entry points as ``_native`` plus ``MSDeformAttnFunction`` as a C++ autograd node, which removes the
Python-side marshalling from an eager step (bench.py ``eager_ms_per_step``).  Optional: when it has not been built the
package runs the identical kernels through the ctypes binding (``_native``); nothing here computes anything."""
import importlib.util
import os

from . import _native

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_msda_torch.so")
_mod = None
_tried = False


def get(retry=False):
    """The extension module, or None if it is not built / cannot be loaded (ABI mismatch with the installed torch, or
    compiled against another include/msda.h than the loaded libmsda_hip.so).  retry=True looks again (after a rebuild)."""
    global _mod, _tried
    if _tried and not (retry and _mod is None):
        return _mod
    _tried = True
    if os.path.exists(_PATH) and os.path.exists(_native.LIB_PATH):
        try:
            _native.load()                                   # libmsda_hip.so first: same checks, same error if stale
            spec = importlib.util.spec_from_file_location("_msda_torch", _PATH)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            # compile-time MSDA_ABI_VERSION of the extension vs the run-time version of the library it just linked to
            if mod.abi_version() == _native.load().msda_version():
                _mod = mod
        except (ImportError, OSError):
            _mod = None
    return _mod
