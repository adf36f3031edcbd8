#!/usr/bin/env python3
"""Builds uvhand_amd/_msda_torch.so in-tree: one g++ command against the installed torch headers (no kernels in it —
it links libmsda_hip.so, found next to it through an $ORIGIN runpath).  Called by __graft_entry__.build(); skipped when
the output is newer than its inputs."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(os.path.dirname(HERE))
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "_msda_torch.so")
SRC = os.path.join(HERE, "msda_torch.cpp")


STAMP = os.path.join(ROOT, "build", "torch_ext", "built_with.txt")


def _stamp():
    """What the extension was built against besides its sources: the torch build (its headers and ABI) and the C-ABI
    library next to it.  A torch upgrade or a rebuilt libmsda_hip.so makes the stamp differ -> rebuild."""
    import torch
    lib = os.path.join(PKG, "libmsda_hip.so")
    return "torch %s\nlibmsda_hip.so %d\n" % (torch.__version__, int(os.path.getmtime(lib)) if os.path.exists(lib) else 0)


def build(verbose=False, force=False):
    deps = [SRC, os.path.join(ROOT, "include", "msda.h"), os.path.abspath(__file__)]
    try:
        with open(STAMP) as f:
            same_env = f.read() == _stamp()
    except OSError:
        same_env = False
    if (not force and same_env and os.path.exists(OUT)
            and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)):
        return OUT
    import torch
    from torch.utils.cpp_extension import include_paths, library_paths
    try:
        incs, libs = include_paths("cuda"), library_paths("cuda")
    except TypeError:                                  # older signature: include_paths(cuda=True)
        incs, libs = include_paths(True), library_paths(True)
    cmd = ["g++", "-O2", "-fPIC", "-std=c++17", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
           "-DTORCH_EXTENSION_NAME=_msda_torch", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for p in incs + ["/opt/rocm/include", sysconfig.get_paths()["include"]]:
        cmd += ["-isystem", p]
    cmd += ["-I" + os.path.join(ROOT, "include"), SRC, "-o", OUT]
    for p in libs + ["/opt/rocm/lib", PKG]:
        cmd += ["-L" + p]
    cmd += ["-lmsda_hip", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-lamdhip64",
            "-Wl,-rpath,$ORIGIN", "-Wl,--enable-new-dtags"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.makedirs(os.path.dirname(STAMP), exist_ok=True)
    with open(STAMP, "w") as f:
        f.write(_stamp())
    return OUT


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv, force="--force" in sys.argv))
