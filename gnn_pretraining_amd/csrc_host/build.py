"""Builds the host-side native module (csrc_host/hostdraw.cpp -> gnn_pretraining_amd/_hostdraw.so) with g++ against the
installed torch headers.  `python -m gnn_pretraining_amd.csrc_host.build`; __graft_entry__.build() calls it."""
import os
import subprocess
import sys
import sysconfig


def build(verbose: bool = True) -> str:
    import pybind11
    import torch
    from torch.utils import cpp_extension as E
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "_hostdraw.so")
    src = os.path.join(here, "hostdraw.cpp")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-DTORCH_EXTENSION_NAME=_hostdraw", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", src, "-o", out,
           "-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include()]
    cmd += ["-I" + p for p in E.include_paths()]
    for lp in E.library_paths():
        cmd += ["-L" + lp, "-Wl,-rpath," + lp]
    cmd += ["-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print("built", build())
