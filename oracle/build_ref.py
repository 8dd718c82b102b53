"""Build the reference's own host code into oracle/_ref/ (test infrastructure).

Compiles /root/reference/GNNAdvisor/GNNConv/GNNAdvisor.cpp -- the file that holds the
reference's CPU partitioner ``build_part`` (GNNAdvisor.cpp:210-251) -- straight from
where it lies, with g++ against the installed torch headers, into
``oracle/_ref/GNNAdvisor_ref.so``.  No reference source is copied and no stand-in for
the CUDA side is written: the five ``*_cuda`` launchers the file declares
(GNNAdvisor.cpp:4-69) stay undefined in the shared object and the module is loaded
with lazy binding (``RTLD_LAZY``), so only ``build_part`` is ever callable.  The
reference's GPU half (GNNAdvisor_kernel.cu) is unbuildable here (CUDA-only, and it no
longer compiles against torch 2.10 -- SURVEY.md 8c).

oracle/_ref/ is git-ignored (kept out of history) but not gpurun-ignored.
"""
import os
import subprocess
import sys
import sysconfig

REF_CPP = "/root/reference/GNNAdvisor/GNNConv/GNNAdvisor.cpp"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "GNNAdvisor_ref.so")


def available() -> bool:
    return os.path.exists(REF_CPP)


def build(force: bool = False) -> str | None:
    if not available():
        return OUT if os.path.exists(OUT) else None
    if os.path.exists(OUT) and not force and os.path.getmtime(OUT) >= os.path.getmtime(REF_CPP):
        return OUT
    import pybind11
    import torch
    tdir = os.path.dirname(torch.__file__)
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w",
           "-DTORCH_EXTENSION_NAME=GNNAdvisor_ref", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I" + os.path.join(tdir, "include"),
           "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
           "-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include(),
           REF_CPP, "-o", OUT,
           "-L" + os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
           "-Wl,-rpath," + os.path.join(tdir, "lib")]
    subprocess.check_call(cmd)
    return OUT


def load():
    """Import the reference module (lazy binding: the *_cuda symbols stay unresolved)."""
    path = build()
    if path is None:
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY)
    try:
        spec = importlib.util.spec_from_file_location("GNNAdvisor_ref", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
