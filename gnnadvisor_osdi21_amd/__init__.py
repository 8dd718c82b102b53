"""gnnadvisor_osdi21_amd -- MI355X-native neighbor-group aggregation runtime.

A from-scratch gfx950 implementation of GNNAdvisor's one hot path (the
neighbor-group-partitioned SpMM aggregation behind the ``GNNAdvisor`` extension API),
not a port of its CUDA code.  Layout:

* ``csrc/``          HIP kernels + C ABI (``libgnna.so``, declared in ``include/gnna.h``)
                     and the pybind/torch module ``GNNAdvisor`` (``GNNAdvisor.so``).
* ``_lib``           ctypes binding of the C ABI.
* ``decider``        ``inputProperty`` / Decider   (reference: GNNAdvisor/param.py)
* ``ops``            autograd ops + GCNConv/GINConv (reference: GNNAdvisor/gnn_conv.py)
* ``verify``         ``Verification`` harness       (reference: GNNAdvisor/unitest.py)
* ``graph``          synthetic graphs, CSR + degree builder (reference: GNNAdvisor/dataset.py:99-122)
* ``dist``           dst-range sharding + RCCL all-gather halo exchange (new)

The extension is required: importing ``GNNAdvisor`` raises ImportError when it has not
been built -- there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import importlib
import os
import sys

__version__ = "0.6.0"          # = the library's (include/gnna.h GNNA_VERSION 600, gnna_build_id)
_PKG = os.path.dirname(os.path.abspath(__file__))


def load_extension():
    """Import and return the ``GNNAdvisor`` extension module (built in-tree)."""
    path = os.path.join(_PKG, "GNNAdvisor.so")
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the HIP extension first "
            "(python -m gnnadvisor_osdi21_amd.build). There is no CPU fallback.")
    import torch  # noqa: F401  (libtorch / libamdhip64 must be loaded first)
    return importlib.import_module(__name__ + ".GNNAdvisor")


def install_reference_aliases() -> None:
    """Register the reference's top-level module names so that its scripts' imports
    (``import GNNAdvisor as GNNA``, ``from param import *``, ``from gnn_conv import *``,
    ``from dataset import *``, ``from unitest import *`` -- GNNA_main.py:10-13,117,131) resolve to this
    package (``dataset`` -> loader.custom_dataset, which needs neither dgl nor rabbit)."""
    sys.modules.setdefault("GNNAdvisor", load_extension())
    for ref_name, ours in (("param", "decider"), ("gnn_conv", "ops"), ("dataset", "loader"), ("unitest", "verify")):
        sys.modules.setdefault(ref_name, importlib.import_module(f"{__name__}.{ours}"))
