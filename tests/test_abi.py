"""CPU-side checks of the boundary: the library loads and exports every symbol include/pfotgn.h declares."""
import os
import re

import numpy as np
import pytest

from conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "pfotgn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pfo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pfotgnrec_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), name
        assert name in _lib.PROTOTYPES, "no ctypes prototype for %s" % name
    assert lib.pfo_abi_version() == 6


def test_param_layout_matches_reference_inventory():
    """1.551 M trainable fp32 at C2 (D=172, Ef=4, L=2), 34 304 at C1 incl. the dead layer_norm."""
    import ctypes
    from pfotgnrec_amd import _lib
    cfg = _lib.TgnConfig(50501, 1000001, 172, 4, 2, 2, 1, 2560, 20, 512)
    lay = _lib.TgnLayout()
    _lib.call("pfo_tgn_param_layout", ctypes.byref(cfg), ctypes.byref(lay))
    per_layer = 344 * 344 + 2 * 344 * 348 + 1032 + 344 * 344 + 344 + 172 * 516 + 172 + 172 * 172 + 172
    assert per_layer == 596152
    assert lay.total == 344 + 358104 + 2 * per_layer
    assert lay.time_b == lay.time_w + 172
    assert _lib.load().pfo_tgn_workspace_bytes(ctypes.byref(cfg)) > 0


def test_invalid_config_is_rejected_with_message():
    import ctypes
    from pfotgnrec_amd import _lib
    cfg = _lib.TgnConfig(100, 10, 30, 4, 1, 2, 1, 8, 4, 4)       # D not a multiple of 4
    lay = _lib.TgnLayout()
    with pytest.raises(_lib.PfoError, match="multiple of 4"):
        _lib.call("pfo_tgn_param_layout", ctypes.byref(cfg), ctypes.byref(lay))


def test_compute_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pfotgnrec_amd as P
    from pfotgnrec_amd import _lib
    from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
    g = make_graph(SyntheticConfig("t", 50, 10, 400, 8, 1, 4, 2), with_prices=False)
    nf = P.get_neighbor_finder(g.data, False)
    with pytest.raises(_lib.PfoError):
        nf.get_temporal_neighbor(g.data.sources[:3], g.data.timestamps[:3], 4)
    tgn = P.TGN(nf, g.node_features, g.edge_features, "cpu", n_layers=1, n_heads=2, use_memory=True, memory_dimension=8,
                message_function="identity")
    with pytest.raises(_lib.PfoError):
        tgn.compute_temporal_embeddings(g.data.sources[:2], g.data.destinations[:2], g.data.destinations[:6],
                                        g.data.timestamps[:2], g.data.edge_idxs[:2], 4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "pfotgnrec_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
