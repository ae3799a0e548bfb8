"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares;
the product package never imports the oracle; no compute is called here."""
import ast
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gnn_pretraining_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    l = _lib.lib()
    syms = _lib.declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(l, s), s
    assert set(_lib._SIGS) == set(syms), set(_lib._SIGS) ^ set(syms)
    assert l.gmp_version() >= 100


def test_product_never_imports_oracle():
    bad = []
    for base in ("gnn_pretraining_amd",):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if not f.endswith(".py"):
                    continue
                tree = ast.parse(open(os.path.join(dp, f)).read())
                for node in ast.walk(tree):
                    names = []
                    if isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        names = [node.module or ""]
                    if any(n == "oracle" or n.startswith("oracle.") for n in names):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_missing_gpu_tensor_is_a_loud_error():
    import torch
    from gnn_pretraining_amd import ops
    from gnn_pretraining_amd._lib import GnnmpError
    with pytest.raises(GnnmpError):
        ops.csr_build(torch.zeros(2, 3, dtype=torch.long), 4)       # CPU tensor: no fallback
