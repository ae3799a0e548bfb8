"""gnn_pretraining_amd -- MI355X-native message-passing engine behind the module
and operator API of alonbebchuk/GNN-Pretraining.

(The repository layout names this package ``gnn-pretraining_amd``; a hyphen is not
importable in Python, so the directory is ``gnn_pretraining_amd`` with a symlink
under the hyphenated name.)
"""
__all__ = ["graph", "synthetic", "ops"]
