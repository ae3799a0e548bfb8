"""CPU oracle for the GIN message-passing hot path.  TEST INFRASTRUCTURE ONLY.

This package is a torch-only (CPU, fp32) restatement of the reference's
pre-training hot path (alonbebchuk/GNN-Pretraining) and of
the PyTorch-Geometric operators that path calls.  It exists so that the HIP
kernels can be checked against an independent implementation.

Rules (enforced by tests/test_layout.py):
  * only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may
    import anything from here;
  * nothing under gnn_pretraining_amd/ imports it, and the product path raises
    when libgnnmp.so is missing rather than falling back to this code.

Pinning status:
  * schedulers / loss balancer / PCGrad / optimizer groups: PINNED against
    golden vectors produced by importing the reference's own modules
    (tests/golden/make_reference_goldens.py, run in the build container);
  * FinetuneGNN architecture: PINNED against the 12 ``trainable_parameters``
    values in the reference's analysis/results/experiment_results.csv;
  * GINConv / pooling / subgraph / to_undirected / Batch collation live in the
    third-party dependency torch-geometric (>=2.3.0, unpinned,
    requirements.txt:3), which is absent from /root/reference and not
    installable here; the reference ships no tests or fixtures for them.
    Their restatement follows PyG's published semantics and is
    **parity unpinned** at that boundary;
  * graph_properties.py (the 12 graph-property targets) calls networkx, the
    library the reference itself calls, function for function; miner.py (the
    fine-tune hard-negative miner) is a line-by-line torch-CPU restatement.
    The reference holds no fixture for either: **parity unpinned** beyond the
    closed-form cases in tests/test_data.py / tests/test_gpu_miner.py.
"""
