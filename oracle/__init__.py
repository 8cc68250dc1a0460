"""CPU oracle for the FedICRA hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a PyTorch-CPU / numpy restatement of the
reference's algorithm for the hot path SURVEY.md section 8 names.  It is the
checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  ``fedicra_amd`` never does.

Parity status: PINNED against the reference itself.  ``oracle/gen_golden.py``
imports the reference's own modules from /root/reference (this container only,
``.cuda()`` shimmed to a no-op) and writes input/output vectors to
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors.  Two third-party pieces are NOT in /root/reference and
are restated from their published definition ("parity unpinned" for them):
``flwr==1.0.0`` ``aggregate`` (weighted mean) and ``medpy==0.4.0``
``metric.binary.dc`` (Dice) -- see oracle/fed_ref.py and oracle/metrics_ref.py.
A third piece IS in /root/reference but cannot be run here: the tree filter's
device kernels (``code/kernels/lib_tree_filter/src/bfs/bfs.cu`` and
``.../refine/refine.cu``: CUDA + THC).  ``oracle/tree_ref.py``'s BFS ordering and
``root_leaf_prop`` / ``leaf_root_aggr`` restatement follows them BY READING ONLY
("parity unpinned" for those two functions); what pins the tree-energy stack
end to end is the reference's own ``boruvka.cpp`` built by plain ``g++``
(``oracle/_ref/libboruvka_ref.so``) for the spanning tree, plus the property
tests in ``tests/test_losses_gpu.py`` (the filter of a chain / star tree in
closed form, symmetry of the edge weights, gradient checks in fp64).
"""
