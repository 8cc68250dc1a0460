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
"""
