"""CPU oracle for the UDA_CLR per-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``uda_clr_amd/`` imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may.  It restates, in plain PyTorch CPU ops on a flat ``state_dict``, the
algorithm of the reference's hot path (each function cites the reference
file:line it follows).

Pinning: every function here is checked against outputs of the reference itself
(imported from ``/root/reference`` in the build container by
``tests/golden/make_golden.py``); the resulting fixtures live in
``tests/golden/`` and are re-checked by ``tests/test_oracle_golden.py``.
The two losses of SURVEY.md §8 row a15 have no shipped source: their restatement
in ``losses_ref.py`` is marked "parity unpinned".
"""
