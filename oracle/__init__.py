"""CPU restatement of the reference's hot path - TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is the parity checker for the HIP path: plain
numpy (+ a few lines of C for the exact fp32 FMA), each function citing the
``/root/reference`` file:line it follows.  It may be imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  The product package ``pfotgnrec_amd`` never imports it and has
no CPU fallback: without the HIP library the product raises.

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md
§4), so the oracle is pinned against outputs of the reference itself, imported
in the build container by ``tools/make_golden.py`` and committed as
``tests/golden/*.npz`` (see ``tests/test_oracle_golden.py``).
"""
