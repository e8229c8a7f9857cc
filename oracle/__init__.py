"""CPU oracle for the DynaVSR EDVR + inner-MAML hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dynavsr_amd/`` imports this package; the only
allowed callers are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  See ``DESIGN.md`` ("Oracle") for what each file restates and how it is pinned.
"""
