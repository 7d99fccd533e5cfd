"""CPU oracle for the nested-dissection elimination path (TEST INFRASTRUCTURE ONLY).

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  See ``oracle/hs_oracle.py`` for the
parity-pinning statement.
"""
