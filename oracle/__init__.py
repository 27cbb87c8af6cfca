"""CPU oracle for the LiteRate RJMCMC birth-death likelihood path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product
(``literate_amd``) never imports this package and fails loudly when its HIP
library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference (``/root/reference``) in the build container and records its outputs
on the same inputs; ``tests/test_oracle_golden.py`` checks every function here
against those vectors and against the reference's own known-answer values
(SURVEY.md section 4).  One exception: ``sim_oracle.py`` (the lineage simulator,
a "next" row of SURVEY.md section 8f) is UNPINNED - see its header.
"""
