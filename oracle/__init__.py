"""CPU oracle for the HorizonNet hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker.  The product path
(``horizonnet_amd``) never imports this package and fails loudly when its HIP
library is missing.

Parity pinning: the reference has no tests and no golden vectors
(SURVEY.md section 4).  The restatements here are pinned against the
reference's *own code* run in the build container (``oracle/gen_golden.py``
imports ``/root/reference`` with the stand-ins under ``oracle/standins`` for
the absent third-party ``torchvision``) and the resulting vectors are
committed under ``tests/golden/``.  The ResNet-50 arithmetic itself lives in
torchvision (absent, un-vendored): that part is "parity unpinned" against
torchvision proper and is restated from the published architecture.
"""
