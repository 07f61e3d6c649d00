"""CPU oracle for the PS-NeRF hot path (TEST INFRASTRUCTURE, not product code).

This package is a plain-PyTorch (CPU, fp32 or fp64) restatement of the
reference algorithm for the path named in BASELINE.json's north_star.  It is
pinned against golden vectors captured from the imported reference
(tools/gen_golden.py -> tests/golden/*.npz) and is allowed to be imported ONLY
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the
checker.  The product path (psnerf_amd/) never imports it and fails loudly when
the HIP library is missing.

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
