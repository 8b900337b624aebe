"""CPU oracle for the deephar pose-regression hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain CPU restatement (PyTorch-CPU fp32/fp64 + NumPy fp64) of what the reference
computes on the path named by BASELINE.json `north_star`; every function cites the reference file:line it
follows.  It exists so that tests/, __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py can
check / time something; the product package `deephar_amd` never imports it and fails loudly when its HIP
library is missing.

PARITY UNPINNED: the reference's arithmetic lives in tensorflow-gpu==1.6.0 via keras==2.1.4
(reference requirements.txt:2-3); neither is installed or installable here (no network), and the reference
ships no tests, golden vectors or fixtures.  The oracle is therefore anchored on
  (1) the reference sources cited per function plus the Keras/TF defaults listed in SURVEY.md A.3,
  (2) analytic known-answer tests (tests/test_oracle_ops.py),
  (3) agreement between two independent implementations (torch fp32/fp64 vs NumPy fp64) of the decoder ops,
not on outputs of the reference itself.
"""
