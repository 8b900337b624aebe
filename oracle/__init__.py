"""CPU oracle for the deephar pose-regression hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain CPU restatement (PyTorch-CPU fp32/fp64 + NumPy fp64) of what the reference
computes on the path named by BASELINE.json `north_star`; every function cites the reference file:line it
follows.  It exists so that tests/, __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py can
check / time something; the product package `deephar_amd` never imports it and fails loudly when its HIP
library is missing.

PARITY: PARTLY PINNED.  The reference's arithmetic lives in tensorflow-gpu==1.6.0 via keras==2.1.4
(reference requirements.txt:2-3); neither is installed or installable here (no network), and the reference
ships no tests, golden vectors or fixtures.  What IS pinned, by running reference code
in the build container and committing the outputs (tests/golden/, scripts alongside):
  (1) everything the reference's own Python decides -- graph wiring, layer order, slice indices, constants,
      frozen soft-argmax weights, output order of ReceptionNet / merge / SPNet: deephar/layers.py, activations.py
      and models/*.py are imported UNMODIFIED and executed on oracle/refrun/minikeras.py (a PyTorch-CPU stand-in
      for the ~40 Keras/TF entry points they call); this oracle agrees with those outputs to 1e-9 in fp64
      (tests/test_reference_golden.py), and the HIP engine is tested against the same vectors;
  (2) the NumPy-only modules run as they are: soft-argmax grids (utils/math.py) bit-exact, affine / camera
      post-processing, metrics, pose-layout tables (tests/test_golden_host.py).
What stays UNPINNED (restated from the Keras 2.1.4 / TF 1.6 defaults of SURVEY.md A.3, checked only by analytic
known answers in tests/test_oracle_ops.py and an independent NumPy-fp64 decoder): the numerics inside the
Keras layers themselves -- TF-"SAME" padding, BatchNormalization epsilon, pooling / up-sampling semantics.
"""
