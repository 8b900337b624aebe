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
What stays UNPINNED (restated from the Keras 2.1.4 / TF 1.6 defaults of SURVEY.md A.3): the numerics inside the
Keras layers themselves -- TF-"SAME" padding, BatchNormalization epsilon, pooling / up-sampling semantics.  They are
checked three ways that do not involve Keras: analytic known answers (tests/test_oracle_ops.py), an independent
NumPy-fp64 decoder, and -- so that this module and the mini-Keras stand-in (same author, same torch back-end) are not
each other's only witness -- plain NumPy float64 LOOP implementations of conv / separable conv / max-pool / max-min
pool / up-sampling / BatchNormalization written from the TensorFlow op documentation
(tests/test_oracle_ops.py::test_layer_semantics_against_independent_numpy_loops).  The kit that closes the gap with the
real Keras is tools/make_keras_parity_kit.py (needs a machine with keras==2.1.4 + tensorflow==1.6).
"""
