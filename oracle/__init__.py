"""CPU oracle for the fusion hot path -- TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the reference's cross-attention fusion
model (forward, loss, backward, clip, AdamW).  It exists so that the HIP path
can be checked against an independent implementation on a box where the
reference itself is absent.

``torch_port.py`` restates the same training step with torch CPU ops and autograd (the reference's own CPU path runs
on those kernels); ``bench.py`` times it as the fairer CPU baseline and a test holds it to the golden step.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``camouflage_multimodal_amd/`` does, and the
product path raises when its HIP extension is missing instead of falling back
to this code.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced in the build
container by importing the reference's own ``fusion_model.py`` /
``train_multimodal.py`` (script: ``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` holds this oracle to them.
"""
