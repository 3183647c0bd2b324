"""CPU oracle for the DiffusionSAT denoising-sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / CPU baseline.

The oracle restates the reference's arithmetic (TensorFlow 2.4 code under
``/root/reference``, which cannot be imported here: TensorFlow is not
installed) with every random draw turned into an explicit input.

PARITY UNPINNED for the model path: the reference holds no golden vectors, no
checkpoint and no test for ``QuerySAT.loop``; the third-party arithmetic it
calls (``tensorflow~=2.4.0`` ops, ``tensorflow_probability~=0.12.1``
``Bernoulli.kl_divergence``) is restated from its published definition.  What
pins the oracle instead is listed in DESIGN.md ("Oracle").  The adjacent
numpy-only classes of the reference (``utils/DimacsFile.py``,
``utils/VariableAssignment.py``) ARE importable; their outputs are committed as
fixtures under ``tests/golden/`` (see ``tests/golden/make_dimacs_golden.py``).
"""
