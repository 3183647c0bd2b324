#!/usr/bin/env python3
"""Convert a trained DiffusionSAT checkpoint (TF2 object-graph format) to the ``.npz`` weight file
``diffusionsat_amd`` loads (``diffusionsat_amd.weights.save_npz`` layout).

Run this where TensorFlow is installed (the training machine); it never travels as a dependency of
the package and cannot be exercised in the build container (no TensorFlow, no checkpoint in the
reference: README.md:38-40 points to GitHub Releases).  The package's own pure-Python reader
(``diffusionsat_amd/tf_checkpoint.py``) reads the same files without TensorFlow, but has never seen
a TensorFlow-written file either; this script is the path that relies only on TensorFlow's reader.

    python tools/export_tf_checkpoint.py <model_dir_or_ckpt_prefix> weights.npz

It reads exactly the variables ``tf.train.Checkpoint(step=..., model=QuerySAT).restore(...)`` restores
(satuniformity/DiffusionSampler.py:215-227): for each of the five MLPs of model/query_sat.py:117-122
the Dense kernels [in,out] and biases [out] of model/mlp.py:24,37-39, stored under
``model/<mlp>/dense_layers/<i>/{kernel,bias}/.ATTRIBUTES/VARIABLE_VALUE``.
"""
import sys

import numpy as np

MLPS = ("variables_query", "lit_mlp", "clause_mlp", "update_gate", "variables_output")
SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"


def main(src, dst):
    import tensorflow as tf   # only here
    prefix = tf.train.latest_checkpoint(src) or src
    reader = tf.train.load_checkpoint(prefix)
    shapes = reader.get_variable_to_shape_map()
    out, missing = {}, []
    for mlp in MLPS:
        i = 0
        while True:
            k = "model/%s/dense_layers/%d/kernel%s" % (mlp, i, SUFFIX)
            b = "model/%s/dense_layers/%d/bias%s" % (mlp, i, SUFFIX)
            if k not in shapes:
                break
            out["%s/dense_layers/%d/kernel" % (mlp, i)] = np.asarray(reader.get_tensor(k), dtype=np.float32)
            out["%s/dense_layers/%d/bias" % (mlp, i)] = np.asarray(reader.get_tensor(b), dtype=np.float32)
            i += 1
        if i == 0:
            missing.append(mlp)
    if missing:
        sys.exit("checkpoint %s has no variables for %s; keys look like: %s"
                 % (prefix, missing, sorted(shapes)[:8]))
    np.savez(dst, **out)
    print("wrote %s: %d arrays from %s" % (dst, len(out), prefix))
    for k in sorted(out):
        print("  %-44s %s" % (k, out[k].shape))


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    main(sys.argv[1], sys.argv[2])
