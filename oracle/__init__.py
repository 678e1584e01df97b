"""CPU oracle for the RenderNet forward render path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (NumPy for the resampler / binvox / Phong code,
torch-CPU ``nn.functional`` with explicit TF-``SAME`` padding for the convolutions) of the
reference's TensorFlow-1 graph.  It exists to *check* the HIP path; it is never the thing
that is shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``rendernet_amd/`` does.

PARITY UNPINNED: the reference ships no tests, no golden vectors, no seeds and no trained
weights, and its arithmetic lives in TensorFlow 1.8 (``demo/requirement.txt:1``)
which is not installable in the build container -- so this oracle cannot be
checked against outputs of the reference itself.  What pins it instead:
  * analytic known-answer tests of the resampler (identity-like pose, axis-aligned poses,
    out-of-range cancellation) -- tests/test_oracle_resample.py;
  * a loop-level restatement of TF's documented SAME / conv_transpose definitions on small
    cases -- tests/test_oracle_layers.py;
  * the five shipped binvox fixtures decoded and compared with the survey's occupancy counts.

Each function cites the reference file:line it follows (paths relative to /root/reference).
"""
