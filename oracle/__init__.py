"""CPU oracle for the uncertainty-guided sampling hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain fp32 CPU restatement
(torch-CPU / numpy) of the algorithms of the upstream reference
(Michedev/diffusion-uncertainty); every function cites the reference file:line
it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it - as the checker, never as
the thing measured or shipped.  The product package
(``diffusion-uncertainty_amd/``) never imports it and has no CPU fallback.

Pinning: the reference's own tests hold no golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the
reference itself, generated in the build container by
``tests/golden/make_golden.py`` (which imports the reference modules from
/root/reference through the diffusers stand-in of ``tests/golden/refimport.py``)
and committed as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py``
checks every oracle function against those vectors.
"""
