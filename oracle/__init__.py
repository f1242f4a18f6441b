"""CPU oracle for the FCRN training hot path — TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 restatement of the reference algorithm
(xeTaiz/mono-depth-estimation: network/FCRN.py, criteria.py, metrics.py).  It is
the checker the HIP path is compared against; it is never the product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The shipped package (``mono_depth_estimation_amd``) must not,
and fails loudly when its HIP extension is missing instead of falling back here.

Parity pinning:
  * losses / metrics / FCRN decoder+head: pinned against the reference's own
    code, imported in the build container by ``tests/golden/gen_golden.py``;
    the resulting vectors live in ``tests/golden/*.npz``.
  * ResNet-50 encoder trunk: torchvision is not in ``/root/reference`` nor in the
    image, and the reference has no test vectors for it -> architecture restated
    from the public torchvision definition ("v1.5", stride on the 3x3);
    encoder parity is pinned by our own goldens only.
"""
