"""CPU oracle for the InsTaG 3DGS render/train hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``instag_amd/`` (the product) may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

Parity status (see DESIGN.md "Oracle"):
  * rasterizer  -- PARITY UNPINNED against the reference rasterizer: the
    reference's ``diff_gauss`` (slothfulxtx/diff-gaussian-rasterization,
    un-vendored, unpinned submodule, /root/reference/.gitmodules:4-6) is not
    in the container.  The restatement follows the published 3DGS algorithm
    (Kerbl et al. 2023) constrained by the reference's call sites and is
    pinned only through the reference's importable Python helpers
    (eval_sh, projection/view matrices, covariance construction).
  * grid / SH encoders -- restated from gridencoder/src/gridencoder.cu and
    shencoder/src/shencoder.cu; SH pinned by golden vectors evaluated from
    the reference's own polynomial table, grid by the reference's offsets
    table and vertex / midpoint / out-of-range identities.
"""
