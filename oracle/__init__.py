"""oracle/ — CPU restatement of the reference's fitness path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker / the timed CPU baseline — never as the
thing shipped.  The product (clip_glass_amd + libglass.so) never imports it and
fails loudly when the HIP extension is missing.

Parity pinning: the reference has no tests / golden vectors of its own
(SURVEY.md §4), so the oracle is pinned against the reference itself, imported
verbatim in the build container (tests/golden/ref_harness.py,
tests/test_oracle_vs_reference.py — skipped where /root/reference is absent) and
against the fixtures that import generated (tests/golden/*.npz, made by
tests/golden/make_golden.py).  Unpinned third-party pieces (source absent from
/root/reference): kornia==0.4.1 resize (restated as bilinear,
align_corners=False, no antialias).
"""
