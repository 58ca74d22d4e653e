"""TEST-ONLY placeholder: tfrt/distributions.py imports imageio at module level; nothing exercised
by tests/golden/make_reference_host_golden.py reads an image."""
