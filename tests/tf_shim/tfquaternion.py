"""TEST-ONLY placeholder: tfrt/geometry.py imports tfquaternion at module level but none of the
functions exercised by tests/golden/make_reference_golden.py uses it."""


def __getattr__(name):
    raise NotImplementedError(f"tfquaternion.{name} is not available (test shim)")
