"""End-to-end: the example counterparts of the reference's dev scripts run and behave."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def test_hexalens_optimisation_reduces_image_error():
    import tensorflowraytrace_amd.distributions as distributions
    import hexalens
    distributions.seed(7)
    errors, s = hexalens.run(ray_count=6000, steps=24, lens_res_scale=0.2, verbose=False)
    assert len(errors) == 24 and all(np.isfinite(errors))
    first, last = np.mean(errors[:3]), np.mean(errors[-3:])
    assert last < 0.8 * first, (first, last)          # the lens learns to image the object
    s["system"].update()                              # constraints are enforced by update()
    p0, p1 = [p.detach() for p in s["lens"].parameters]
    assert abs(float(p0.min())) < 1e-12
    assert abs(float((p1 - p0).min()) - 0.2) < 1e-12
    assert float(p0.max()) > 1e-4                      # and the surfaces actually moved


def test_single_pass_example_structure():
    import single_pass
    engine, new_rays = single_pass.main()
    res = engine.last_projection_result
    assert set(res["rays"].keys()) <= {"active", "dead", "finished", "stopped"}
    assert res["rays"]["active"]["x_start"].shape[0] == 60          # 10 beam points x 6 wavelengths
    assert set(new_rays.keys()) == {"x_start", "y_start", "x_end", "y_end", "wavelength"}
    assert set(res["optical"].keys()) >= {"mat_in", "mat_out"}
    # shorter wavelengths refract more strongly: |angle| grows as wavelength drops
    ang = torch.atan2(new_rays["y_end"] - new_rays["y_start"], new_rays["x_end"] - new_rays["x_start"])
    wl = new_rays["wavelength"]
    top = new_rays["y_start"] == new_rays["y_start"].max()
    order = torch.argsort(wl[top])
    a = ang[top][order].abs().cpu().numpy()
    assert np.all(np.diff(a) < 0)
