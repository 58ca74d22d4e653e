"""
Ray-set assembly of the sources against the numpy restatement of tfrt/sources.py
(oracle/sources.py: SourceBase._resize / make_vars / publish_extra_fields, sources.py:170-315, and
the _internal_update of PointSource / AngularSource / AperatureSource): same rays in the same
ORDER -- a dense source enumerates its domains through tf.meshgrid's 'xy' indexing, which swaps
the first two -- same inherited extra fields.  Runs on the CPU here and, marked gpu, on device
tensors (what the trace consumes).
"""
import numpy as np
import pytest
import torch

from oracle import sources as osrc

GEO2 = ("x_start", "y_start", "x_end", "y_end")
GEO3 = ("x_start", "y_start", "z_start", "x_end", "y_end", "z_end")


def _np(t):
    return t.detach().cpu().numpy()


def _check(src, want, fields):
    for f in fields:
        assert tuple(src[f].shape) == want[f].shape, f
        np.testing.assert_allclose(_np(src[f]), want[f], rtol=0, atol=1e-14, err_msg=f)


def _cases():
    import tensorflowraytrace_amd.distributions as distributions
    import tensorflowraytrace_amd.sources as sources

    # 1. config 1 style: 2-D dense AngularSource, 3 angles x 10 beam points x 2 wavelengths
    angles = distributions.StaticUniformAngularDistribution(-0.1, 0.1, 3)
    beam = distributions.StaticUniformBeam(-1.5, 1.5, 10)
    wl = [680.0, 450.0]
    src = sources.AngularSource(2, (-1.0, 0.25), 0.3, angles, beam, wl, ray_length=2.0)
    want = osrc.angular_source_2d((-1.0, 0.25), 0.3, _np(angles.angles), _np(beam.points), wl, True,
                                  ray_length=2.0)
    assert src["x_start"].shape[0] == 60
    _check(src, want, GEO2 + ("wavelength",))

    # 2. 2-D dense PointSource, converging (rays END on the centre)
    ang = distributions.StaticUniformAngularDistribution(-0.5, 0.5, 7)
    src = sources.PointSource(2, (0.5, -0.5), -0.2, ang, [500.0, 600.0, 700.0], start_on_center=False)
    want = osrc.point_source_2d((0.5, -0.5), -0.2, _np(ang.angles), [500.0, 600.0, 700.0], True,
                                start_on_center=False)
    _check(src, want, GEO2 + ("wavelength",))

    # 3. the bench's source: 3-D undense AperatureSource with an inherited extra field
    a = distributions.StaticUniformCircle(11, 0.2)
    distributions.BasePointTransformation(a, translation=(-10, 0, 0))
    b = distributions.StaticUniformCircle(11, 0.9)
    distributions.BasePointTransformation(b)
    src = sources.AperatureSource(3, a, b, [575.0], dense=False,
                                  extra_fields={"object_coords": ("start_point", a, "points")})
    want = osrc.aperature_source(_np(a.points), _np(b.points), [575.0], False,
                                 {"object_coords": ("start_point", _np(a.points))})
    _check(src, want, GEO3 + ("wavelength", "object_coords"))

    # 4. 3-D DENSE AperatureSource: 4 start points x 5 end points x 2 wavelengths, extra field
    #    attached to the end points
    a = distributions.StaticUniformCircle(4, 0.2)
    distributions.BasePointTransformation(a, translation=(-3, 0, 0))
    b = distributions.StaticUniformCircle(5, 0.9)
    distributions.BasePointTransformation(b)
    tag = np.arange(5, dtype=np.float64) * 10
    src = sources.AperatureSource(3, a, b, [450.0, 650.0], dense=True,
                                  extra_fields={"end_tag": ("end_point", tag)})
    want = osrc.aperature_source(_np(a.points), _np(b.points), [450.0, 650.0], True,
                                 {"end_tag": ("end_point", tag)})
    assert src["x_start"].shape[0] == 40
    _check(src, want, GEO3 + ("wavelength", "end_tag"))
    # meshgrid 'xy': with domains (start, end, wavelength) the END index varies slowest
    np.testing.assert_allclose(_np(src["end_tag"])[:8], 0.0)


def test_sources_equal_the_reference_restatement_on_the_host():
    import tensorflowraytrace_amd as tfa
    import tensorflowraytrace_amd.config as config
    old = config._device
    tfa.set_device("cpu")
    try:
        _cases()
    finally:
        config._device = old


@pytest.mark.gpu
def test_sources_equal_the_reference_restatement_on_the_device():
    import tensorflowraytrace_amd as tfa
    tfa.set_device("cuda:0")
    _cases()
