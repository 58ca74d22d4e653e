"""
oracle/ -- CPU restatement (torch-CPU float64) of the tfrt hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
there only as the checker / the timed CPU baseline -- never as the thing shipped.  The
product package ``tensorflowraytrace_amd`` does not import this package anywhere.

What it restates (citations are into the upstream reference, ecpoppenheimer/TensorFlowRayTrace):

* ``oracle.geom``    -- tfrt/geometry.py  (line x line, line x triangle, line x circle,
                        Snell 2-D / 3-D, angle_in_interval), same operation order, float64.
* ``oracle.tracer``  -- tfrt/engine.py (merge order, ``_intersection``,
                        ``_segment_intersection``, ``_arc_intersection``, ``_seg_or_arc``,
                        ``_get_arc_norm``, ``process_projection_2D/3D``, ``single_pass``,
                        ``ray_trace``), tfrt/operation.py ``StandardReaction.main``,
                        tfrt/materials.py, tfrt/boundaries.py ``update_fields_from_vertices``.

Like the reference it materialises dense (M boundaries x N rays) float64 temporaries;
``tracer`` chunks over rays so the temporaries stay bounded.  Because it is written with
torch ops, ``torch.autograd`` through it plays the role of ``tf.GradientTape`` and is the
gradient oracle for the hand-derived HIP backward.

Pinning status
--------------
TensorFlow is not installed in the build container (ordinary ``ModuleNotFoundError``, see
SURVEY.md section 8c), so the reference cannot be executed as published.  Two kinds of evidence
pin this restatement:

1. The reference's own test properties for this path (tests/geometry/test_line_intersect_1to1.py,
   test_line_circle_intersect_1to1.py, test_angle_in_interval.py), restated in
   tests/test_oracle_reference_properties.py.  They cover the 2-D geometry functions only.
2. Outputs of the reference's OWN SOURCE FILES, executed in the build container with a minimal
   stand-in for the ``tensorflow`` module (tests/tf_shim: ~60 tf names mapped one-to-one onto
   torch-CPU float64 ops, each a single correctly rounded operation; ``tf.GradientTape`` ->
   ``torch.autograd``; Keras SGD -> ``var -= 0.01 * grad``; placeholder ``pyvista`` /
   ``imageio`` / ``tfquaternion`` modules that are imported but never used).  The generating
   scripts and their fixtures are committed under tests/golden/ (make_reference_*.py,
   reference_*.npz):
     - tfrt/geometry.py: raw_line_intersect, raw_line_triangle_intersect,
       raw_line_circle_intersect, angle_in_interval, snells_law_2D, snells_law_3D;
     - tfrt/engine.py + operation.py + materials.py: a 4-pass 3-D trace of the lens scene (and,
       through torch.autograd over the reference's op sequence, its parameter gradients) and three
       4-pass 2-D traces (arcs, segments, both -- the last with the reference's mis-paired concat);
     - tfrt/boundaries.py: parametric / multi / master-slave triangle boundaries, constraints,
       vector generators, the 2-D segment arithmetic; tfrt/sources.py + distributions.py: the
       3-D AperatureSource, dense and undense, and the static distributions;
     - tfrt/optimizer.py on top of all of them: six SGD_Optimizer steps + smoothing.
   oracle.geom / oracle.tracer reproduce the geometry and trace fixtures BIT FOR BIT
   (tests/test_reference_golden.py); the product reproduces all of them (HIP path: bit for bit
   where no libm function is involved, 1e-9 / 1e-12 otherwise).
   What this does and does not show: the reference's formulas, operation order, masking, merge
   and classification order, inheritance and optimiser bookkeeping are executed, not re-read;
   TensorFlow's own kernels are not (the stand-in supplies IEEE float64 arithmetic, which is what
   Eigen's CPU kernels compute).  Classes the reference cannot construct at its HEAD (2-D
   PointSource / AngularSource: sources.py:447-449, 660-662; ParametricSegmentBoundary:
   boundaries.py:487) have no such vectors, and 3-D source rotations (tfquaternion) stay unpinned.

Build-authored evidence on top: analytic optics cases and finite differences
(tests/test_oracle_analytic.py) and the older oracle-generated vectors (tests/golden/lens3d.npz,
geometry.npz, scene2d.npz from make_golden.py).
"""
