/*
 * tfrt_hip.h -- C ABI of libtfrt_hip.so, the MI355X (gfx950) implementation of the tfrt
 * hot path: batched ray x boundary intersection, nearest-hit reduction, classification /
 * stable compaction, Snell refraction / reflection, iterated over passes, plus the
 * hand-derived reverse sweep.
 *
 * The upstream reference (ecpoppenheimer/TensorFlowRayTrace) is pure Python on TensorFlow
 * eager ops: it has NO native/FFI boundary on this path.  Each entry point below therefore
 * cites the Python seam it replaces (file:line in the reference); INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add at that seam.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensors); nothing is
 *    allocated, freed or retained; no global state; re-entrant
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *    synchronises, so a caller may capture a call into a hipGraph
 *  - ray sets are SoA: a "ray block" is 6 rows (3-D: xs,ys,zs,xe,ye,ze) or 4 rows
 *    (2-D: xs,ys,xe,ye) of `stride` elements each; `state_dtype` selects the element type
 *    of ray blocks (TFRT_F32 / TFRT_F64 / TFRT_F16).  Geometry, indices of refraction and gradients
 *    are always float64.
 *  - return value: 0 on success, negative TFRT_E_* otherwise (see tfrt_strerror)
 */
#ifndef TFRT_HIP_H
#define TFRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFRT_VERSION 107 /* 0.1.0 */

#define TFRT_F32 0
#define TFRT_F64 1
#define TFRT_F16 2 /* storage-only experiment (BASELINE config 5): half ray state, math stays f32/f64 */

/* boundary catagory, tfrt/engine.py:14-16 */
#define TFRT_OPTICAL 0
#define TFRT_STOP 1
#define TFRT_TARGET 2

/* ray classes produced by a pass (tfrt/engine.py:2027-2111) */
#define TFRT_CLS_ACTIVE 0
#define TFRT_CLS_FINISHED 1
#define TFRT_CLS_STOPPED 2
#define TFRT_CLS_DEAD 3

/* flags (OpticalEngine constructor kwargs, tfrt/engine.py:1216-1230) */
#define TFRT_COMPILE_ACTIVE 1u
#define TFRT_COMPILE_FINISHED 2u
#define TFRT_COMPILE_STOPPED 4u
#define TFRT_COMPILE_DEAD 8u

#define TFRT_E_BADARG (-1)
#define TFRT_E_WORKSPACE (-2)
#define TFRT_E_LAUNCH (-3)
#define TFRT_E_UNSUPPORTED (-4)

int tfrt_version(void);
const char* tfrt_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Triangle boundaries: vertices -> per-face data.
 * Replaces TriangleBoundaryBase.update_fields_from_vertices, tfrt/boundaries.py:890-923
 * (gather faces, cross, normalize) and, in reverse, the tape's scatter through that gather
 * with the per-corner stop_gradient mask (`vertex_update_map`, boundaries.py:900-913).
 *
 *   vertices    (V,3) f64 row-major
 *   faces       (F,3) i32 vertex indices
 *   face_verts  (F,9) f64 : xp,yp,zp,x1,y1,z1,x2,y2,z2 per face
 *   norm        (F,3) f64 : normalize((P1-P0) x (P2-P1))
 */
int tfrt_build_faces_forward(const double* vertices, int64_t n_vertices, const int32_t* faces,
                             int64_t n_faces, double* face_verts, double* norm, void* stream);

/* grad_norm and update_mask may be NULL.  update_mask (F,3) u8: 0 = stop_gradient for that
 * corner.  corner_start (V+1) / corner_list (3F) i32, both or neither: the face corners f*3+c
 * sorted by the vertex they reference (corner_list[corner_start[v] .. corner_start[v+1]) are
 * vertex v's).  With them the reverse is a gather -- one lane per vertex sums its corners in
 * list order: no atomics, grad_vertices (V,3) is OVERWRITTEN and the result is bit-identical
 * from run to run.  Without them (NULL) it scatters with float64 atomics and ACCUMULATES into
 * grad_vertices (caller zeroes). */
int tfrt_build_faces_backward(const double* grad_face_verts, const double* grad_norm,
                              const double* face_verts, const int32_t* faces,
                              const uint8_t* update_mask, int64_t n_faces, int64_t n_vertices,
                              const int32_t* corner_start, const int32_t* corner_list,
                              double* grad_vertices, void* stream);

/* Parametric surface: parameters -> per-face data in one launch.
 * Replaces ParametricTriangleBoundary._update, tfrt/boundaries.py:1065-1078
 * (`zero_points + parameters[:,None] * vectors`, boundaries.py:1080-1092) followed by
 * update_fields_from_vertices (boundaries.py:890-923); bit-identical with the two-step path
 * (product and sum stay separate roundings).
 *
 *   zero_points, vectors  (V,3) f64 row-major;  parameters (V,) f64
 * The reverse sums, per face corner that update_mask lets through, dot(grad of that corner,
 * vectors[vertex]) into grad_parameters (V,): as a gather when corner_start / corner_list are
 * given (see tfrt_build_faces_backward: overwritten, deterministic), else with atomics
 * (ACCUMULATED, caller zeroes). */
int tfrt_param_faces_forward(const double* zero_points, const double* vectors,
                             const double* parameters, int64_t n_vertices, const int32_t* faces,
                             int64_t n_faces, double* face_verts, double* norm, void* stream);
int tfrt_param_faces_backward(const double* grad_face_verts, const double* grad_norm,
                              const double* face_verts, const int32_t* faces,
                              const uint8_t* update_mask, const double* vectors, int64_t n_faces,
                              int64_t n_vertices, const int32_t* corner_start,
                              const int32_t* corner_list, double* grad_parameters, void* stream);

/* The same for several surfaces of one optical system with ONE launch each way -- all the
 * parametric boundaries of OpticalSystem3D.update (engine.py:971-1018 merges their fields with
 * tf.concat afterwards): every surface writes its rows of the system's merged (M, 9) block
 * directly, and a surface with `copy_from` (a fixed boundary: target, stop) has its rows copied
 * into its place, so the merged block needs no concatenation.  At most TFRT_MAX_SURFACES
 * descriptors (a host array; they travel by value in the kernel arguments).  The reverse is the
 * gather form of tfrt_param_faces_backward per surface (corner lists required). */
#define TFRT_MAX_SURFACES 8
typedef struct tfrt_face_surface {
  const double* zero_points;   /* (V,3)  unused with copy_from */
  const double* vectors;       /* (V,3) */
  const double* parameters;    /* (V,) */
  const int32_t* faces;        /* (F,3) vertex indices */
  int64_t n_vertices, n_faces;
  double* face_verts;          /* (F,9) out: this surface's rows of the merged block */
  double* norm;                /* (F,3) out, or NULL */
  const double* copy_from;     /* (F,9) fixed faces to copy instead, or NULL */
} tfrt_face_surface;
typedef struct tfrt_face_surface_grad {
  const double* grad_face_verts;  /* (F,9) or NULL */
  const double* grad_norm;        /* (F,3) or NULL */
  const double* face_verts;       /* (F,9) forward result (needed with grad_norm) */
  const uint8_t* update_mask;     /* (F,3) or NULL */
  const double* vectors;          /* (V,3) */
  const int32_t* corner_start;    /* (V+1) */
  const int32_t* corner_list;     /* face * 3 + corner, grouped by vertex */
  int64_t n_vertices;
  double* grad_parameters;        /* (V,) out (overwritten) */
} tfrt_face_surface_grad;
int tfrt_param_faces_forward_multi(const tfrt_face_surface* surfaces, int32_t n_surfaces,
                                   void* stream);
int tfrt_param_faces_backward_multi(const tfrt_face_surface_grad* surfaces, int32_t n_surfaces,
                                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser step, parameter side (SGD_Optimizer.process_gradient / single_step,
 * tfrt/optimizer.py:223-257 and :316).
 *
 * tfrt_sgd_process: per element, in the tensors' own dtype (TFRT_F32 / TFRT_F64)
 *     g = isfinite(grad) ? grad : 0            (optimizer.py:226-229)
 *     g = g * scale                            (:233, scale = lr_scale*individual_lr*learning_rate)
 *     g = min(max(g, -clip), clip)             (:236-247, clip >= 0)
 *     processed[i] = g                         (skipped when processed == NULL)
 *     param[i]    -= sgd_learning_rate * g     (skipped when param == NULL; the Keras SGD
 *                                               apply of optimizer.py:316, no momentum)
 * `processed` may alias `grad`.
 */
int tfrt_sgd_process(const void* grad, void* processed, void* param, int64_t n, int32_t dtype,
                     double scale, double clip, double sgd_learning_rate, void* stream);

/* Same, with {scale, clip, sgd_learning_rate} read from three float64 on the DEVICE: the values
 * change from step to step (learning-rate schedules, optimizer.py:384-389) while a captured
 * launch graph of the step replays unchanged. */
int tfrt_sgd_process_dev(const void* grad, void* processed, void* param, int64_t n, int32_t dtype,
                         const double* hyper, void* stream);

/* tfrt_sgd_process_dev for up to 8 float64 tensors in ONE launch (the per-parameter loop of
 * optimizer.py:223-247, 316): host arrays of n_tensors device pointers (`processed` / `param`
 * and their entries may be NULL as above) and element counts; `hyper` holds {scale, clip,
 * sgd_learning_rate} per tensor, 3 * n_tensors float64 on the device. */
/* The second stage of tfrt_goal_error3d's error sum, left to the caller
 * (tfrt_goal_error3d_deferred): a dependent launch costs ~4.5 us whatever it does, and nothing on
 * the device waits for the sum. */
typedef struct tfrt_goal_pending {
  const double* partial;       /* per-workgroup partial sums (in the goal workspace) */
  int32_t n_partial;
  const int32_t* n_finished;   /* device count of finished rays */
  int32_t n_fields;
  double* error_out;           /* {sum, n_terms, mean} */
  const int32_t* tests_lo_hi;  /* the trace's test count (two int32) or NULL */
  int64_t* tests_total;        /* running total it is added to, or NULL */
  /* A reverse sweep over an in-place tape (tfrt_scene3d.in_place) whose trace was not asked for
   * ray sets counts for itself: per partial sum two int32 {finished rays, passes the wavefront's
   * rays entered}.  When given, the number of error terms and the trace's test count (passes x
   * n_faces) are summed from them instead of being read through n_finished / tests_lo_hi, and
   * left in counts_tail ([1] finished total, [4] / [5] test count) if that is given too. */
  const int32_t* partial_counts;
  int64_t n_faces;
  int32_t* counts_tail;
} tfrt_goal_pending;

int tfrt_sgd_process_multi(int32_t n_tensors, const void* const* grad, void* const* processed,
                           void* const* param, const int64_t* n, const double* hyper,
                           void* stream);

/* ... and finishes a pending error sum in one more workgroup of the same launch. */
int tfrt_sgd_process_multi_finish(int32_t n_tensors, const void* const* grad,
                                  void* const* processed, void* const* param, const int64_t* n,
                                  const double* hyper, const tfrt_goal_pending* pending,
                                  void* stream);

/* y = A x, A in CSR form (int64 indices, f64 values): the accumulator (optimizer.py:250-255)
 * and smoother (optimizer.py:277-282) products for the sparse matrices the mesh tools
 * produce (mesh_tools.py:221-421).  x and y must not alias. */
int tfrt_csr_matvec(const int64_t* crow_indices, const int64_t* col_indices, const double* values,
                    const double* x, double* y, int64_t n_rows, void* stream);

/* ------------------------------------------------------------------------------------------
 * Scene description shared by the 3-D entry points (the merged boundary set of
 * OpticalSystem3D._merge_boundaries, tfrt/engine.py:971-1018: optical, stop, target order).
 */
typedef struct tfrt_scene3d {
  const double* face_verts; /* (M,9) f64 */
  const int32_t* catagory;  /* (M) TFRT_OPTICAL / _STOP / _TARGET */
  const int32_t* mat_in;    /* (M) material index opposite the norm, or NULL ("value" mode) */
  const int32_t* mat_out;   /* (M) material index on the norm side, or NULL */
  const double* n_in;       /* (M) per-face refractive index ("value" mode) or NULL */
  const double* n_out;      /* (M) or NULL */
  int64_t n_faces;          /* M */
  /* n_table[m * n_table_stride + source_ray] = n_m(wavelength of that source ray); the
   * host evaluates the material callables once per source ray (wavelength is inherited
   * unchanged, tfrt/operation.py:238-239).  NULL in "value" mode. */
  const double* n_table;
  int64_t n_table_stride;
  int32_t n_materials;
  double intersect_epsilion; /* OpticalSystemBase kwargs, tfrt/engine.py:174-190 */
  double size_epsilion;
  double ray_start_epsilion;
  /* (M) u8, backward only: 0 = this face's vertices are constants (e.g. a target plane),
   * skip its gradient accumulation; NULL = accumulate for every face. */
  const uint8_t* face_grad_mask;
  /* (M) i32, optional: a permutation of the face indices that puts spatially close faces next
   * to each other (e.g. Morton order of the centroids).  When given (and M >= 64) the trace
   * visits faces in clusters of 16 consecutive entries behind a bounding-sphere test (a
   * two-level conservative filter); results are identical to the all-pairs path (NULL). */
  const int32_t* cluster_order;
  /* 1: every source ray has the same wavelength -- n_table has ONE column, n_table[m *
   * n_table_stride] = n_m(that wavelength), read by every ray (a source made from one wavelength,
   * e.g. dev/hexalens.py:48 `[drawing.YELLOW]`).  0: one column per source ray. */
  int32_t n_table_uniform;
  /* Reverse sweep only.  0 (default): the 9 face-gradient terms of every ray are summed with
   * float64 atomics (LDS windows, then global): fastest, but float64 addition is not
   * associative, so the last bits of a face's sum depend on the arrival order and differ from
   * run to run.  1: ordered mode -- per pass the terms are scaled by a power of two taken from
   * their largest magnitude, rounded to 64-bit integers and summed with integer atomics (exact,
   * hence order-independent), then converted back: bit-identical results on every run, at a
   * resolution of 2^-40 of the pass's largest term. */
  int32_t deterministic;
  /* With cluster_order.  1: the caller hands the source rays over in a COHERENT order --
   * neighbouring rays have neighbouring lines, e.g. sorted along a Hilbert curve through their
   * aperture points (tensorflowraytrace_amd.ops.ray_order).  A wavefront of 64 consecutive rays is
   * then tried as ONE narrow bundle that walks the face hierarchy once for all of them
   * (k_intersect_beam; wavefronts that are no narrow bundles are cut, or left to the grouped
   * per-ray walk), and the reverse sweep sums the face gradients of a wavefront's rays before
   * touching memory.  The outputs are what they always are -- every class compacted stably in the
   * order of the rays handed in -- so a caller that sorted its rays maps the ray ids back and
   * restores its own order per pass (the Python engine does).  Results never depend on the flag;
   * with rays that are not coherent it only costs time. */
  int32_t coherent_rays;
  /* With coherent_rays.  0: behind k_intersect_beam the grouped kernel is launched for the
   * wavefronts that are no (few) narrow bundles; their number, summed over the passes, comes back
   * in counts[TFRT_COUNTS_LEN - 1].  1: no such launch -- k_intersect_beam finishes every
   * wavefront itself, cutting it down to single rays if need be: always correct, but slow for
   * rays that are not coherent; meant for a source whose earlier traces left no wavefront over
   * (saves one kernel launch per pass). */
  int32_t coherent_only;
  /* Reverse sweep only, "value" mode (n_in / n_out given): (M) f64 each, ACCUMULATED into, or
   * NULL -- d error / d n_in[face], d error / d n_out[face] (tfrt/operation.py:268-272 reads the
   * two fields as ordinary tensors, which a tape can differentiate).  Both or neither. */
  double* grad_n_in;
  double* grad_n_out;
  /* Forward only, optional: clear_count f64 at clear_buffer are set to zero by the trace's set-up
   * launch -- the (M,9) block a reverse sweep will ACCUMULATE into (tfrt_trace3d_backward,
   * tfrt_trace3d_backward_goal), cleared without a launch of its own.  NULL / 0: nothing. */
  double* clear_buffer;
  int64_t clear_count;
  /* With coherent_rays and cluster_order, max_passes >= 1.  1: ALL passes of the trace run in ONE
   * launch with every ray kept in its slot (k_trace_inplace): a lane intersects, classifies and
   * refracts its ray pass after pass, the tape holds one record per pass at slot = ray index, and
   * nothing is compacted between passes -- StandardReaction emits exactly one child per active ray
   * (tfrt/operation.py:255-307), so the reference's per-pass boolean_mask (tfrt/engine.py:2069-2111)
   * is not needed to go on tracing.  The reference's output order is made afterwards: a scan of
   * per-wavefront class counts always (it fills `counts`), and a gather of the class rows only when
   * tfrt_trace3d_forward is given somewhere to put them (any tfrt_ray_out with rays, or
   * `unfinished`) -- or later by tfrt_trace3d_compact.  Outputs, counts and gradients are those of
   * the per-pass path bit for bit (reverse-sweep sums: to the last bits of a differently ordered
   * sum).  Every wavefront is finished by the one kernel, however wide its bundle (like
   * coherent_only, which it implies): meant for sources whose earlier traces left no wavefront
   * over.  The SAME value must be passed to the reverse sweep.  Ignored (per-pass path) when the
   * conditions above do not hold.
   * `counts`: a forward call that is given no room for ray sets does not run the scan either -- a
   * folded reverse sweep (tfrt_trace3d_backward_goal) needs neither and counts the finished rays
   * and the tests itself; tfrt_trace3d_compact then fills `counts` together with the sets.
   * 2: as 1, and the FINISHED rays are handed out where they are -- `finished` (capacity >= n_rays,
   * the only class given) receives ray r's finished row (start, hit point on the target) at
   * column r, finished->face[r] = the target face hit or -1 when ray r did not finish (its column
   * then holds the source ray: finite stand-in values that carry no gradient), finished->ray_id[r]
   * = the number of passes ray r entered; nothing is compacted, `counts` is not written.  For
   * error functions that work row by row on fixed-shape tensors (no ray count to read back:
   * tfrt/optimizer.py:216-220 with a mask instead of a boolean_mask); tfrt_trace3d_backward then
   * takes grad_finished in the same layout (capacity >= n_rays, no other class gradient). */
  int32_t in_place;
  /* With in_place, optional: (n_rays) i32, ray_slot[r] = the column of src_rays that holds the
   * CALLER's ray r -- the rays were handed over in another order than the caller's own, e.g. the
   * coherent order of tfrt_ray_order (ray_slot is that permutation's inverse).  The ray sets are
   * then compacted in the caller's numbering: every class lists, pass after pass, its rays by
   * ascending r with ray_id = r, exactly what a trace of the rays in the caller's order returns
   * (tfrt/engine.py:2069-2111, 1379-1403), and class gradients are read in that order by the
   * reverse sweep -- no tfrt_restore_order afterwards.  NULL: the order of src_rays. */
  const int32_t* ray_slot;
} tfrt_scene3d;

/* One class of output rays (finished / active history / stopped / dead), compacted stably in
 * source order pass after pass, exactly like the reference's per-pass boolean_mask + concat
 * (tfrt/engine.py:2069-2111, 1379-1399). */
typedef struct tfrt_ray_out {
  void* rays;       /* 6 x capacity ray block (state dtype) or NULL if not compiled */
  int32_t* ray_id;  /* (capacity) index of the source ray each output row descends from */
  int32_t* face;    /* (capacity) merged face index hit (-1 for dead rays) */
  int64_t capacity; /* row stride of `rays` */
} tfrt_ray_out;

/* Bytes of scratch+tape the trace needs.  The SAME buffer must be passed to forward and,
 * untouched, to backward. */
size_t tfrt_trace3d_workspace_bytes(int64_t n_rays, int64_t n_faces, int32_t max_passes,
                                    int32_t state_dtype);

/* Number of int32 in `counts`: per pass 8 ints {n_active,n_finished,n_stopped,n_dead,
 * base_active,base_finished,base_stopped,base_dead}, then 8 trailing ints:
 * {total_active,total_finished,total_stopped,total_dead,n_tests_lo,n_tests_hi,error,
 *  wavefronts left to the grouped kernel (coherent-ray traces)}. */
#define TFRT_COUNTS_PER_PASS 8
#define TFRT_COUNTS_LEN(max_passes) (8 * ((max_passes) + 1))

/* Whole trace, forward.  Replaces OpticalEngine.ray_trace -> single_pass ->
 * process_projection_3D -> OpticalSystem3D.intersect/_intersection ->
 * geometry.line_triangle_intersect, then StandardReaction.main -> geometry.snells_law_3D
 * (tfrt/engine.py:2311-2330, 2193-2302, 1988-2191, 1020-1166; tfrt/geometry.py:191-320,
 * 671-753; tfrt/operation.py:255-307).
 *
 *   src_rays   6 x src_stride ray block, n_rays valid columns
 *   unfinished ray block (6 x n_rays) receiving the rays still active after the last pass
 *              (what single_pass returns, engine.py:2302), + their source ids; may be NULL
 *   counts     TFRT_COUNTS_LEN(max_passes) int32, written by the device
 */
int tfrt_trace3d_forward(const void* src_rays, int64_t src_stride, int64_t n_rays,
                         const tfrt_scene3d* scene, double new_ray_length,
                         double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                         uint32_t flags, tfrt_ray_out* finished, tfrt_ray_out* active,
                         tfrt_ray_out* stopped, tfrt_ray_out* dead, void* unfinished,
                         int32_t* unfinished_id, int32_t* counts, void* workspace,
                         size_t workspace_bytes, void* stream);

/* The class outputs of an in-place trace (tfrt_scene3d.in_place) that tfrt_trace3d_forward was
 * given no room for: every class compacted stably into `finished` / `active` / `stopped` / `dead`
 * (+ `unfinished`), rows recomputed from the tape in `workspace`, exactly as a forward call with
 * these outputs would have left them, and `counts` filled like that call would have filled it (its
 * error flag is set when a capacity is exceeded).  Also records every tape entry's output row,
 * which tfrt_trace3d_backward / _backward_goal need to read class gradients (grad_finished, ...):
 * call it (or give forward its outputs) before a reverse sweep that is handed any.  Same
 * src_rays / n_rays / max_passes / state_dtype / flags / dead_ray_length as the forward call;
 * ray_slot as tfrt_scene3d.ray_slot (NULL: the sets in the order of src_rays).
 * Replaces the ray-set properties of tfrt/engine.py:1379-1403 for a trace whose sets are cut
 * lazily. */
int tfrt_trace3d_compact(const void* src_rays, int64_t src_stride, int64_t n_rays,
                         double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                         uint32_t flags, tfrt_ray_out* finished, tfrt_ray_out* active,
                         tfrt_ray_out* stopped, tfrt_ray_out* dead, void* unfinished,
                         int32_t* unfinished_id, int32_t* counts, int64_t n_faces,
                         const int32_t* ray_slot, void* workspace, size_t workspace_bytes,
                         void* stream);

/* 1 when tfrt_trace3d_forward takes the in-place route for this scene, ray count and pass count
 * (tfrt_scene3d.in_place is honoured), 0 when it runs the per-pass launch sequence, < 0 on bad
 * arguments.  A caller that hands over ray_slot needs to know which of the two orders its ray
 * sets come back in. */
int tfrt_trace3d_in_place(const tfrt_scene3d* scene, int64_t n_rays, int32_t max_passes);

/* Work an in-place trace actually executed (measurement; no reference counterpart -- the reference
 * executes every pair, tfrt/engine.py:1103-1166): executed[0] = (ray, face) pairs that reached the
 * exact float64 test of tfrt/geometry.py:286-311, executed[1] = candidate faces tested as triangles
 * against a wavefront's bundle; both summed over the passes of the trace whose tape is in
 * `workspace`.  `counts` tells how many pairs the trace DECIDED (n_tests). */
int tfrt_trace3d_executed(int64_t n_rays, int64_t n_faces, int32_t max_passes, int32_t state_dtype,
                          const void* workspace, size_t workspace_bytes, int64_t* executed,
                          void* stream);

/* Reverse sweep over the tape left in `workspace` by tfrt_trace3d_forward with the same
 * arguments.  Replaces the ray-dependent part of tape.gradient in
 * SGD_Optimizer.process_gradient, tfrt/optimizer.py:216-220.
 *
 *   grad_finished/active/stopped/dead : 6 x capacity f64 blocks (row stride = the matching
 *              tfrt_ray_out.capacity of the forward call) or NULL
 *   grad_face_verts (M,9) f64, ACCUMULATED into (caller zeroes)
 *   grad_src_rays   6 x n_rays f64 or NULL
 */
int tfrt_trace3d_backward(const void* src_rays, int64_t src_stride, int64_t n_rays,
                          const tfrt_scene3d* scene, double new_ray_length,
                          double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                          const double* grad_finished, int64_t cap_finished,
                          const double* grad_active, int64_t cap_active,
                          const double* grad_stopped, int64_t cap_stopped,
                          const double* grad_dead, int64_t cap_dead, double* grad_face_verts,
                          double* grad_src_rays, const int32_t* counts, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Built-in image-forming error of the reference's optimisation scripts
 * (dev/hexalens.py:144-168: output = stack(finished[field_c]), goal = f(inherited source
 * fields), error = tf.math.squared_difference(output, goal)) evaluated on the device, with its
 * gradient seed, so that an optimiser step reads no ray count back and is a fixed launch
 * sequence.  Replaces the user error function + the first node of tape.gradient
 * (tfrt/optimizer.py:216-220) for error functions of that form.
 *
 *   finished_rays  6 x capacity ray block written by tfrt_trace3d_forward (state dtype)
 *   finished_id    (capacity) source-ray index per finished row (tfrt_ray_out.ray_id)
 *   counts         the device-side counters of that trace (TFRT_COUNTS_LEN(max_passes) int32):
 *                  the number of finished rows and the trace's test count are read from its tail
 *   fields[c]      row of the ray block (0..5 = x_start..z_end) compared with goal column c
 *   goal           f64 goal table, entry (c, source ray r) at goal[c * goal_stride +
 *                  r * goal_ray_stride]: one contiguous column per field (goal_ray_stride 1) or
 *                  one row per ray (goal_stride 1, goal_ray_stride n_fields)
 *   grad_finished  6 x capacity f64: rows fields[c] receive 2 * (output - goal) for the
 *                  finished rows; other rows and rows beyond n_finished are left untouched
 *                  (zero them once: tfrt_trace3d_backward reads finished rows only)
 *   error_out      3 f64: {sum of the error terms, number of terms, sum / max(terms, 1)}; the
 *                  sum is formed in a fixed order (bit-identical from run to run)
 *   zero_buffer    optional: zero_count f64 cleared by the same launch (the (M,9) face-gradient
 *                  block the reverse sweep accumulates into), or NULL / 0
 *   tests_total    optional device int64: incremented by the trace's ray-face test count
 *   workspace      tfrt_goal_error3d_workspace_bytes(capacity) bytes of scratch
 */
size_t tfrt_goal_error3d_workspace_bytes(int64_t capacity);
int tfrt_goal_error3d(const void* finished_rays, int64_t capacity, const int32_t* finished_id,
                      int32_t state_dtype, const int32_t* counts, int32_t max_passes,
                      const int32_t* fields, int32_t n_fields, const double* goal,
                      int64_t goal_stride, int64_t goal_ray_stride, double* grad_finished,
                      double* error_out, double* zero_buffer, int64_t zero_count,
                      int64_t* tests_total, void* workspace, size_t workspace_bytes,
                      void* stream);

/* The same without the second stage of the sum: `pending` (host struct) is filled in and whoever
 * runs next finishes it -- tfrt_sgd_process_multi_finish in a spare workgroup of the parameter
 * update's launch, or tfrt_goal_finish (the launch tfrt_goal_error3d would have made). */
int tfrt_goal_error3d_deferred(const void* finished_rays, int64_t capacity,
                               const int32_t* finished_id, int32_t state_dtype,
                               const int32_t* counts, int32_t max_passes, const int32_t* fields,
                               int32_t n_fields, const double* goal, int64_t goal_stride,
                               int64_t goal_ray_stride, double* grad_finished, double* error_out,
                               double* zero_buffer, int64_t zero_count, int64_t* tests_total,
                               void* workspace, size_t workspace_bytes,
                               tfrt_goal_pending* pending, void* stream);
int tfrt_goal_finish(const tfrt_goal_pending* pending, void* stream);

/* tfrt_goal_error3d_deferred and tfrt_trace3d_backward in ONE launch, for a trace over coherent
 * rays (tfrt_scene3d.coherent_rays, not deterministic, max_passes <= 8; TFRT_E_UNSUPPORTED
 * otherwise -- call the two entry points instead): the lane that walks a finished ray's chain of
 * slots back forms that ray's residuals itself -- output row minus goal row, the output as stored
 * in `finished` (state dtype) -- seeds the walk with 2 (output - goal) and leaves the squared
 * residuals in per-wavefront partial sums (a fixed order: the error is bit-identical from run to
 * run).  No seed block is written and read back, one dependent launch less per step.
 *   finished        the finished-ray block tfrt_trace3d_forward wrote (rays + capacity; ids unused:
 *                   a chain knows its source ray)
 *   fields, goal    as for tfrt_goal_error3d; goal rows are indexed by the source ray in the
 *                   trace's order
 *   goal_workspace  tfrt_trace3d_backward_goal_workspace_bytes(n_rays) bytes (the partial sums)
 *   pending         filled in like tfrt_goal_error3d_deferred's: finish with
 *                   tfrt_sgd_process_multi_finish or tfrt_goal_finish
 *   grad_face_verts (M,9) f64, ACCUMULATED into (cleared beforehand: tfrt_scene3d.clear_buffer)
 * Everything else as for tfrt_trace3d_backward (grad_active / grad_stopped / grad_dead may be
 * given as well; the finished rows' gradient is the goal's). */
size_t tfrt_trace3d_backward_goal_workspace_bytes(int64_t n_rays);
int tfrt_trace3d_backward_goal(const void* src_rays, int64_t src_stride, int64_t n_rays,
                               const tfrt_scene3d* scene, double new_ray_length,
                               double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                               const tfrt_ray_out* finished, const int32_t* fields,
                               int32_t n_fields, const double* goal, int64_t goal_stride,
                               int64_t goal_ray_stride, double* error_out, int64_t* tests_total,
                               void* goal_workspace, size_t goal_workspace_bytes,
                               tfrt_goal_pending* pending, const double* grad_active,
                               int64_t cap_active, const double* grad_stopped,
                               int64_t cap_stopped, const double* grad_dead, int64_t cap_dead,
                               double* grad_face_verts, double* grad_src_rays,
                               const int32_t* counts, void* workspace, size_t workspace_bytes,
                               void* stream);

/* Benchmark instrumentation (the only global state in the library; not used by the product
 * path).  While enabled, the launches of the hot kernels made by tfrt_trace3d_forward /
 * tfrt_trace3d_backward / tfrt_intersect3d are bracketed by HIP event pairs recorded on the
 * launch stream, one record per pass and kind:
 *   TFRT_PROF_INTERSECT   the pass's intersect launch(es): k_intersect_beam + k_intersect_group,
 *                         or k_intersect_group, or k_intersect3d
 *   TFRT_PROF_REACT       k_react3d
 *   TFRT_PROF_BACKWARD    k_backward3d (one per pass, last pass first)
 *   TFRT_PROF_ACCUMULATE  k_face_accumulate (one per reverse sweep)
 * tfrt_profile_read_kind synchronises the events of one kind and writes the elapsed
 * milliseconds, in launch order, into ms[0..max_records); it returns the record count (or a
 * negative error).  tfrt_profile_read = the intersect records.  tfrt_profile_enable(0|1) also
 * clears the records. */
#define TFRT_PROF_INTERSECT 0
#define TFRT_PROF_REACT 1
#define TFRT_PROF_BACKWARD 2
#define TFRT_PROF_ACCUMULATE 3
int tfrt_profile_enable(int enable);
int tfrt_profile_read(float* ms, int32_t max_records);
int tfrt_profile_read_kind(int32_t kind, float* ms, int32_t max_records);

/* ------------------------------------------------------------------------------------------
 * Seam-level single kernels (same math, no pass loop).
 */

/* OpticalSystem3D._intersection, tfrt/engine.py:1103-1166: nearest valid triangle per ray.
 * Outputs are (n_rays) arrays: x,y,z,ray_u,trig_u,trig_v f64; valid u8; gather_trig i32
 * (0 where invalid, like tf.argmin over an all-sentinel column). */
size_t tfrt_intersect3d_workspace_bytes(int64_t n_rays, int64_t n_faces);
int tfrt_intersect3d(const void* rays, int64_t stride, int64_t n_rays, int32_t state_dtype,
                     const double* face_verts, int64_t n_faces, double intersect_epsilion,
                     double size_epsilion, double ray_start_epsilion, double* x, double* y,
                     double* z, uint8_t* valid, double* ray_u, double* trig_u, double* trig_v,
                     int32_t* gather_trig, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * The public pairwise functions of tfrt/geometry.py (dense outputs, forward only).
 *
 * Outputs are row-major (n_rows, n_cols) grids, f64 (valid: u8).  Every operand array of a
 * set is read as p[col * col_stride + row * row_stride]:
 *   - tf.meshgrid forms (line_intersect, geometry.py:27-78; line_triangle_intersect, :191-251;
 *     line_circle_intersect, :338-402): first set (n_cols values) strides (1, 0), second set
 *     (n_rows values) strides (0, 1) -- the meshgrid copies of the reference are never made;
 *   - element-wise "raw" forms (raw_line_intersect, :96-167; raw_line_triangle_intersect,
 *     :275-320; raw_line_circle_intersect, :420-547): n_rows = 1, both sets strides (1, 0).
 * Same operation order and "safe value" masking as the reference: invalid elements hold the
 * reference's placeholder results (u = v = 1, ...), never NaN from a zero denominator.
 */
/* first = {x1s,y1s,x1e,y1e}, second = {x2s,y2s,x2e,y2e}; u: parameter on the first line, v: on
 * the second; valid = |denominator| >= epsilion. */
int tfrt_line_intersect(int64_t n_cols, int64_t n_rows, const double* const first[4],
                        int64_t first_col_stride, int64_t first_row_stride,
                        const double* const second[4], int64_t second_col_stride,
                        int64_t second_row_stride, double epsilion, double* x, double* y,
                        uint8_t* valid, double* u, double* v, void* stream);

/* rays = {rx1,ry1,rz1,rx2,ry2,rz2}, triangles = {xp,yp,zp,x1,y1,z1,x2,y2,z2} (pivot, first,
 * second vertex).  No range tests on ray_u / trig_u / trig_v (engine.py:1138-1141 does those). */
int tfrt_line_triangle_intersect(int64_t n_cols, int64_t n_rows, const double* const rays[6],
                                 int64_t ray_col_stride, int64_t ray_row_stride,
                                 const double* const triangles[9], int64_t tri_col_stride,
                                 int64_t tri_row_stride, double epsilion, double* x, double* y,
                                 double* z, uint8_t* valid, double* ray_u, double* trig_u,
                                 double* trig_v, void* stream);

/* lines = {xs,ys,xe,ye}, circles = {xc,yc,r}; plus / minus = {x,y,u,v} of the two roots of the
 * quadratic (v = angle of the hit on the circle), *_valid = root exists. */
int tfrt_line_circle_intersect(int64_t n_cols, int64_t n_rows, const double* const lines[4],
                               int64_t line_col_stride, int64_t line_row_stride,
                               const double* const circles[3], int64_t circle_col_stride,
                               int64_t circle_row_stride, double epsilion, double* const plus[4],
                               uint8_t* plus_valid, double* const minus[4], uint8_t* minus_valid,
                               void* stream);

/* geometry.snells_law_3D, tfrt/geometry.py:671-753.  All arrays (n) f64; norm is (n,3).
 * Writes the new ray (start = old end). */
int tfrt_snell3d(int64_t n, const double* x_start, const double* y_start, const double* z_start,
                 const double* x_end, const double* y_end, const double* z_end,
                 const double* norm, const double* n_in, const double* n_out,
                 double new_ray_length, double* out6 /* 6 x n */, void* stream);

/* geometry.snells_law_2D, tfrt/geometry.py:565-653.  out4 = 4 x n (xs,ys,xe,ye). */
int tfrt_snell2d(int64_t n, const double* x_start, const double* y_start, const double* x_end,
                 const double* y_end, const double* norm, const double* n_in,
                 const double* n_out, double new_ray_length, double* out4, void* stream);

/* Arithmetic self-test (no reference counterpart): out[i] = a[i] / b[i], sqrt(a[i]),
 * 1.0 / sqrt(a[i]) or a[i] + b[i] * b[i] (product and sum rounded separately), evaluated on the
 * device with the library's own compile flags.  The decisions of the trace (valid masks, nearest
 * hit, ties) equal the reference's eager float64 TensorFlow ops only if these are correctly
 * rounded; tests compare them with the host's IEEE results bit for bit. */
#define TFRT_SELFTEST_DIV 0
#define TFRT_SELFTEST_SQRT 1
#define TFRT_SELFTEST_RSQRT 2
#define TFRT_SELFTEST_MULADD 3
/* the reverse sweep's own reciprocal and reciprocal square root (hardware estimate + two Newton
 * steps, csrc/trace_math.h adj_rcp / adj_rsqrt; `b` unused): not correctly rounded by design --
 * the test bounds their error against the host's IEEE results */
#define TFRT_SELFTEST_ADJ_RCP 4
#define TFRT_SELFTEST_ADJ_RSQRT 5
int tfrt_selftest_f64(int op, int64_t n, const double* a, const double* b, double* out,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * 2-D: segments + arcs (OpticalSystem2D, tfrt/engine.py:254-866).
 */
typedef struct tfrt_scene2d {
  const double* seg;         /* (Ms,4) f64: x_start,y_start,x_end,y_end */
  const int32_t* seg_cat;    /* (Ms) */
  const int32_t* seg_mat_in; /* (Ms) or NULL */
  const int32_t* seg_mat_out;
  const double* seg_n_in;    /* (Ms) or NULL ("value" mode) */
  const double* seg_n_out;
  int64_t n_segments;
  const double* arc;         /* (Ma,5) f64: x_center,y_center,angle_start,angle_end,radius */
  const int32_t* arc_cat;
  const int32_t* arc_mat_in;
  const int32_t* arc_mat_out;
  const double* arc_n_in;
  const double* arc_n_out;
  int64_t n_arcs;
  const double* n_table;
  int64_t n_table_stride;
  int32_t n_materials;
  double intersect_epsilion, size_epsilion, ray_start_epsilion;
  /* Reverse sweep only.  0 (default, the reference): the gradient of a totally reflected ray
   * is NaN -- geometry.py:640-646 evaluates asin(theta2) with |theta2| > 1 in the unselected
   * branch of tf.where, and 0 * (d asin) = NaN flows into every boundary entry the ray touched
   * up to the reflecting one; tfrt_sgd_process zeroes such entries (optimizer.py:226-229).
   * 1: the reflect branch's own finite gradient (new_angle = norm + theta1 + pi) instead. */
  int32_t finite_tir_gradient;
} tfrt_scene2d;

/* OpticalSystem2D._segment_intersection, tfrt/engine.py:688-749 (rays: 4 x stride block). */
int tfrt_segment_intersection(const void* rays, int64_t stride, int64_t n_rays,
                              int32_t state_dtype, const double* seg, int64_t n_segments,
                              double intersect_epsilion, double size_epsilion,
                              double ray_start_epsilion, double* x, double* y, uint8_t* valid,
                              double* ray_u, double* seg_u, int32_t* gather_segment,
                              void* stream);

/* OpticalSystem2D._arc_intersection, tfrt/engine.py:768-866. */
int tfrt_arc_intersection(const void* rays, int64_t stride, int64_t n_rays, int32_t state_dtype,
                          const double* arc, int64_t n_arcs, double intersect_epsilion,
                          double size_epsilion, double ray_start_epsilion, double* x, double* y,
                          uint8_t* valid, double* ray_u, double* arc_u, int32_t* gather_arc,
                          void* stream);

size_t tfrt_trace2d_workspace_bytes(int64_t n_rays, int64_t n_segments, int64_t n_arcs,
                                    int32_t max_passes, int32_t state_dtype);

/* OpticalEngine.ray_trace for dimension 2 (process_projection_2D, tfrt/engine.py:1544-1986,
 * + snells_law_2D).  In a mixed segment+arc system every pass emits, per class, the
 * segment-hit rays first and then the arc-hit rays (engine.py:1955-1981); the face index
 * written is the merged segment index, or n_segments + merged arc index. */
int tfrt_trace2d_forward(const void* src_rays, int64_t src_stride, int64_t n_rays,
                         const tfrt_scene2d* scene, double new_ray_length,
                         double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                         uint32_t flags, tfrt_ray_out* finished, tfrt_ray_out* active,
                         tfrt_ray_out* stopped, tfrt_ray_out* dead, void* unfinished,
                         int32_t* unfinished_id, int32_t* counts, void* workspace,
                         size_t workspace_bytes, void* stream);

/* Reverse sweep of tfrt_trace2d_forward (the tape part of tfrt/optimizer.py:216-220 for a
 * 2-D system, e.g. dev/optimize_single_arc.py where the variable is an arc's centre/radius).
 *   grad_seg (Ms,4) f64 and grad_arc (Ma,5) f64 are ACCUMULATED into (caller zeroes); the
 *   angle_start/angle_end columns of grad_arc stay zero (they only feed comparisons).
 *   grad_src_rays 4 x n_rays f64 or NULL. */
int tfrt_trace2d_backward(const void* src_rays, int64_t src_stride, int64_t n_rays,
                          const tfrt_scene2d* scene, double new_ray_length,
                          double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                          const double* grad_finished, int64_t cap_finished,
                          const double* grad_active, int64_t cap_active,
                          const double* grad_stopped, int64_t cap_stopped,
                          const double* grad_dead, int64_t cap_dead, double* grad_seg,
                          double* grad_arc, double* grad_src_rays, const int32_t* counts,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ray order.  The reference's ray sets are ORDERED: every class lists, pass after pass, its rays
 * in the order of the source set (OpticalEngine.ray_trace / single_pass own both: tfrt/engine.py:
 * 2311-2330, the per-pass boolean_mask of :2069-2111, the ray-set properties :1379-1403).  The
 * trace is fastest over rays in a COHERENT order (tfrt_scene3d.coherent_rays).  These four entry
 * points give a caller both: order the source, trace the permuted block, put every class back.
 * Nothing synchronises and every launch has data-independent arguments: a source whose rays are
 * re-drawn every optimiser step (dev/hexalens.py:36-48 builds its source from RandomUniformCircle,
 * tfrt/distributions.py:1590-1592) is ordered inside the step's launch graph.
 *
 * tfrt_ray_order: index[j] = the ray that comes j-th along a Hilbert curve through the points where
 * the rays' lines pass the middle of the scene (foot of the perpendicular from the mean centroid of
 * 64 sampled faces -- or, face_verts == NULL, from the mean end point of 256 sampled rays -- in the
 * plane perpendicular to `axis`, 3 doubles on the HOST, or NULL: the mean direction of the sampled
 * rays; rays without a common direction, |mean| <= 1/2: octahedral map of the directions).  The
 * key has 2 b bits, b = ceil(log2(n) / 2) in [4, 13]; ties keep the source order (a stable
 * radix sort), so index == argsort(keys, stable).  keys_out (n_rays u32, natural order) or NULL. */
size_t tfrt_ray_order_workspace_bytes(int64_t n_rays);
int tfrt_ray_order(const void* rays, int64_t stride, int64_t n_rays, int32_t state_dtype,
                   const double* face_verts, int64_t n_faces, const double* axis, int32_t* index,
                   uint32_t* keys_out, void* workspace, size_t workspace_bytes, void* stream);

/* dst[:, j] = src[:, index[j]] for a 6-row ray block (through 8-element records: one random
 * access per ray instead of six).  src and dst must not overlap. */
size_t tfrt_permute_rays_workspace_bytes(int64_t n_rays, int32_t state_dtype);
int tfrt_permute_rays(const void* src_rays, int64_t src_stride, int64_t n_rays,
                      int32_t state_dtype, const int32_t* index, void* dst_rays,
                      int64_t dst_stride, void* workspace, size_t workspace_bytes, void* stream);

/* dst[k * dst_stride + j] = src[k * src_stride + index[j]], k < n_rows, j < min(n, *n_valid):
 * per-ray rows of any element size (1, 2, 4 or 8 bytes) through an index -- the n(lambda) table
 * and goal rows of an ordered source, the rows of an output class on their way back.  n_valid: a
 * count on the device (entries of `index` beyond it are not read) or NULL. */
int tfrt_gather_rows(const void* src, int64_t src_stride, int32_t n_rows, int32_t elem_bytes,
                     const int32_t* index, int64_t n, const int32_t* n_valid, void* dst,
                     int64_t dst_stride, void* stream);

/* tfrt_scene3d.cluster_order made on the device: the permutation of the faces that makes every
 * aligned run of `leaf` (16: the trace kernels' clusters) and of `group` (128: their superclusters)
 * consecutive entries a compact patch -- recursive median split of the face centroids along the
 * longest axis of their bounding box, the left part a multiple of `group` (of `leaf` below that);
 * faces eight times larger than the median face go last.  No counterpart in the reference (it
 * tests every ray against every face, tfrt/engine.py:1103-1166); once per mesh topology.  No host
 * sync.  group must be a multiple of leaf. */
size_t tfrt_cluster_order_workspace_bytes(int64_t n_faces, int32_t leaf, int32_t group);
int tfrt_cluster_order(const double* face_verts, int64_t n_faces, int32_t leaf, int32_t group,
                       int32_t* order, void* workspace, size_t workspace_bytes, void* stream);

/* One output class of a trace over permuted rays (source ray j of the trace = original ray
 * index[j]) back in the reference's order: inside every pass by original ray index.
 *   ray_id      the class's tfrt_ray_out.ray_id (ids in the trace's numbering), n_rows rows at most
 *   seg_n / seg_base / seg_stride / n_segments   rows and first row of the class in pass p at
 *               seg_n[p * seg_stride] / seg_base[...]: counts + TFRT_CLS_x and counts + 4 +
 *               TFRT_CLS_x with stride TFRT_COUNTS_PER_PASS, max_passes segments (<= 1024).
 *               Both NULL: one segment of n_rows rows (the unfinished set).
 *   total_rows  device count of the class's rows (counts + 8 max_passes + TFRT_CLS_x) or NULL: n_rows
 *   index       the permutation the trace ran over, or NULL (identity: the rows only get ranked)
 *   inv         out, row j of the restored class = row inv[j] of the trace's output   (or NULL)
 *   dest_of     out, the inverse: row r of the output goes to row dest_of[r]          (or NULL)
 *   ray_id_out  out, original ray index of restored row j                             (or NULL)
 * Rows of the restored class: gather with tfrt_gather_rows(.., inv, ..); gradients w.r.t. restored
 * rows on their way into tfrt_trace3d_backward: gather with dest_of. */
size_t tfrt_restore_order_workspace_bytes(int64_t n_src, int32_t n_segments);
int tfrt_restore_order(const int32_t* ray_id, int64_t n_rows, const int32_t* seg_n,
                       const int32_t* seg_base, int32_t seg_stride, int32_t n_segments,
                       const int32_t* total_rows, const int32_t* index, int64_t n_src,
                       int32_t* inv, int32_t* dest_of, int32_t* ray_id_out, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Source rays made on the device, in place.  Replaces, for sources that re-draw their rays at
 * every update (dev/hexalens.py:36-48; SGD_Optimizer.single_step calls optical_system.update()
 * first, tfrt/optimizer.py:217): the Random* distributions' _update (tfrt/distributions.py:
 * 1375-1393 square, 1586-1598 circle, 1751-1775 / 1814-1850 spherical caps: tf.random.uniform
 * pushed through the distribution's formula), BasePointTransformation (:2014-2120) and the
 * assembly of the rays by AperatureSource / PointSource / AngularSource._update
 * (tfrt/sources.py:464-1095, undense sources: sample i of every input makes ray i).
 *
 * Random numbers come from a counter-based generator (Philox4x32-10): sample i of a distribution
 * at epoch e is a pure function of (seed, stream, e, i), so rays and points can be written into
 * the caller's persistent buffers by one launch, in any order (through `index`), any number of
 * times, and only the epoch counters -- on the device, advanced by tfrt_epoch_advance -- change
 * from one optimiser step to the next.  The stream of numbers differs from TensorFlow's (whose
 * generator the reference leaves unseeded): parity here is the distribution, not the sequence. */
#define TFRT_PTS_TABLE 0          /* row i of `table` (a static distribution, already transformed) */
#define TFRT_PTS_CIRCLE 1         /* r = sqrt(u0), theta = theta_mod(2 pi u1): (0, R r cos, R r sin) */
#define TFRT_PTS_SQUARE 2         /* (0, -xs + 2 xs u0, -ys + 2 ys u1) */
#define TFRT_PTS_SPHERE_UNIFORM 3 /* phi = acos(lo + (1 - lo) u0), theta = theta_mod(pi (1 + sqrt 5) u1) */
#define TFRT_PTS_SPHERE_LAMBERT 4 /* phi = acos(sqrt(lo + (1 - lo) u0)), lo = cos^2(angular_size) */
typedef struct tfrt_points_program {
  int32_t kind;          /* TFRT_PTS_* */
  int32_t stream;        /* distinguishes the distributions that share a seed */
  int64_t count;         /* samples of the distribution */
  const double* table;   /* TFRT_PTS_TABLE: (count, 3) f64 */
  /* circle: {radius, theta_start, theta_end, -}; square: {x_size, -, -, y_size} (centre to
   * edge); spheres: {radius, theta_start, theta_end, lo}, lo = cos(angular_size) [uniform] or its
   * square [Lambertian] */
  double p[4];
  double scale[3], quat[4], shift[3]; /* BasePointTransformation: scale, unit quaternion (w, x, y, z), translation */
  int32_t has_scale, has_quat, has_shift;
  int32_t reserved0;
  uint64_t seed;
  const int64_t* epoch;  /* device counter (one int64): the number of updates so far */
} tfrt_points_program;

#define TFRT_SRC_APERTURE 0 /* start = a[i], end = b[i]                                      sources.py:918-1095 */
#define TFRT_SRC_POINT 1    /* start = center, end = center + L rot(b[i])                    sources.py:464-675 */
#define TFRT_SRC_ANGULAR 2  /* start = center + rot(a[i]), end = start + L rot(b[i])         sources.py:678-915 */
typedef struct tfrt_source3d_program {
  int32_t kind;          /* TFRT_SRC_* */
  int32_t swap;          /* start and end exchanged (start_on_center / start_on_base false) */
  tfrt_points_program a; /* start points / base points (unused by TFRT_SRC_POINT) */
  tfrt_points_program b; /* end points / direction vectors */
  double center[3];
  double quat[4];        /* central angle as a unit quaternion */
  int32_t has_quat, reserved0;
  double ray_length;
  int64_t n_rays;        /* every input has 1 or n_rays samples */
} tfrt_source3d_program;

/* tfrt_ray_order for rays first .. first + n_rays of a source program, without the rays ever being
 * written in source order: index[j] = the j-th ray (counted from `first`) in the coherent order.
 * Workspace: tfrt_ray_order_workspace_bytes(n_rays).  tfrt_source3d_generate(program, index, first,
 * ...) then makes the ordered block. */
int tfrt_source3d_order(const tfrt_source3d_program* program, int64_t first, int64_t n_rays,
                        const double* face_verts, int64_t n_faces, const double* axis,
                        int32_t* index, uint32_t* keys_out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* The same order up to ties, made faster, for a source that is re-drawn before every optimiser step
 * (distributions.py:1586-1598: the order is made again every step): most significant digit first --
 * ONE stable scatter by the key's high digit, then every bucket of it is sorted by the low digit in
 * LDS (one workgroup per bucket) -- five launches instead of the two-pass sort's eight.  Rays with
 * the SAME key (the same cell of the key grid: about one ray per cell) land in an order that may vary
 * from run to run.  A trace's results do not depend on the order of its rays (tfrt_scene3d.ray_slot
 * numbers them as the caller does); only the rounding of sums over rays does, as with any atomic
 * accumulation.  Not for deterministic runs.  Same arguments and workspace as tfrt_source3d_order. */
int tfrt_source3d_order_cells(const tfrt_source3d_program* program, int64_t first, int64_t n_rays,
                              const double* face_verts, int64_t n_faces, const double* axis,
                              int32_t* index, uint32_t* keys_out, void* workspace,
                              size_t workspace_bytes, void* stream);

/* epochs[k][0] += 1 for k < n (n <= 8 distinct device counters, host array of pointers): one
 * launch for all the distributions a source draws from. */
int tfrt_epoch_advance(int64_t* const* epochs, int32_t n, void* stream);

/* Samples first + index[j] (NULL: first + j), j < n, of one distribution at its current epoch: the points after
 * the transformation as (n, 3) f64 rows (point_columns 3) or the distribution's own plane as
 * (n, 2) rows (point_columns 2: components y, z), and the two numbers its rank properties are made
 * of (circle: r in [0, 1] and theta; spheres: phi and theta; square: the point before the
 * transformation) -- each output may be NULL. */
int tfrt_points_generate(const tfrt_points_program* program, const int32_t* index,
                         int64_t first, int64_t n, double* points, int32_t point_columns, double* aux0, double* aux1,
                         void* stream);

/* Rays first + index[j] (NULL: first + j), j < n (`first`: a rank's shard of the source), of the
 * source at the current epochs of its distributions:
 * `rays` a 6 x stride block of the state dtype and / or `fields` a 6 x field_stride f64 block
 * (x_start ... z_end: the source's field columns); either may be NULL. */
int tfrt_source3d_generate(const tfrt_source3d_program* program, const int32_t* index,
                           int64_t first, int64_t n, int32_t state_dtype, void* rays, int64_t stride, double* fields,
                           int64_t field_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TFRT_HIP_H */
