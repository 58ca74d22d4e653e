// placeholder, replaced below in this round
#include "tfrt_common.h"
extern "C" {
int tfrt_segment_intersection(const void*, int64_t, int64_t, int32_t, const double*, int64_t, double, double, double, double*, double*, uint8_t*, double*, double*, int32_t*, void*) { return TFRT_E_UNSUPPORTED; }
int tfrt_arc_intersection(const void*, int64_t, int64_t, int32_t, const double*, int64_t, double, double, double, double*, double*, uint8_t*, double*, double*, int32_t*, void*) { return TFRT_E_UNSUPPORTED; }
size_t tfrt_trace2d_workspace_bytes(int64_t, int64_t, int64_t, int32_t, int32_t) { return 0; }
int tfrt_trace2d_forward(const void*, int64_t, int64_t, const tfrt_scene2d*, double, double, int32_t, int32_t, uint32_t, tfrt_ray_out*, tfrt_ray_out*, tfrt_ray_out*, tfrt_ray_out*, void*, int32_t*, int32_t*, void*, size_t, void*) { return TFRT_E_UNSUPPORTED; }
int tfrt_trace2d_backward(const void*, int64_t, int64_t, const tfrt_scene2d*, double, double, int32_t, int32_t, const double*, int64_t, const double*, int64_t, const double*, int64_t, const double*, int64_t, double*, double*, double*, const int32_t*, void*, size_t, void*) { return TFRT_E_UNSUPPORTED; }
}
