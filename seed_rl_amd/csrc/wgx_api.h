// Host interface of the bf16x6 weight-gradient kernels (wgx.h, compiled in wgx.hip) for conv.hip's dispatch.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/seedhip.h"

namespace seedhip {
namespace wgx {

// 0: geometry (or image count) not served; otherwise the index of the instantiated geometry
int plan(const seedhip_conv_geom* g);
// workgroups = partial slices [grid][kh * kw * cin * cout (+ cout)] the launch writes
int grid_for(int k, int n_img, int* per_wg_out = nullptr);
int launch(int k, const seedhip_conv_geom* g, const float* X, int in_relu, const float* dY, float* partial_w,
           float* partial_b, int* slices, hipStream_t s);

}  // namespace wgx
}  // namespace seedhip
