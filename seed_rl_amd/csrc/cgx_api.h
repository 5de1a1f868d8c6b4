// Host interface of the bf16x6 kernels of the DQN torso's third convolution (cgx.h, compiled in cgx.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/seedhip.h"

namespace seedhip {
namespace cgx {

// true: 3 x 3 'valid' convolution 64 -> 64 on 9 x 9 maps, dense layouts, a training-sized batch
bool plan(const seedhip_conv_geom* g);
// true: the torso's second convolution, 4 x 4 stride 2 'valid' 32 -> 64 on 20 x 20 maps (forward only)
bool plan_fwd2(const seedhip_conv_geom* g);
int launch_fwd(const seedhip_conv_geom* g, const float* X, const float* W, const float* bias, float* Y, int out_relu, hipStream_t s);
int launch_dgrad(const seedhip_conv_geom* g, const float* dY, const float* W, float* dX, const float* relu_mask, hipStream_t s);
// the second convolution's data gradient (cgx2.h: four stride-parity classes)
bool plan_dgrad2(const seedhip_conv_geom* g);
int launch_dgrad2(const seedhip_conv_geom* g, const float* dY, const float* W, float* dX, const float* relu_mask, hipStream_t s);

}  // namespace cgx
}  // namespace seedhip
