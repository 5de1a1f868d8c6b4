// Host interface of the bf16x6 kernels of ImpalaDeep's 16 -> 32 stack-entry convolution (fgx.h, compiled in fgx.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/seedhip.h"

namespace seedhip {
namespace fgx {

// true: 3 x 3 'same' convolution 16 -> 32 on 36 x 48 maps, dense layouts, enough images to fill the chip
bool plan(const seedhip_conv_geom* g);
int launch_fwd(const seedhip_conv_geom* g, const float* X, const float* W, const float* bias, float* Y, hipStream_t s);
int launch_dgrad(const seedhip_conv_geom* g, const float* dY, const float* W, float* dX, hipStream_t s);
// the same behind MaxPool2D(3, 2, 'same'): dY rebuilt in the loader from the pooled map's gradient and the argmax bytes
// and written to DA as well
int launch_dgrad_pool(const seedhip_conv_geom* g, const float* dpooled, const unsigned char* argmax, const float* W, float* dX,
                      float* DA, hipStream_t s);

}  // namespace fgx
}  // namespace seedhip
