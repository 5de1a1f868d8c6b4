// Host interface of the shallow Atari torso's second convolution on cgx.h's machine (agx.h, compiled in agx.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/seedhip.h"

namespace seedhip {
namespace agx {

constexpr int kMinImages = 2048;            // training-sized batches; inference batches stay on wfx.h / wdx.h
// true: 4 x 4 stride 2 'valid' convolution 16 -> 32 on 20 x 20 maps, dense layouts, >= kMinImages images
bool plan(const seedhip_conv_geom* g);
// relu_bits (may be null): the ReLU mask of the output as bytes [pixel][8] (seedhip_conv2d_fwd_bits)
int launch_fwd(const seedhip_conv_geom* g, const float* X, const float* W, const float* bias, float* Y, int out_relu,
               unsigned char* relu_bits, hipStream_t s);
// relu_mask (fp32) or relu_bits (bytes [pixel][4]) of the layer's input, or neither
int launch_dgrad(const seedhip_conv_geom* g, const float* dY, const float* W, float* dX, const float* relu_mask,
                 const unsigned char* relu_bits, hipStream_t s);

}  // namespace agx
}  // namespace seedhip
