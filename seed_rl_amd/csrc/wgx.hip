// Weight gradients of the small convolutions on the bf16 matrix pipe (wgx.h) -- their own translation unit.
#include "wgx.h"
