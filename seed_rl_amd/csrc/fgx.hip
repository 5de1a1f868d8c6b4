// Stack-entry 16 -> 32 convolution of ImpalaDeep on the bf16 matrix pipe (fgx.h) -- its own translation unit.
#include "fgx.h"
