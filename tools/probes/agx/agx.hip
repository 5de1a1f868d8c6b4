// Second convolution of the shallow Atari torso, forward + data gradient at training batch sizes (agx.h).
#include "agx.h"
