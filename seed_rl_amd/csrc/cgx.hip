// Second / third convolution of the DQN torso on the bf16 matrix pipe (cgx.h, cgx2.h) -- their own translation unit.
#include "cgx.h"
#include "cgx2.h"
