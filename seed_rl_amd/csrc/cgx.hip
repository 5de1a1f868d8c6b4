// Third convolution of the DQN torso on the bf16 matrix pipe (cgx.h) -- its own translation unit.
#include "cgx.h"
