// conv_thin_impl.h instantiated for mode THIN_G
#define THIN_MODE THIN_G
#include "conv_thin_impl.h"
THIN_ENTRY(thin_launch_g)
