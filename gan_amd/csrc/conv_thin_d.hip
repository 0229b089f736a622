// conv_thin_impl.h instantiated for mode THIN_D
#define THIN_MODE THIN_D
#include "conv_thin_impl.h"
THIN_ENTRY(thin_launch_d)
