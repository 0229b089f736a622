// conv_thin_impl.h instantiated for mode THIN_DPS
#define THIN_MODE THIN_DPS
#include "conv_thin_impl.h"
THIN_ENTRY(thin_launch_dps)
