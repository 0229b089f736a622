// conv_thin_impl.h instantiated for mode THIN_DP
#define THIN_MODE THIN_DP
#include "conv_thin_impl.h"
THIN_ENTRY(thin_launch_dp)
