// conv_thin_impl.h instantiated for mode THIN_DPR
#define THIN_MODE THIN_DPR
#include "conv_thin_impl.h"
THIN_ENTRY(thin_launch_dpr)
