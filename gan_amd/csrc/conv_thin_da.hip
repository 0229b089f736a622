// conv_thin_impl.h instantiated for mode THIN_DA
#define THIN_MODE THIN_DA
#include "conv_thin_impl.h"
THIN_ENTRY(thin_launch_da)
