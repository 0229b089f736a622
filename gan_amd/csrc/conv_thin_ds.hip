// conv_thin_impl.h instantiated for mode THIN_DS
#define THIN_MODE THIN_DS
#include "conv_thin_impl.h"
THIN_ENTRY(thin_launch_ds)
