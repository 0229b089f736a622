// conv_thin_impl.h instantiated for mode THIN_DAS
#define THIN_MODE THIN_DAS
#include "conv_thin_impl.h"
THIN_ENTRY(thin_launch_das)
