/* k_shade< 64, ... >: see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"
ACN_DEFINE_LAUNCH_SHADE( acn_launch_shade64, 64, 0 )
