/* k_shade< 4, ... >: see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"
ACN_DEFINE_LAUNCH_SHADE( acn_launch_shade4, 4, 2 )
