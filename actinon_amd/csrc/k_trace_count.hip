/* k_trace_rays< true, ... > (instrumented): see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"

#define ACN_WQ_( q ) ( q ).tasks, ( q ).idx[ 0 ], ( q ).idx[ 1 ], ( q ).idx[ 2 ], ( q ).idx[ 3 ], ( q ).counts, ( q ).task_cap, ( q ).rays_out, ( q ).ray_cap
#define ACN_LT_( C, L, R ) hipLaunchKernelGGL( ( k_trace_rays< C, L, R > ), dim3( ( n + 255 ) / 256 ), dim3( 256 ), lds_bytes, stream, \
    ACN_SCENE_ARGS_OF( s ), ACN_WQ_( q ), rays_in, pos_xy, first_pixel, base, n, accum, counters )

void acn_launch_trace_count( KernelFlags f, uint32_t n, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                             const WalkQueueArgs& q, const RayTask* rays_in, const double* pos_xy, size_t first_pixel, uint32_t base,
                             unsigned long long* accum, unsigned long long* counters )
{
    if( f.lds_nodes ) ACN_LT_( true, true, false ); else ACN_LT_( true, false, false );
}
