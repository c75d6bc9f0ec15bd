/* k_trace_chase< ... >: see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"

#define ACN_WQ_( q ) ( q ).tasks, ( q ).idx[ 0 ], ( q ).idx[ 1 ], ( q ).idx[ 2 ], ( q ).idx[ 3 ], ( q ).counts, ( q ).task_cap, ( q ).rays_out, ( q ).ray_cap
#define ACN_LC_( L, R ) hipLaunchKernelGGL( ( k_trace_chase< false, L, R > ), dim3( ( n + 255 ) / 256 ), dim3( 256 ), lds_bytes, stream, \
    ACN_SCENE_ARGS_OF( s ), ACN_WQ_( q ), rays_in, n, chase_buf, accum, counters )

size_t acn_chase_buffer_bytes( uint32_t max_rays ) { return ( size_t )( ( max_rays + 255 ) / 256 ) * 2 * ACN_CHASE_CAP * sizeof( RayTask ); }

void acn_launch_trace_chase( KernelFlags f, uint32_t n, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                             const WalkQueueArgs& q, const RayTask* rays_in, RayTask* chase_buf,
                             unsigned long long* accum, unsigned long long* counters )
{
    if( f.prune ) { if( f.lds_nodes ) ACN_LC_( true, true );  else ACN_LC_( false, true ); }
    else          { if( f.lds_nodes ) ACN_LC_( true, false ); else ACN_LC_( false, false ); }
}
