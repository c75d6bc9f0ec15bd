/* k_trace_rays< ... > and k_shade_hits: see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"

#define ACN_WQ_( q ) ( q ).tasks, ( q ).idx[ 0 ], ( q ).idx[ 1 ], ( q ).idx[ 2 ], ( q ).idx[ 3 ], ( q ).counts, ( q ).task_cap, ( q ).rays_out, ( q ).ray_cap
#define ACN_LT_( P, C, L, R ) hipLaunchKernelGGL( ( k_trace_rays< P, C, L, R > ), dim3( ( n + 255 ) / 256 ), dim3( 256 ), lds_bytes, stream, \
    ACN_SCENE_ARGS_OF( s ), ACN_WQ_( q ), rays_in, pos_xy, first_pixel, base, n, accum, counters )
#define ACN_DT_( P ) do { \
    if( f.count )      { if( f.lds_nodes ) ACN_LT_( P, true, true, false );  else ACN_LT_( P, true, false, false ); } \
    else if( f.prune ) { if( f.lds_nodes ) ACN_LT_( P, false, true, true );  else ACN_LT_( P, false, false, true ); } \
    else               { if( f.lds_nodes ) ACN_LT_( P, false, true, false ); else ACN_LT_( P, false, false, false ); } } while( 0 )

void acn_launch_trace( bool primary, KernelFlags f, uint32_t n, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                       const WalkQueueArgs& q, const RayTask* rays_in, const double* pos_xy, size_t first_pixel, uint32_t base,
                       unsigned long long* accum, unsigned long long* counters )
{
    if( primary ) ACN_DT_( true ); else ACN_DT_( false );
}

void acn_launch_shade_hits( bool count, uint32_t n, hipStream_t stream, const SceneArgs& s, const WalkQueueArgs& q,
                            const HitRec* recs, unsigned long long* accum, unsigned long long* counters )
{
    if( count ) hipLaunchKernelGGL( ( k_shade_hits< true > ), dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, stream, ACN_SCENE_ARGS_OF( s ), ACN_WQ_( q ), recs, n, accum, counters );
    else        hipLaunchKernelGGL( ( k_shade_hits< false > ), dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, stream, ACN_SCENE_ARGS_OF( s ), ACN_WQ_( q ), recs, n, accum, counters );
}
