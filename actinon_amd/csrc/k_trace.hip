/* k_trace_rays< ... > and k_shade_hits: see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"

#define ACN_WQ_( q ) ( q ).tasks, ( q ).idx[ 0 ], ( q ).idx[ 1 ], ( q ).idx[ 2 ], ( q ).idx[ 3 ], ( q ).counts, ( q ).task_cap, ( q ).rays_out, ( q ).ray_cap
#define ACN_LT_( C, L, R ) hipLaunchKernelGGL( ( k_trace_rays< C, L, R > ), dim3( ( n + 255 ) / 256 ), dim3( 256 ), lds_bytes, stream, \
    ACN_SCENE_ARGS_OF( s ), ACN_WQ_( q ), rays_in, pos_xy, first_pixel, base, n, accum, counters )
void acn_launch_trace_count( KernelFlags f, uint32_t n, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                             const WalkQueueArgs& q, const RayTask* rays_in, const double* pos_xy, size_t first_pixel, uint32_t base,
                             unsigned long long* accum, unsigned long long* counters );   /* k_trace_count.hip */
#define ACN_DT_() do { \
    if( f.count )      acn_launch_trace_count( f, n, lds_bytes, stream, s, q, rays_in, pos_xy, first_pixel, base, accum, counters ); \
    else if( f.prune ) { if( f.lds_nodes ) ACN_LT_( false, true, true );  else ACN_LT_( false, false, true ); } \
    else               { if( f.lds_nodes ) ACN_LT_( false, true, false ); else ACN_LT_( false, false, false ); } } while( 0 )

void acn_launch_trace( bool primary, KernelFlags f, uint32_t n, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                       const WalkQueueArgs& q, const RayTask* rays_in, const double* pos_xy, size_t first_pixel, uint32_t base,
                       unsigned long long* accum, unsigned long long* counters )
{
    ( void )primary;   /* camera rays are requested with rays_in == nullptr */
    ACN_DT_();
}

void acn_launch_shade_hits( bool count, uint32_t n, hipStream_t stream, const SceneArgs& s, const WalkQueueArgs& q,
                            const HitRec* recs, unsigned long long* accum, unsigned long long* counters )
{
    if( count ) hipLaunchKernelGGL( ( k_shade_hits< true > ), dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, stream, ACN_SCENE_ARGS_OF( s ), ACN_WQ_( q ), recs, n, accum, counters );
    else        hipLaunchKernelGGL( ( k_shade_hits< false > ), dim3( ( n + 255 ) / 256 ), dim3( 256 ), 0, stream, ACN_SCENE_ARGS_OF( s ), ACN_WQ_( q ), recs, n, accum, counters );
}
