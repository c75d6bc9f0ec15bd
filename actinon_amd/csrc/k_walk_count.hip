/* k_walk< true, ... > (instrumented) and k_shade_hits: see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"

void acn_launch_walk_count_prune( KernelFlags f, uint32_t pass, bool last, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                                  const double* pos_xy, size_t first_pixel, uint32_t base, uint32_t n_cam, TileOrder order,
                                  unsigned long long* accum, unsigned long long* counters );   /* k_walk_count_prune.hip */

void acn_launch_walk_count( KernelFlags f, uint32_t pass, bool last, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                            const double* pos_xy, size_t first_pixel, uint32_t base, uint32_t n_cam, TileOrder order,
                            unsigned long long* accum, unsigned long long* counters )
{
    if( f.prune )          acn_launch_walk_count_prune( f, pass, last, q, lds_bytes, stream, s, pos_xy, first_pixel, base, n_cam, order, accum, counters );
    else if( f.lds_nodes ) ACN_LW_( true, true, false );
    else                   ACN_LW_( true, false, false );
}

#define ACN_LSH_( C ) hipLaunchKernelGGL( ( k_shade_hits< C > ), dim3( q.grid ), dim3( 256 ), 0, stream, ACN_SCENE_ARGS_OF( s ), ACN_TASKQ_ARGS_OF( q ), \
    ( const HitRec* )q.children, q.prev_children, q.child_cap, q.fetch_hard, q.rays[ 0 ], q.ray_cap, accum, counters )
void acn_launch_shade_hits( bool count, const LevelQ& q, hipStream_t stream, const SceneArgs& s,
                            unsigned long long* accum, unsigned long long* counters )
{
    if( count ) ACN_LSH_( true ); else ACN_LSH_( false );
}
