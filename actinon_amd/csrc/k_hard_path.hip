/* k_hard_path< ... >: see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"

#define ACN_LHP_( C, L, P ) hipLaunchKernelGGL( ( k_hard_path< C, L, P > ), dim3( q.grid ), dim3( 256 ), lds_bytes, stream, \
    ACN_SCENE_ARGS_OF( s ), ( const HardPath* )q.hard_path, q.hard_cap, q.fetch_hard, q.children, q.child_cap, q.counts, accum, counters )
void acn_launch_hard_path( KernelFlags f, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                           unsigned long long* accum, unsigned long long* counters )
{
    if( f.count && f.prune ) { if( f.lds_nodes ) ACN_LHP_( true, true, true );  else ACN_LHP_( true, false, true ); }
    else if( f.count ) { if( f.lds_nodes ) ACN_LHP_( true, true, false );  else ACN_LHP_( true, false, false ); }
    else if( f.prune ) { if( f.lds_nodes ) ACN_LHP_( false, true, true );  else ACN_LHP_( false, false, true ); }
    else               { if( f.lds_nodes ) ACN_LHP_( false, true, false ); else ACN_LHP_( false, false, false ); }
}
