/* k_hard_path< ... >: see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"

#define ACN_LHP_( C, L, P ) hipLaunchKernelGGL( ( k_hard_path< C, L, P > ), dim3( ( n + 255 ) / 256 ), dim3( 256 ), lds_bytes, stream, \
    ACN_SCENE_ARGS_OF( s ), recs, n, children, child_cap, counts, accum, counters )
void acn_launch_hard_path( KernelFlags f, uint32_t n, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                           const HardPath* recs, HitRec* children, uint32_t child_cap, uint32_t* counts,
                           unsigned long long* accum, unsigned long long* counters )
{
    if( f.count )      { if( f.lds_nodes ) ACN_LHP_( true, true, false );  else ACN_LHP_( true, false, false ); }
    else if( f.prune ) { if( f.lds_nodes ) ACN_LHP_( false, true, true );  else ACN_LHP_( false, false, true ); }
    else               { if( f.lds_nodes ) ACN_LHP_( false, true, false ); else ACN_LHP_( false, false, false ); }
}
