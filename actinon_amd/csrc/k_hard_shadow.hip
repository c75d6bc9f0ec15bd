/* k_hard_shadow< ... >: see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"

#define ACN_LHS_( C, L, P ) hipLaunchKernelGGL( ( k_hard_shadow< C, L, P > ), dim3( q.grid ), dim3( 256 ), lds_bytes, stream, \
    ACN_SCENE_ARGS_OF( s ), ( const HardShadow* )q.hard_shadow, q.hs_cap, q.fetch_hard, q.counts, accum, counters )
void acn_launch_hard_shadow( KernelFlags f, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                             unsigned long long* accum, unsigned long long* counters )
{
    if( f.count && f.prune ) { if( f.lds_nodes ) ACN_LHS_( true, true, true );  else ACN_LHS_( true, false, true ); }
    else if( f.count ) { if( f.lds_nodes ) ACN_LHS_( true, true, false );  else ACN_LHS_( true, false, false ); }
    else if( f.prune ) { if( f.lds_nodes ) ACN_LHS_( false, true, true );  else ACN_LHS_( false, false, true ); }
    else               { if( f.lds_nodes ) ACN_LHS_( false, true, false ); else ACN_LHS_( false, false, false ); }
}
