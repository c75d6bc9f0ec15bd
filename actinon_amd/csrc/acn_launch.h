/* acn_launch.h -- launch wrappers of the templated pipeline kernels, one translation unit per kernel family
 * (k_shade_*.hip, k_trace.hip, k_hard.hip) so that `make -j` compiles the families in parallel: the 50-odd kernel
 * instantiations in one file took 8.5 minutes, the families side by side take about 3.  The wrappers pick the
 * instantiation from runtime flags; the orchestration stays in actinon_hip.hip. */
#ifndef ACN_LAUNCH_H
#define ACN_LAUNCH_H

#include "acn_pipeline.h"

/* what every pipeline kernel receives first (ACN_SCENE_PARAMS) */
struct SceneArgs
{
    DevScene dev;
    const GNode* nodes;
    const GMat* mats;
    const int32_t* elems;
    const acn_texture* textures;
};

/* variant selection: instrumented kernels (count) never carry prune programs */
struct KernelFlags { bool count, leaf_lights, lds_nodes, prune; };

/* ACN_WALK_QUEUE_PARAMS */
struct WalkQueueArgs
{
    DTask* tasks; uint32_t* idx[ 4 ]; uint32_t* counts; uint32_t task_cap; RayTask* rays_out; uint32_t ray_cap;
};

void acn_launch_shade( int lanes_per_task, KernelFlags f, unsigned blocks, hipStream_t stream, const SceneArgs& s,
                       const DTask* tasks, const uint32_t* idx, uint32_t n_tasks, HitRec* children, uint32_t child_cap,
                       HardShadow* hard_shadow, HardPath* hard_path, uint32_t hard_cap, uint32_t* counts,
                       unsigned long long* accum, unsigned long long* counters );
void acn_launch_shade64( KernelFlags, unsigned, hipStream_t, const SceneArgs&, const DTask*, const uint32_t*, uint32_t, HitRec*, uint32_t, HardShadow*, HardPath*, uint32_t, uint32_t*, unsigned long long*, unsigned long long* );
void acn_launch_shade16( KernelFlags, unsigned, hipStream_t, const SceneArgs&, const DTask*, const uint32_t*, uint32_t, HitRec*, uint32_t, HardShadow*, HardPath*, uint32_t, uint32_t*, unsigned long long*, unsigned long long* );
void acn_launch_shade4( KernelFlags, unsigned, hipStream_t, const SceneArgs&, const DTask*, const uint32_t*, uint32_t, HitRec*, uint32_t, HardShadow*, HardPath*, uint32_t, uint32_t*, unsigned long long*, unsigned long long* );
void acn_launch_shade1( KernelFlags, unsigned, hipStream_t, const SceneArgs&, const DTask*, const uint32_t*, uint32_t, HitRec*, uint32_t, HardShadow*, HardPath*, uint32_t, uint32_t*, unsigned long long*, unsigned long long* );

void acn_launch_trace( bool primary, KernelFlags f, uint32_t n, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                       const WalkQueueArgs& q, const RayTask* rays_in, const double* pos_xy, size_t first_pixel, uint32_t base,
                       unsigned long long* accum, unsigned long long* counters );
/* the tail of the walk in one launch (uninstrumented kernels only); chase_buf holds acn_chase_buffer_bytes( max rays ) */
size_t acn_chase_buffer_bytes( uint32_t max_rays );
void acn_launch_trace_chase( KernelFlags f, uint32_t n, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                             const WalkQueueArgs& q, const RayTask* rays_in, RayTask* chase_buf,
                             unsigned long long* accum, unsigned long long* counters );
void acn_launch_shade_hits( bool count, uint32_t n, hipStream_t stream, const SceneArgs& s, const WalkQueueArgs& q,
                            const HitRec* recs, unsigned long long* accum, unsigned long long* counters );
void acn_launch_hard_shadow( KernelFlags f, uint32_t n, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                             const HardShadow* recs, uint32_t* counts, unsigned long long* accum, unsigned long long* counters );
void acn_launch_hard_path( KernelFlags f, uint32_t n, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                           const HardPath* recs, HitRec* children, uint32_t child_cap, uint32_t* counts,
                           unsigned long long* accum, unsigned long long* counters );

#define ACN_SCENE_ARGS_OF( s ) ( s ).dev, ( s ).nodes, ( s ).mats, ( s ).elems, ( s ).textures

/* body of acn_launch_shade<LPT>: shared by the four k_shade translation units */
#define ACN_DEFINE_LAUNCH_SHADE( NAME, LPT ) \
void NAME( KernelFlags f, unsigned blocks, hipStream_t stream, const SceneArgs& s, const DTask* tasks, const uint32_t* idx, \
           uint32_t n_tasks, HitRec* children, uint32_t child_cap, HardShadow* hard_shadow, HardPath* hard_path, \
           uint32_t hard_cap, uint32_t* counts, unsigned long long* accum, unsigned long long* counters ) \
{ \
    _Pragma( "clang diagnostic push" ) \
    /* the prune-program variants exist for the uninstrumented kernels only; count_work runs the plain ones */ \
    if( f.count )      { if( f.leaf_lights ) ACN_LS_( LPT, true, true, false );  else ACN_LS_( LPT, true, false, false ); } \
    else if( f.prune ) { if( f.leaf_lights ) ACN_LS_( LPT, false, true, true );  else ACN_LS_( LPT, false, false, true ); } \
    else               { if( f.leaf_lights ) ACN_LS_( LPT, false, true, false ); else ACN_LS_( LPT, false, false, false ); } \
    _Pragma( "clang diagnostic pop" ) \
}
#define ACN_LS_( LPT, C, L, P ) hipLaunchKernelGGL( ( k_shade< LPT, C, L, P > ), dim3( blocks ), dim3( 256 ), 0, stream, ACN_SCENE_ARGS_OF( s ), \
    tasks, idx, n_tasks, children, child_cap, hard_shadow, hard_path, hard_cap, counts, accum, counters )

#endif
