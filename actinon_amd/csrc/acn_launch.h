/* acn_launch.h -- launch wrappers of the templated pipeline kernels, one translation unit per kernel family
 * (k_shade_*.hip, k_walk_*.hip, k_hard_*.hip) so that `make -j` compiles the families in parallel: the 50-odd kernel
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

/* variant selection.  The instrumented kernels (count) exist with and without the prune programs / in-line simple
 * compounds as well, so that a counted pass walks the traversal the timed pass walks */
struct KernelFlags { bool count, leaf_lights, lds_nodes, prune; };

/* the queues of one pipeline run (a handle's or a lane's workspace) as the kernels of path level L see them:
 * `counts` is the level's counter block, `prev_children` the QC_CHILDREN word of the level before */
struct LevelQ
{
    DTask* tasks; uint32_t* idx[ ACN_NCLASS ]; uint32_t task_cap;
    HitRec* children; uint32_t child_cap;
    HardShadow* hard_shadow; uint32_t hs_cap;               /* twice the other queues: it also takes the probes of the walk (probe_push) */
    HardPath* hard_path; uint32_t hard_cap;
    RayTask* rays[ 2 ]; uint32_t ray_cap;                   /* generation g of the walk waits in rays[ g & 1 ] */
    RayTask* stacks; uint32_t stack_cap;                    /* private ray stacks of the k_walk waves: grid * 4 of them */
    uint32_t stack_use;                                     /* slots of a stack every pass but the last uses (< stack_cap: tests) */
    uint32_t* counts;
    const uint32_t* prev_children;
    unsigned grid;                                          /* workgroups of the persistent kernels */
    unsigned shade_grid;                                    /* workgroups of k_shade */
    uint32_t fetch_walk, fetch_hard;                        /* input items a wave reserves per cursor atomic */
    uint32_t fetch_shade;                                   /* k_shade: steps of 64 / LPT tasks a wave reserves per atomic */
    uint32_t private_limit;                                 /* generations of at most this many rays are finished on private stacks */
    uint32_t shard_rank, shard_world;                       /* ACN_SHARD_SAMPLES at level 0: the rank's share of the sample loops; else 0, 1 */
    uint32_t emit_terms;                                    /* 0: k_walk drops its pixel terms (level 0 of a rank > 0 of such a call) */
};

/* pass `pass` of the specular walk of the level.  n_cam > 0 (pass 0 of level 0): the input are the camera rays of
 * positions [ base, base + n_cam ); else generation `pass` of the level's ray queues.  last: the input is finished on the
 * private stacks whatever its size. */
void acn_launch_walk( KernelFlags f, uint32_t pass, bool last, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                      const double* pos_xy, size_t first_pixel, uint32_t base, uint32_t n_cam, TileOrder order,
                      unsigned long long* accum, unsigned long long* counters );
void acn_launch_shade_hits( bool count, const LevelQ& q, hipStream_t stream, const SceneArgs& s,
                            unsigned long long* accum, unsigned long long* counters );
void acn_launch_shade( int cls, KernelFlags f, const LevelQ& q, hipStream_t stream, const SceneArgs& s,
                       unsigned long long* accum, unsigned long long* counters );
/* part: ACN_SHADE_BOTH, or one half of a fissioned launch (ACN_SHADE_DIRECT / ACN_SHADE_PATH) */
void acn_launch_shade64( KernelFlags, const LevelQ&, hipStream_t, const SceneArgs&, unsigned long long*, unsigned long long*, int part );
void acn_launch_shade16( KernelFlags, const LevelQ&, hipStream_t, const SceneArgs&, unsigned long long*, unsigned long long*, int part );
void acn_launch_shade4( KernelFlags, const LevelQ&, hipStream_t, const SceneArgs&, unsigned long long*, unsigned long long*, int part );
void acn_launch_shade1( KernelFlags, const LevelQ&, hipStream_t, const SceneArgs&, unsigned long long*, unsigned long long*, int part );
void acn_launch_hard_shadow( KernelFlags f, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                             unsigned long long* accum, unsigned long long* counters );
void acn_launch_hard_path( KernelFlags f, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                           unsigned long long* accum, unsigned long long* counters );

#define ACN_SCENE_ARGS_OF( s ) ( s ).dev, ( s ).nodes, ( s ).mats, ( s ).elems, ( s ).textures
#define ACN_TASKQ_ARGS_OF( q ) ( q ).tasks, ( q ).idx[ 0 ], ( q ).idx[ 1 ], ( q ).idx[ 2 ], ( q ).idx[ 3 ], ( q ).counts, ( q ).task_cap, ( q ).hard_shadow, ( q ).hs_cap, ( q ).emit_terms

/* k_walk< C, L, R > */
#define ACN_LW_( C, L, R ) \
    hipLaunchKernelGGL( ( k_walk< C, L, R > ), dim3( q.grid ), dim3( 256 ), lds_bytes, stream, ACN_SCENE_ARGS_OF( s ), ACN_TASKQ_ARGS_OF( q ), \
        n_cam ? ( const RayTask* )nullptr : ( const RayTask* )q.rays[ pass & 1 ], q.ray_cap, pass, pos_xy, first_pixel, base, n_cam, order, \
        q.rays[ ( pass + 1 ) & 1 ], q.ray_cap, last ? 0xFFFFFFFFu : q.private_limit, \
        q.stacks, q.stack_cap, last ? q.stack_cap : q.stack_use, q.fetch_walk, q.emit_terms, accum, counters )

/* body of acn_launch_shade<LPT>: shared by the four k_shade translation units */
#define ACN_DEFINE_LAUNCH_SHADE( NAME, LPT, CLS ) \
template< int PART > static void NAME##_part( KernelFlags f, const LevelQ& q, hipStream_t stream, const SceneArgs& s, unsigned long long* accum, unsigned long long* counters ) \
{ \
    if( f.count && f.prune ) { if( f.leaf_lights ) ACN_LS_( LPT, CLS, true, true, true, PART );  else ACN_LS_( LPT, CLS, true, false, true, PART ); } \
    else if( f.count ) { if( f.leaf_lights ) ACN_LS_( LPT, CLS, true, true, false, PART );  else ACN_LS_( LPT, CLS, true, false, false, PART ); } \
    else if( f.prune ) { if( f.leaf_lights ) ACN_LS_( LPT, CLS, false, true, true, PART );  else ACN_LS_( LPT, CLS, false, false, true, PART ); } \
    else               { if( f.leaf_lights ) ACN_LS_( LPT, CLS, false, true, false, PART ); else ACN_LS_( LPT, CLS, false, false, false, PART ); } \
} \
void NAME( KernelFlags f, const LevelQ& q, hipStream_t stream, const SceneArgs& s, unsigned long long* accum, unsigned long long* counters, int part ) \
{ \
    if( part == ACN_SHADE_DIRECT )    NAME##_part< ACN_SHADE_DIRECT >( f, q, stream, s, accum, counters ); \
    else if( part == ACN_SHADE_PATH ) NAME##_part< ACN_SHADE_PATH >( f, q, stream, s, accum, counters ); \
    else                              NAME##_part< ACN_SHADE_BOTH >( f, q, stream, s, accum, counters ); \
}
#define ACN_LS_( LPT, CLS, C, L, P, PART ) hipLaunchKernelGGL( ( k_shade< LPT, C, L, P, PART > ), dim3( q.shade_grid ), dim3( 256 ), 0, stream, ACN_SCENE_ARGS_OF( s ), \
    ( const DTask* )q.tasks, ( const uint32_t* )q.idx[ CLS ], CLS, q.task_cap, q.fetch_shade * ( 64u / LPT ), q.children, q.child_cap, q.hard_shadow, q.hard_path, q.hs_cap, q.hard_cap, q.counts, q.shard_rank, q.shard_world, accum, counters )

#endif
