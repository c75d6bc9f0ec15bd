/* acn_pipeline.h -- the level-synchronous wavefront formulation of scene_s_lum (src/scene.c:420-667) for gfx950.
 *
 * The reference evaluates one pixel by a branching recursion: specular chains (Fresnel reflection / chromatic
 * reflection / refraction, depth-1 each) with, at every diffuse shading point, a direct-light loop over
 * direct_samples*I cap samples per light and -- while depth > 10 -- a path loop over path_samples*I hemisphere
 * samples whose hits recurse with depth-10.  Every term is linear in what the recursion returns, so each pending
 * piece of work carries a colour throughput T and adds T * value into its pixel.  That turns the recursion into
 * three kinds of records and two kernels that alternate once per path level:
 *
 *   k_walk   (one LANE per sample position or per child hit)    primary ray / child hit -> depth-first walk of the
 *            specular tree with a small per-lane ray stack; light hits and background go straight to the pixel;
 *            every diffuse shading point becomes a DTask, appended to a size-class queue through a wave
 *            ballot + prefix (one atomic per wave and class).
 *   k_shade  (one WAVEFRONT per DTask; 16 / 4 / 1 lanes per task for the small size classes)   lanes = samples:
 *            lane j jumps the shading point's LCG stream ahead by 2j draws (the reference consumes exactly two
 *            draws per sample, scene.c:558,598), casts its cap sample at the light, Oren-Nayar weight, shadow ray;
 *            then the path samples: hemisphere sample, transition hit against matter; misses take the background,
 *            hits are compacted (ballot + prefix) into the HitRec queue of the next level.
 *
 * Scene access: root-compound loops run in lock-step over all lanes, so node records are fetched with wave-uniform
 * indices (scalar cache -> SGPRs); only rays that enter a CSG envelope go through the per-lane hit machine.
 *
 * Pixel accumulation is order-independent and therefore bit-reproducible: contributions are added as 2^-40
 * fixed-point integers with 64-bit integer atomics (resolution 9.1e-13, contributions clamped to +-65536, far
 * above the 1.0 at which cl_s_sat saturates).
 */
#ifndef ACN_PIPELINE_H
#define ACN_PIPELINE_H

#include "acn_device.h"

/* ---- LCG jump-ahead: x -> a^(2^K) x + c_K in one step, constants folded at compile time ---- */
struct LcgStep { uint64_t a, c; };

constexpr LcgStep lcg_pow2( int k )
{
    LcgStep s = { ACN_LCG00_A, ACN_LCG00_C };
    for( int i = 0; i < k; i++ ) { s.c = ( s.a + 1 ) * s.c; s.a = s.a * s.a; }
    return s;
}

template< int K > DEV uint64_t lcg_jump_pow2( uint64_t x )
{
    constexpr LcgStep s = lcg_pow2( K );
    return s.a * x + s.c;
}

/* jump by 2*sub draws for sub < 64 */
DEV uint64_t lcg_jump_lane( uint64_t x, int sub )
{
    if( sub & 1 )  x = lcg_jump_pow2< 1 >( x );
    if( sub & 2 )  x = lcg_jump_pow2< 2 >( x );
    if( sub & 4 )  x = lcg_jump_pow2< 3 >( x );
    if( sub & 8 )  x = lcg_jump_pow2< 4 >( x );
    if( sub & 16 ) x = lcg_jump_pow2< 5 >( x );
    if( sub & 32 ) x = lcg_jump_pow2< 6 >( x );
    return x;
}

/* ---- fixed-point pixel accumulation ---- */
#define ACN_FIX_SCALE 1099511627776.0          /* 2^40 */
#define ACN_FIX_INV   9.094947017729282e-13    /* 2^-40 */
#define ACN_FIX_CLAMP 65536.0

DEV long long to_fixed( double x )
{
    x = x < ACN_FIX_CLAMP ? x : ACN_FIX_CLAMP;
    x = x > -ACN_FIX_CLAMP ? x : -ACN_FIX_CLAMP;   /* NaN falls through both and converts to 0 below */
    if( x != x ) return 0;
    return __double2ll_rn( x * ACN_FIX_SCALE );
}

DEV void pixel_add( unsigned long long* accum, size_t i, V3 c )
{
    long long x = to_fixed( c.x ), y = to_fixed( c.y ), z = to_fixed( c.z );
    if( x ) atomicAdd( &accum[ i * 3 + 0 ], ( unsigned long long )x );
    if( y ) atomicAdd( &accum[ i * 3 + 1 ], ( unsigned long long )y );
    if( z ) atomicAdd( &accum[ i * 3 + 2 ], ( unsigned long long )z );
}

/* ---- records ---- */

/* pending ray of the specular walk */
struct RayTask
{
    V3 p, d;
    V3 T;               /* colour throughput applied to whatever this ray returns */
    double intensity;
    int depth;
    uint32_t pixel;
};

/* a diffuse shading point whose sample loops are still to run (scene.c:526-621) */
struct DTask
{
    V3 pos;              /* surface.p */
    V3 surface_d;        /* -exit_nor */
    V3 ray_projection;
    double theta_i, on_a, on_b;
    double diffuse_intensity;
    V3 Tc;               /* T * obj_color( enter_obj ) */
    uint64_t rv;         /* seed of the shading point's LCG stream (scene.c:537) */
    int depth;
    uint32_t pixel;
};

/* a path-sample hit: the arguments of the recursive scene_s_lum call of scene.c:610 */
struct HitRec
{
    V3 p, d;
    double offs;
    V3 exit_nor;
    V3 T;
    double intensity;
    int exit_obj, enter_obj;
    int depth;
    uint32_t pixel;
};

/* a shadow ray of k_shade that entered the envelope of a CSG / SDF / compound element: finished by k_hard_shadow */
struct HardShadow
{
    V3 pos, d;
    double limit;        /* distance of the light hit */
    V3 contrib;          /* what the sample adds to the pixel if it is not occluded */
    uint32_t pixel, pad;
};

/* a path ray of k_shade that did: k_hard_path finishes the transition hit */
struct HardPath
{
    V3 pos, d;
    V3 T;
    double intensity;
    int depth;
    uint32_t pixel;
};

#define ACN_NCLASS 4
enum { QC_TASKS = 0, QC_CLASS0 = 1, QC_CHILDREN = 5, QC_FLAGS = 6, QC_HARD_SHADOW = 7, QC_HARD_PATH = 8, QC_RAYS = 9, QC_CHASED = 10, QC_N = 12 };

struct Queues
{
    DTask*    tasks;
    uint32_t* idx[ ACN_NCLASS ];    /* per size class: indices into tasks[] */
    HitRec*   children;
    HardShadow* hard_shadow;
    HardPath*   hard_path;
    RayTask*    rays_out;           /* specular rays spawned by this pass, traced by the next one */
    RayTask*    loc_out;            /* k_trace_chase: the block's own next-generation queue (else nullptr) ... */
    uint32_t*   loc_count;          /* ... its counter (LDS) and capacity; what does not fit goes to rays_out */
    uint32_t    loc_cap;
    uint32_t* counts;               /* QC_* */
    uint32_t  task_cap, child_cap, hard_cap, ray_cap;
};

#ifndef ACN_CLASS0_MIN
#define ACN_CLASS0_MIN 32
#endif
/* lanes per task of the size classes, and the smallest sample count that goes to each */
DEV int size_class( uint64_t n )
{
    return n > ACN_CLASS0_MIN ? 0 : n > 8 ? 1 : n > 2 ? 2 : 3;
}

/* one atomic per wave: every lane with `want` gets a distinct slot */
DEV uint32_t wave_alloc( uint32_t* counter, bool want )
{
    unsigned long long mask = __ballot( want );
    if( !want ) return 0xFFFFFFFFu;
    int lane = ( int )( threadIdx.x & 63 );
    int leader = __builtin_amdgcn_readfirstlane( __ffsll( ( long long )mask ) - 1 );
    uint32_t base = 0;
    if( lane == leader ) base = atomicAdd( counter, ( uint32_t )__popcll( mask ) );
    base = ( uint32_t )__builtin_amdgcn_readlane( ( int )base, leader );
    return base + ( uint32_t )__popcll( mask & ( ( 1ull << lane ) - 1ull ) );
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* scene_s_lum for one hit, everything except the two sample loops (scene.c:420-537, 623-664).  Specular children
 * (Fresnel reflection, chromatic reflection, refraction) are appended to the ray queue of the next pass; the diffuse
 * block becomes a DTask. */
DEV void push_ray( const Queues& q, bool want, V3 p, V3 d, V3 T, double intensity, int depth, uint32_t pixel )
{
    if( q.loc_out )   /* block-local queue of k_trace_chase (wave-uniform branch) */
    {
        uint32_t ls = wave_alloc( q.loc_count, want );
        if( want && ls < q.loc_cap )
        {
            RayTask& c = q.loc_out[ ls ];
            c.p = p; c.d = d; c.T = T; c.intensity = intensity; c.depth = depth; c.pixel = pixel;
        }
        want = want && ls >= q.loc_cap;   /* overflow of the local queue: the ray joins the next ordinary pass */
    }
    uint32_t slot = wave_alloc( &q.counts[ QC_RAYS ], want );
    if( want )
    {
        if( slot < q.ray_cap )
        {
            RayTask& c = q.rays_out[ slot ];
            c.p = p; c.d = d; c.T = T; c.intensity = intensity; c.depth = depth; c.pixel = pixel;
        }
        else
        {
            atomicOr( &q.counts[ QC_FLAGS ], ACN_FLAG_CHILD_OVERFLOW );
        }
    }
}

template< class CT >
DEV void shade_hit( const DevScene& sc, const Queues& q, V3 rp, V3 rd, double offs, const Trans& trans, int depth,
                    double intensity, V3 T, uint32_t pixel, V3& acc, CT* cnt )
{
    const double min_intensity = sc.prm.trace_min_intensity;
    /* every lane takes part in the queue appends below; `go` masks the ones with nothing to shade */
    bool go = !( depth == 0 || intensity < min_intensity );
    if( go ) cnt->inc( CNT_LUM );
    V3 pos = ray_pos( rp, rd, offs );
    MatP enter_obj = ( go && trans.enter_obj >= 0 ) ? &sc.mats[ trans.enter_obj ] : nullptr;
    MatP exit_obj  = ( go && trans.exit_obj  >= 0 ) ? &sc.mats[ trans.exit_obj  ] : nullptr;

    if( go && enter_obj && enter_obj->radiance > 0 )   /* :432-437 */
    {
        double diff_sqr = v_diff_sqr( pos, ld3( sc.nodes[ trans.enter_obj ].pos ) );
        double light_intensity = ( diff_sqr > 0 ) ? ( enter_obj->radiance / diff_sqr ) : F3_MAG;
        V3 c = v_mlf( obj_color_dev( sc, trans.enter_obj, pos ), light_intensity * intensity );
        acc.x += T.x * c.x; acc.y += T.y * c.y; acc.z += T.z * c.z;
        go = false;
    }

    double trix = 1.0;
    double fresnel_reflectivity = 0, chromatic_reflectivity = 0, diffuse_reflectivity = 0;
    double on_a = 1.0, on_b = 0.0;
    bool transparent = false;
    V3 enter_color = mk( 1, 1, 1 );
    if( go && enter_obj )   /* :448-462 */
    {
        trix = enter_obj->refractive_index;
        fresnel_reflectivity   = ( enter_obj->fresnel_reflectivity != 0 && enter_obj->refractive_index != 1.0 ) ? 1.0 : 0.0;
        chromatic_reflectivity = enter_obj->chromatic_reflectivity;
        diffuse_reflectivity   = enter_obj->diffuse_reflectivity;
        transparent            = v_sqr( ld3( enter_obj->transparency ) ) > 0;
        double sigma           = enter_obj->sigma;
        if( sigma > 0 )
        {
            double sigma_sqr = f_sqr( sigma );
            on_a = 1.0 - 0.5 * sigma_sqr / ( sigma_sqr + 0.33 );
            on_b = 0.45 * sigma_sqr / ( sigma_sqr + 0.09 );
        }
        enter_color = obj_color_dev( sc, trans.enter_obj, pos );
    }
    if( go && exit_obj )   /* :464-470 and the absorption of :656-664, which scales everything this call returns */
    {
        trix /= exit_obj->refractive_index;
        fresnel_reflectivity = 1.0;
        diffuse_reflectivity = chromatic_reflectivity = 0;
        transparent = true;
        if( offs > 0 )
        {
            T.x *= acn_pow( exit_obj->transparency[ 0 ], offs );
            T.y *= acn_pow( exit_obj->transparency[ 1 ], offs );
            T.z *= acn_pow( exit_obj->transparency[ 2 ], offs );
        }
    }

    /* fresnel reflection :473-495 */
    {
        bool f = go && fresnel_reflectivity > 0 && intensity >= min_intensity;
        V3 out_d = rd;
        double reflectance = 0;
        if( f ) reflectance = fresnel_reflection( rd, trans.exit_nor, trix, &out_d ) * fresnel_reflectivity;
        push_ray( q, f, pos, out_d, T, reflectance * intensity, depth - 1, pixel );
        if( f ) intensity *= ( 1.0 - reflectance );
    }

    /* chromatic reflection :498-523 */
    {
        bool f = go && chromatic_reflectivity > 0 && intensity >= min_intensity;
        V3 out_d = rd;
        if( f ) out_d = v_reflection( rd, trans.exit_nor );
        push_ray( q, f, pos, out_d, v_mld( T, enter_color ), chromatic_reflectivity * intensity, depth - 1, pixel );
        if( f ) intensity *= ( 1.0 - chromatic_reflectivity );
    }

    /* diffuse reflection :526-537: hand the sample loops to k_shade */
    bool diffuse = go && intensity * diffuse_reflectivity >= min_intensity;
    double diffuse_intensity = intensity * diffuse_reflectivity;
    uint64_t n_direct = 0, n_path = 0;
    if( diffuse )
    {
        if( sc.nodes[ sc.light_root ].child1 > 0 )
        {
            n_direct = ( uint64_t )( sc.prm.direct_samples * diffuse_intensity );
            n_direct = ( n_direct == 0 ) ? 1 : n_direct;
        }
        if( sc.prm.path_samples && depth > 10 )
        {
            n_path = ( uint64_t )( sc.prm.path_samples * diffuse_intensity );
            n_path = ( n_path == 0 ) ? 1 : n_path;
        }
    }
    bool emit = diffuse && ( n_direct | n_path ) != 0;
    {
        uint32_t slot = wave_alloc( &q.counts[ QC_TASKS ], emit );
        bool ok = emit && slot < q.task_cap;
        if( emit && !ok ) atomicOr( &q.counts[ QC_FLAGS ], ACN_FLAG_TASK_OVERFLOW );
        int cls = ok ? size_class( n_direct > n_path ? n_direct : n_path ) : -1;
        if( ok )
        {
            DTask& t = q.tasks[ slot ];
            V3 surface_d = v_neg( trans.exit_nor );
            t.pos = pos;
            t.surface_d = surface_d;
            t.theta_i = acn_acos( -v_mlv( rd, surface_d ) );
            t.ray_projection = v_of_length( v_orthogonal_projection( rd, surface_d ), 1.0 );
            t.on_a = on_a; t.on_b = on_b;
            t.diffuse_intensity = diffuse_intensity;
            t.Tc = v_mld( T, enter_color );
            t.rv = v_random_seed( pos, 3294479285ull ) + v_random_seed( surface_d, 3247146734ull );
            t.depth = depth;
            t.pixel = pixel;
        }
        for( int k = 0; k < ACN_NCLASS; k++ )
        {
            uint32_t is = wave_alloc( &q.counts[ QC_CLASS0 + k ], cls == k );
            if( cls == k ) q.idx[ k ][ is ] = slot;
        }
    }
    if( diffuse ) intensity *= ( 1.0 - diffuse_reflectivity );

    /* refraction :633-653 */
    {
        bool f = go && transparent && intensity >= min_intensity;
        V3 out_d = rd;
        if( f ) out_d = fresnel_refraction( rd, trans.exit_nor, trix );
        push_ray( q, f, ray_pos( rp, rd, offs + 2.0 * F3_EPS ), out_d, T, intensity, depth - 1, pixel );
    }
}

DEV void wave_add_counters( unsigned long long* global, const Cnt< true >& mine )
{
    for( int k = 0; k < CNT_N; k++ )
    {
        unsigned long long v = mine.c[ k ];
        for( int off = 32; off > 0; off >>= 1 ) v += __shfl_down( v, off, 64 );
        if( ( threadIdx.x & 63 ) == 0 && v ) atomicAdd( &global[ k ], v );
    }
}
DEV void wave_add_counters( unsigned long long*, const Cnt< false >& ) {}

#ifndef ACN_SHADE_WAVES
#define ACN_SHADE_WAVES 4
#endif
#ifndef ACN_WALK_WAVES
#define ACN_WALK_WAVES 4
#endif
/* Kernel parameter convention: every buffer is passed as its own __restrict__ pointer and the DevScene / Queues views
 * are rebuilt inside (scene arrays are then read through the constant address space, see acn_device.h). */
#define ACN_SCENE_PARAMS  DevScene sc_in, const GNode* __restrict__ p_nodes, const GMat* __restrict__ p_mats, const int32_t* __restrict__ p_elems, const acn_texture* __restrict__ p_textures
#define ACN_SCENE_ARGS( h ) ( h )->dev, ( h )->d_nodes, ( h )->d_mats, ( h )->d_elems, ( h )->d_textures
#define ACN_SCENE_VIEW    DevScene sc = sc_in; sc.nodes = ( NodeP )p_nodes; sc.mats = ( MatP )p_mats; sc.elems = ( ElemP )p_elems; sc.textures = ( TexP )p_textures; sc.flags = p_counts + QC_FLAGS; sc.lds_stack = ACN_NO_LDS_STACK;

/* LDS staging of the node array (kernels whose node reads are per-lane: the CSG machines).  The block copies the
 * GNode array into dynamic shared memory once; per-lane node reads then are ds_read instead of global loads. */
#define ACN_STAGE_NODES( sc ) \
    { \
        const double* src_ = ( const double* )p_nodes; \
        uint32_t words_ = ( sc ).n_nodes * ( uint32_t )( sizeof( GNode ) / sizeof( double ) ); \
        for( uint32_t k_ = threadIdx.x; k_ < words_; k_ += blockDim.x ) acn_lds_raw[ k_ ] = src_[ k_ ]; \
        __syncthreads(); \
    }

#define ACN_WALK_QUEUE_PARAMS DTask* __restrict__ p_tasks, uint32_t* __restrict__ p_idx0, uint32_t* __restrict__ p_idx1, \
    uint32_t* __restrict__ p_idx2, uint32_t* __restrict__ p_idx3, uint32_t* __restrict__ p_counts, uint32_t task_cap, \
    RayTask* __restrict__ p_rays_out, uint32_t ray_cap
#define ACN_WALK_QUEUE_VIEW \
    Queues q; \
    q.tasks = p_tasks; q.idx[ 0 ] = p_idx0; q.idx[ 1 ] = p_idx1; q.idx[ 2 ] = p_idx2; q.idx[ 3 ] = p_idx3; \
    q.children = nullptr; q.counts = p_counts; q.task_cap = task_cap; q.child_cap = 0; \
    q.hard_shadow = nullptr; q.hard_path = nullptr; q.hard_cap = 0; q.rays_out = p_rays_out; q.ray_cap = ray_cap; \
    q.loc_out = nullptr; q.loc_count = nullptr; q.loc_cap = 0;

/* one ray: scene_s_trans_hit, then the hit is shaded; what it spawns goes to the queues.  Every lane of a wave must call
 * this (the queue appends are wave-wide); lanes without a ray pass live = false. */
template< class SCL, class CT >
DEV void trace_one( const DevScene& sc, const SCL& scl, const Queues& q, const RayTask& t, bool live,
                    unsigned long long* __restrict__ accum, CT* cnt )
{
    V3 acc = mk( 0, 0, 0 );
    Trans trans;
    trans.exit_nor = mk( 0, 0, 0 ); trans.exit_obj = -1; trans.enter_obj = -1;
    double offs = F3_INF;
    if( live ) offs = scene_trans_hit_dev( scl, t.p, t.d, &trans, cnt );
    bool hit = live && offs < F3_INF;
    if( live && !hit )
    {
        V3 c = v_mlf( ld3( sc.prm.background_color ), t.intensity );
        acc = v_mld( t.T, c );
    }
    shade_hit( sc, q, t.p, t.d, hit ? offs : 0.0, trans, hit ? t.depth : 0, t.intensity, t.T, t.pixel, acc, cnt );
    if( live ) pixel_add( accum, t.pixel, acc );
}

/* One pass of the specular walk: one lane per ray.  rays_in == nullptr: the rays are the camera rays of the sample positions
 * (lum_machine_s_func, scene.c:976-1011); otherwise they come from the ray queue the previous pass filled.  Each ray
 * is traced (scene_s_trans_hit) and its hit shaded; what it spawns goes to the next pass / the shading-task queues. */
template< bool COUNT, class SCL >
DEV void trace_rays_body( const DevScene& sc, const SCL& scl, const Queues& q, const RayTask* __restrict__ rays_in,
                          const double* __restrict__ pos_xy, size_t first_pixel, uint32_t base, uint32_t n,
                          unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Cnt< COUNT > cnt;
    cnt.clear();
    bool live = i < n;
    RayTask t;
    t.p = mk( 0, 0, 0 ); t.d = mk( 0, 0, 1 ); t.T = mk( 0, 0, 0 ); t.intensity = 0; t.depth = 0; t.pixel = 0;
    if( live )
    {
        if( !rays_in )   /* the camera rays of the sample positions (wave-uniform branch) */
        {
            uint32_t pixel = base + i;
            double mx, my;
            if( pos_xy ) { mx = pos_xy[ ( size_t )pixel * 2 ]; my = pos_xy[ ( size_t )pixel * 2 + 1 ]; }
            else
            {
                size_t pix = first_pixel + pixel;
                mx = ( double )( pix % sc.prm.image_width ) + 0.5;
                my = ( double )( pix / sc.prm.image_width ) + 0.5;
            }
            camera_ray( sc, mx, my, &t.p, &t.d );
            t.T = mk( 1, 1, 1 ); t.intensity = 1.0; t.depth = ( int )sc.prm.trace_depth; t.pixel = pixel;
        }
        else
        {
            t = rays_in[ i ];
        }
    }
    trace_one( sc, scl, q, t, live, accum, &cnt );
    wave_add_counters( counters, cnt );
}

#ifndef ACN_TRACE_WAVES
#define ACN_TRACE_WAVES ACN_WALK_WAVES
#endif
#ifndef ACN_HPATH_WAVES
#define ACN_HPATH_WAVES ACN_WALK_WAVES
#endif
template< bool COUNT, bool LDS, bool PRUNE >
__global__ __launch_bounds__( 256, ACN_TRACE_WAVES )
void k_trace_rays( ACN_SCENE_PARAMS, ACN_WALK_QUEUE_PARAMS, const RayTask* __restrict__ rays_in,
                   const double* __restrict__ pos_xy, size_t first_pixel, uint32_t base, uint32_t n,
                   unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_SCENE_VIEW
    ACN_WALK_QUEUE_VIEW
    if( sc_in.lds_stack != ACN_NO_LDS_STACK ) sc.lds_stack = LDS ? sc.n_nodes * ( uint32_t )sizeof( GNode ) : 0u;   /* the CSG stacks follow the staged nodes */
    if constexpr( LDS )
    {
        ACN_STAGE_NODES( sc )
        trace_rays_body< COUNT >( sc, scene_view< PRUNE >( sc, ( LdsNodeP )acn_lds_raw ), q, rays_in, pos_xy, first_pixel, base, n, accum, counters );
    }
    else
    {
        trace_rays_body< COUNT >( sc, scene_view< PRUNE >( sc, sc.nodes ), q, rays_in, pos_xy, first_pixel, base, n, accum, counters );
    }
}

/* The tail of the specular walk in ONE launch.  Late generations hold few rays, but every generation is a launch plus a
 * host round trip, and a level has ~40 of them: on the share of a frame that one of 8 GPUs renders the chain of
 * launches, not the work, sets the time.  Once a generation is small (ACN_CHASE_MAX rays) each block takes 256 of its
 * rays and follows THEIR descendants by itself: children go to a block-private queue (two ping-pong regions of
 * ACN_CHASE_CAP rays in global memory, counter in LDS) and are traced by the same block in the next round.  No block
 * waits for another one, every loop is bounded (ACN_CHASE_ROUNDS), and whatever does not fit the private queue or is
 * left after the last round goes to the ordinary ray queue for an ordinary pass -- so the kernel cannot hang and never
 * loses a ray.  Per ray the computation is the one of k_trace_rays; only the order of queue entries differs, which the
 * results do not depend on. */
#define ACN_CHASE_CAP    2048
#define ACN_CHASE_ROUNDS 64
template< bool COUNT, bool LDS, bool PRUNE >
__global__ __launch_bounds__( 256, ACN_TRACE_WAVES )
void k_trace_chase( ACN_SCENE_PARAMS, ACN_WALK_QUEUE_PARAMS, const RayTask* __restrict__ rays_in, uint32_t n,
                    RayTask* __restrict__ chase_buf, unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_SCENE_VIEW
    ACN_WALK_QUEUE_VIEW
    if( sc_in.lds_stack != ACN_NO_LDS_STACK ) sc.lds_stack = LDS ? sc.n_nodes * ( uint32_t )sizeof( GNode ) : 0u;
    if constexpr( LDS ) ACN_STAGE_NODES( sc )
    __shared__ uint32_t loc_n[ 2 ];
    Cnt< COUNT > cnt;
    cnt.clear();
    RayTask* region[ 2 ] = { chase_buf + ( size_t )blockIdx.x * 2 * ACN_CHASE_CAP, chase_buf + ( ( size_t )blockIdx.x * 2 + 1 ) * ACN_CHASE_CAP };
    q.loc_cap = ACN_CHASE_CAP;
    uint32_t traced = 0;
    /* round 0 reads the block's 256 rays of the input generation, round r > 0 the private queue round r - 1 filled */
    const RayTask* in = rays_in + ( size_t )blockIdx.x * blockDim.x;
    uint32_t first = blockIdx.x * blockDim.x;
    uint32_t n_loc = first < n ? ( n - first < blockDim.x ? n - first : blockDim.x ) : 0;
    for( int round = 0; round <= ACN_CHASE_ROUNDS && n_loc > 0; round++ )
    {
        const int o = round & 1;
        const bool last = round == ACN_CHASE_ROUNDS;    /* the last round only hands what is left to the ordinary queue */
        if( threadIdx.x == 0 ) loc_n[ o ] = 0;
        __syncthreads();
        q.loc_out = last ? nullptr : region[ o ];
        q.loc_count = &loc_n[ o ];
        for( uint32_t j0 = 0; j0 < n_loc; j0 += blockDim.x )
        {
            uint32_t j = j0 + threadIdx.x;
            bool live = j < n_loc;
            RayTask t;
            t.p = mk( 0, 0, 0 ); t.d = mk( 0, 0, 1 ); t.T = mk( 0, 0, 0 ); t.intensity = 0; t.depth = 0; t.pixel = 0;
            if( live ) t = in[ j ];
            if( last ) push_ray( q, live, t.p, t.d, t.T, t.intensity, t.depth, t.pixel );
            else
            {
                if( live ) traced++;
                if constexpr( LDS ) trace_one( sc, scene_view< PRUNE >( sc, ( LdsNodeP )acn_lds_raw ), q, t, live, accum, &cnt );
                else                trace_one( sc, scene_view< PRUNE >( sc, sc.nodes ), q, t, live, accum, &cnt );
            }
        }
        __syncthreads();
        n_loc = last ? 0 : loc_n[ o ];
        if( n_loc > ACN_CHASE_CAP ) n_loc = ACN_CHASE_CAP;   /* the excess went to the ordinary queue */
        in = region[ o ];
        __syncthreads();   /* loc_n[ o ] is read by everyone before the round after next resets it */
    }
    /* statistics: rays traced here */
    for( int o = 32; o > 0; o >>= 1 ) traced += __shfl_down( traced, o );
    if( ( threadIdx.x & 63 ) == 0 && traced ) atomicAdd( &p_counts[ QC_CHASED ], traced );
    wave_add_counters( counters, cnt );
}

/* first pass of a level >= 1: one lane per path-sample hit (the recursive scene_s_lum call of scene.c:610) */
template< bool COUNT >
__global__ __launch_bounds__( 256, ACN_WALK_WAVES )
void k_shade_hits( ACN_SCENE_PARAMS, ACN_WALK_QUEUE_PARAMS, const HitRec* __restrict__ recs, uint32_t n,
                   unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_SCENE_VIEW
    ACN_WALK_QUEUE_VIEW
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Cnt< COUNT > cnt;
    cnt.clear();
    bool live = i < n;
    HitRec r;
    r.p = mk( 0, 0, 0 ); r.d = mk( 0, 0, 1 ); r.offs = 0; r.exit_nor = mk( 0, 0, 0 ); r.T = mk( 0, 0, 0 ); r.intensity = 0;
    r.exit_obj = -1; r.enter_obj = -1; r.depth = 0; r.pixel = 0;
    if( live ) r = recs[ i ];
    V3 acc = mk( 0, 0, 0 );
    Trans trans;
    trans.exit_nor = r.exit_nor; trans.exit_obj = r.exit_obj; trans.enter_obj = r.enter_obj;
    shade_hit( sc, q, r.p, r.d, r.offs, trans, r.depth, r.intensity, r.T, r.pixel, acc, &cnt );
    if( live ) pixel_add( accum, r.pixel, acc );
    wave_add_counters( counters, cnt );
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* k_shade: the two sample loops of a diffuse shading point, LPT lanes per task */

template< int LPT > DEV double group_sum( double v )
{
    for( int off = LPT / 2; off > 0; off >>= 1 ) v += __shfl_xor( v, off, 64 );
    return v;
}

template< int LPT > DEV uint64_t lcg_stride( uint64_t x )   /* jump by 2*LPT draws */
{
    if( LPT == 64 ) return lcg_jump_pow2< 7 >( x );
    if( LPT == 16 ) return lcg_jump_pow2< 5 >( x );
    if( LPT == 4 )  return lcg_jump_pow2< 3 >( x );
    return lcg_jump_pow2< 1 >( x );
}

/* LEAF_LIGHTS: every light is a plane / sphere / squaroid-free leaf, so the kernel contains no call into the CSG
 * machine at all (the usual case); otherwise the light hit goes through the generic element test. */
template< int LPT, bool COUNT, bool LEAF_LIGHTS, bool PRUNE >
__global__ __launch_bounds__( 256, ACN_SHADE_WAVES )
void k_shade( ACN_SCENE_PARAMS, const DTask* __restrict__ tasks, const uint32_t* __restrict__ idx, uint32_t n_tasks,
              HitRec* __restrict__ p_children, uint32_t child_cap, HardShadow* __restrict__ p_hard_shadow,
              HardPath* __restrict__ p_hard_path, uint32_t hard_cap, uint32_t* __restrict__ p_counts,
              unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_SCENE_VIEW
    const auto scp = scene_view< PRUNE >( sc, sc.nodes );   /* the scene as the two root-traversal fast paths see it */
    Queues q;
    q.tasks = nullptr; q.children = p_children; q.counts = p_counts; q.task_cap = 0; q.child_cap = child_cap;
    q.hard_shadow = p_hard_shadow; q.hard_path = p_hard_path; q.hard_cap = hard_cap; q.rays_out = nullptr; q.ray_cap = 0;
    q.loc_out = nullptr; q.loc_count = nullptr; q.loc_cap = 0;
    constexpr int G = 64 / LPT;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPT;
    const int grp = lane / LPT;
    const uint32_t wave = ( blockIdx.x * blockDim.x + threadIdx.x ) >> 6;
    const uint32_t n_waves = ( gridDim.x * blockDim.x ) >> 6;
    Cnt< COUNT > cnt;
    cnt.clear();
    const V3 bg = ld3( sc.prm.background_color );

    for( uint32_t base = wave * G; base < n_tasks; base += n_waves * G )
    {
        uint32_t ti = base + grp;
        if( ti >= n_tasks ) continue;
        uint32_t slot = ( ( ElemP )( const void* )idx )[ ti ];
        if( LPT == 64 ) slot = __builtin_amdgcn_readfirstlane( slot );
        const DTask ACN_CONST& t = ( ( const DTask ACN_CONST* )tasks )[ slot ];
        const V3 pos = ldc( t.pos ), surface_d = ldc( t.surface_d ), ray_projection = ldc( t.ray_projection );
        const double theta_i = t.theta_i, on_a = t.on_a, on_b = t.on_b, diffuse_intensity = t.diffuse_intensity;
        double sin_i = 0, cos_i = 1;   /* loop invariant half of the Oren-Nayar term */
        if( on_b > 0 ) acn_sincos( theta_i, &sin_i, &cos_i );
        uint64_t rv = t.rv;
        V3 lum = mk( 0, 0, 0 );   /* lum_l of scene.c:539, identical in all lanes of the group after each reduction */

        /* ---- direct light, scene.c:542-581 ---- */
        NodeP light = &sc.nodes[ sc.light_root ];
        const int n_lights = light->child1;
        for( int li = 0; li < n_lights; li++ )
        {
            int light_idx = __builtin_amdgcn_readfirstlane( sc.elems[ light->child0 + li ] );
            NodeP light_src = &sc.nodes[ light_idx ];
            MatP light_mat = &sc.mats[ light_idx ];
            V3 fov_d; double cos_rs;
            obj_fov_dev( light_src, pos, &fov_d, &cos_rs );
            M3 src_con = m_transposed( m_con_z( fov_d ) );
            double cyl_hgt = 1 - cos_rs;
            uint64_t direct_samples = ( uint64_t )( sc.prm.direct_samples * diffuse_intensity );
            direct_samples = ( direct_samples == 0 ) ? 1 : direct_samples;
            V3 light_pos = ld3( light_src->pos );
            double radiance = light_mat->radiance;
            const V3 light_color = obj_color_dev( sc, light_idx, light_pos );   /* scene.c:552 */

            double s = 0;
            uint64_t rvj = lcg_jump_lane( rv, sub );
            for( uint64_t j = sub; j < direct_samples; j += LPT )
            {
                uint64_t r = rvj;
                rvj = lcg_stride< LPT >( rvj );
                cnt.inc( CNT_CAP_SAMPLE );
                V3 out_d = m_mlv( src_con, v_random_sphere_cap( &r, cyl_hgt ) );
                double weight = v_mlv( out_d, surface_d );
                if( weight <= 0 ) continue;
                double a;
                if( LEAF_LIGHTS ) a = leaf_element_hit< false >( light_src, light_src->type, pos, out_d, nullptr, &cnt );
                else a = light_hit_call( sc, light_idx, pos, out_d, &cnt );
                if( a >= F3_INF ) continue;
                if( on_b > 0 ) weight = oren_nayar_weight_pre( weight, theta_i, sin_i, cos_i, on_a, on_b, out_d, surface_d, ray_projection );
                cnt.inc( CNT_SHADOW_RAY );
                V3 hit_pos = ray_pos( pos, out_d, a );
                double diff_sqr = v_diff_sqr( hit_pos, light_pos );
                double local_intensity = ( diff_sqr > 0 ) ? ( radiance / diff_sqr ) : F3_MAG;
                double c = local_intensity * weight * diffuse_intensity;
                int occ = root_occluded_fast( scp, sc.matter_root, pos, out_d, a, &cnt );
                if( occ == 0 ) s += c;
                /* hard shadow rays: compacted into the queue of k_hard_shadow, which adds c itself if unoccluded */
                uint32_t hs = wave_alloc( &q.counts[ QC_HARD_SHADOW ], occ == 2 );
                if( occ == 2 )
                {
                    if( hs < q.hard_cap )
                    {
                        HardShadow& h = q.hard_shadow[ hs ];
                        double f = c * ( 2.0 * cyl_hgt / direct_samples );
                        V3 Tc = ldc( t.Tc );
                        h.pos = pos; h.d = out_d; h.limit = a;
                        h.contrib = mk( Tc.x * ( light_color.x * f ), Tc.y * ( light_color.y * f ), Tc.z * ( light_color.z * f ) );
                        h.pixel = t.pixel; h.pad = 0;
                    }
                    else
                    {
                        atomicOr( &q.counts[ QC_FLAGS ], ACN_FLAG_CHILD_OVERFLOW );
                    }
                }
            }
            rv = lcg00_jump( rv, 2 * direct_samples );
            s = group_sum< LPT >( s );
            double f = s * ( 2.0 * cyl_hgt / direct_samples );
            lum.x += light_color.x * f; lum.y += light_color.y * f; lum.z += light_color.z * f;
        }

        /* ---- path tracing, scene.c:584-621 ---- */
        if( sc.prm.path_samples && t.depth > 10 )
        {
            M3 out_con = m_transposed( m_con_z( surface_d ) );
            uint64_t path_samples = ( uint64_t )( sc.prm.path_samples * diffuse_intensity );
            path_samples = ( path_samples == 0 ) ? 1 : path_samples;
            const double norm = 2.0 / path_samples;
            const V3 Tchild = v_mlf( ldc( t.Tc ), norm );
            double bsum = 0;
            uint64_t rvj = lcg_jump_lane( rv, sub );
            for( uint64_t j = sub; j < path_samples; j += LPT )
            {
                uint64_t r = rvj;
                rvj = lcg_stride< LPT >( rvj );
                cnt.inc( CNT_CAP_SAMPLE );
                V3 out_d = m_mlv( out_con, v_random_sphere_cap( &r, 1.0 ) );
                double weight = v_mlv( out_d, surface_d );
                bool live = weight > 0;
                double a = F3_INF;
                Trans trans;
                trans.exit_nor = mk( 0, 0, 0 ); trans.exit_obj = -1; trans.enter_obj = -1;
                bool hard = false;
                if( live )
                {
                    if( on_b > 0 ) weight = oren_nayar_weight_pre( weight, theta_i, sin_i, cos_i, on_a, on_b, out_d, surface_d, ray_projection );
                    a = root_trans_hit_fast( scp, sc.matter_root, pos, out_d, &trans, &hard, &cnt );
                }
                bool hit = live && !hard && a < sc.prm.max_path_length;
                if( live && !hard && !hit ) bsum += weight * diffuse_intensity;
                /* hard path rays: the transition hit is finished by k_hard_path */
                uint32_t hp = wave_alloc( &q.counts[ QC_HARD_PATH ], hard );
                if( hard )
                {
                    if( hp < q.hard_cap )
                    {
                        HardPath& h = q.hard_path[ hp ];
                        h.pos = pos; h.d = out_d; h.T = Tchild; h.intensity = weight * diffuse_intensity;
                        h.depth = t.depth - 10; h.pixel = t.pixel;
                    }
                    else
                    {
                        atomicOr( &q.counts[ QC_FLAGS ], ACN_FLAG_CHILD_OVERFLOW );
                    }
                }
                /* compaction of the surviving path rays into the next level's queue */
                uint32_t cs = wave_alloc( &q.counts[ QC_CHILDREN ], hit );
                if( hit )
                {
                    if( cs < q.child_cap )
                    {
                        HitRec& c = q.children[ cs ];
                        c.p = pos; c.d = out_d; c.offs = a; c.exit_nor = trans.exit_nor; c.T = Tchild;
                        c.intensity = weight * diffuse_intensity;
                        c.exit_obj = trans.exit_obj; c.enter_obj = trans.enter_obj;
                        c.depth = t.depth - 10; c.pixel = t.pixel;
                    }
                    else
                    {
                        atomicOr( &q.counts[ QC_FLAGS ], ACN_FLAG_CHILD_OVERFLOW );
                    }
                }
            }
            bsum = group_sum< LPT >( bsum ) * norm;
            lum.x += bg.x * bsum; lum.y += bg.y * bsum; lum.z += bg.z * bsum;
        }

        if( sub == 0 ) pixel_add( accum, t.pixel, v_mld( ldc( t.Tc ), lum ) );
    }
    wave_add_counters( counters, cnt );
}

/* the shadow rays k_shade could not decide inline: full occlusion test, one lane per ray */
template< bool COUNT, bool LDS, bool PRUNE >
__global__ __launch_bounds__( 256, ACN_WALK_WAVES )
void k_hard_shadow( ACN_SCENE_PARAMS, const HardShadow* __restrict__ recs, uint32_t n, uint32_t* __restrict__ p_counts,
                    unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_SCENE_VIEW
    if( sc_in.lds_stack != ACN_NO_LDS_STACK ) sc.lds_stack = LDS ? sc.n_nodes * ( uint32_t )sizeof( GNode ) : 0u;   /* the CSG stacks follow the staged nodes */
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Cnt< COUNT > cnt;
    cnt.clear();
    if constexpr( LDS ) ACN_STAGE_NODES( sc )
    if( i < n )
    {
        HardShadow r = recs[ i ];
        bool occ;
        if constexpr( LDS ) occ = root_occluded( scene_view< PRUNE >( sc, ( LdsNodeP )acn_lds_raw ), sc.matter_root, r.pos, r.d, r.limit, &cnt );
        else                occ = root_occluded( scene_view< PRUNE >( sc, sc.nodes ), sc.matter_root, r.pos, r.d, r.limit, &cnt );
        if( !occ ) pixel_add( accum, r.pixel, r.contrib );
    }
    wave_add_counters( counters, cnt );
}

/* the path rays k_shade could not finish inline: full transition hit; hits join the next level's HitRec queue */
template< bool COUNT, bool LDS, bool PRUNE >
__global__ __launch_bounds__( 256, ACN_HPATH_WAVES )
void k_hard_path( ACN_SCENE_PARAMS, const HardPath* __restrict__ recs, uint32_t n, HitRec* __restrict__ p_children, uint32_t child_cap,
                  uint32_t* __restrict__ p_counts, unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_SCENE_VIEW
    if( sc_in.lds_stack != ACN_NO_LDS_STACK ) sc.lds_stack = LDS ? sc.n_nodes * ( uint32_t )sizeof( GNode ) : 0u;   /* the CSG stacks follow the staged nodes */
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Cnt< COUNT > cnt;
    cnt.clear();
    if constexpr( LDS ) ACN_STAGE_NODES( sc )
    bool hit = false;
    HardPath r;
    Trans trans;
    trans.exit_nor = mk( 0, 0, 0 ); trans.exit_obj = -1; trans.enter_obj = -1;
    double a = F3_INF;
    if( i < n )
    {
        r = recs[ i ];
        if constexpr( LDS ) a = root_trans_hit( scene_view< PRUNE >( sc, ( LdsNodeP )acn_lds_raw ), sc.matter_root, r.pos, r.d, &trans, &cnt );
        else                a = root_trans_hit( scene_view< PRUNE >( sc, sc.nodes ), sc.matter_root, r.pos, r.d, &trans, &cnt );
        hit = a < sc.prm.max_path_length;
        if( !hit )
        {
            V3 c = v_mlf( ld3( sc.prm.background_color ), r.intensity );
            pixel_add( accum, r.pixel, v_mld( r.T, c ) );
        }
    }
    uint32_t cs = wave_alloc( &p_counts[ QC_CHILDREN ], hit );
    if( hit )
    {
        if( cs < child_cap )
        {
            HitRec& c = p_children[ cs ];
            c.p = r.pos; c.d = r.d; c.offs = a; c.exit_nor = trans.exit_nor; c.T = r.T;
            c.intensity = r.intensity;
            c.exit_obj = trans.exit_obj; c.enter_obj = trans.enter_obj;
            c.depth = r.depth; c.pixel = r.pixel;
        }
        else
        {
            atomicOr( &p_counts[ QC_FLAGS ], ACN_FLAG_CHILD_OVERFLOW );
        }
    }
    wave_add_counters( counters, cnt );
}


#endif /* ACN_PIPELINE_H */
