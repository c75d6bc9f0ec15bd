/* actinon_hip.hip -- libactinon_hip.so: kernels + the C ABI of include/actinon_hip.h (gfx950 only). */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <mutex>
#include <string>
#include <vector>
#include <functional>
#include <algorithm>
#include <thread>
#include <atomic>
#include <condition_variable>

#include "acn_launch.h"
#include "acn_chunkplan.h"

/* ------------------------------------------------------------------------------------------------------------------ */
/* error plumbing */
static thread_local std::string g_last_error;
static int fail( int code, const std::string& msg ) { g_last_error = msg; return code; }
extern "C" const char* acn_last_error( void ) { return g_last_error.c_str(); }

#define HIP_TRY( expr ) do { hipError_t e_ = ( expr ); if( e_ != hipSuccess ) \
    return fail( ACN_ERR_DEVICE, std::string( #expr ) + ": " + hipGetErrorString( e_ ) ); } while( 0 )

struct StageEvents { hipEvent_t a, b; int stage; };

/* a persistent host thread per lane (creating a thread per call costs a HIP per-thread initialisation each time) */
struct LaneWorker
{
    std::thread thread;
    std::mutex m;
    std::condition_variable cv;
    std::function< void() > job;
    bool has_job = false, done = true, quit = false;
    void start()
    {
        thread = std::thread( [ this ]()
        {
            for( ;; )
            {
                std::function< void() > j;
                {
                    std::unique_lock< std::mutex > lk( m );
                    cv.wait( lk, [ this ] { return has_job || quit; } );
                    if( quit ) return;
                    j = job; has_job = false;
                }
                j();
                { std::lock_guard< std::mutex > lk( m ); done = true; }
                cv.notify_all();
            }
        } );
    }
    void post( std::function< void() > j )
    {
        { std::lock_guard< std::mutex > lk( m ); job = std::move( j ); has_job = true; done = false; }
        cv.notify_all();
    }
    void wait() { std::unique_lock< std::mutex > lk( m ); cv.wait( lk, [ this ] { return done; } ); }
    void stop()
    {
        if( !thread.joinable() ) return;
        { std::lock_guard< std::mutex > lk( m ); quit = true; }
        cv.notify_all();
        thread.join();
    }
};

/* Tunables, read from the environment ONCE per acn_scene_upload (never from the render path: getenv there would race a
 * host that changes its environment, and would let a value change between the concurrent lanes of one call) */
struct Tunables
{
    size_t   workspace_mb = 0;         /* ACN_WORKSPACE_MB: upper bound of the queue workspace of one handle (all its lanes); 0: 64 GiB or a
                                          quarter of the device memory that is free at upload, whichever is less */
    size_t   chunk = 0;                /* ACN_CHUNK: sample positions per pipeline run, 0 = derived from the queue capacity */
    int      lanes = 6;                /* ACN_LANES: concurrent pipeline runs of a large call (4 until round 4, each on grids twice the size: create_lane) */
    unsigned grid = 0;                 /* ACN_GRID: workgroups of the persistent kernels, 0 = 4 per compute unit */
    unsigned shade_grid = 0;           /* ACN_SHADE_GRID: workgroups of k_shade, 0 = 4 per compute unit */
    unsigned walk_grid = 0;            /* ACN_WALK_GRID: workgroups of k_walk (256 VGPRs: two of its waves fill a SIMD's register file), 0 = as ACN_GRID */
    uint32_t stack_cap = 512;          /* ACN_STACK_CAP: private ray slots per k_walk wave */
    uint32_t fetch_walk = 64;          /* ACN_FETCH_WALK: fresh rays a k_walk wave reserves per cursor atomic */
    uint32_t walk_passes = 4;          /* ACN_WALK_PASSES: launches of k_walk per path level (the last one finishes whatever is left on the waves' private
                                          stacks).  12 until round 4: with k_walk at 4 waves per SIMD the private tail is cheap and the launches are not --
                                          1080p 52.0 -> 50.3 ms, the 1/8 share 13.7 -> 12.1, c2 28.2 -> 26.3, paraffin_lamp 367 -> 339 (profiles/r04/ab_walk_passes_*) */
    uint32_t private_limit = 32768;    /* ACN_PRIVATE_LIMIT: a generation of at most this many rays is finished on private stacks */
    bool     private_limit_set = false; /* ... given by the environment: then it holds for chunks of every size (render_chunk) */
    uint32_t class0_min = 0;           /* ACN_CLASS0_MIN: shading tasks with more samples than this take the 64-lane kernel, the others 16 / 4 / 1 lanes; 0: chosen per scene (acn_scene_upload) */
    uint32_t fetch_shade = 16;         /* ACN_FETCH_SHADE: steps ( of 64 / lanes-per-task tasks ) a k_shade wave reserves per cursor atomic */
    uint32_t fetch_hard = 256;         /* ACN_FETCH_HARD: records a wave of the hard-ray kernels / k_shade_hits reserves per atomic */
    uint32_t stack_use = 0;            /* ACN_TEST_STACK_USE: slots of a private stack every walk pass but the last uses (tests of the overflow path) */
    bool     debug_chunks = false;     /* ACN_DEBUG_CHUNKS=1: one line per chunk on stderr (size, queue marks, rates, capacities) */
    bool     ws_uniform = false;       /* ACN_WS_UNIFORM=1: every queue gets the same share of the whole bound at once (round 2's layout; diagnostic) */
    int      learn_grids = 0;          /* ACN_LEARN_GRIDS: 0 every launch of a chain gets the full persistent grid; 1 the grids follow the input of the
                                        * last chunk (learned_grid); 2 only launches whose input was empty then get a small grid.  Measured, off: see learned_grid */
    double   grid_passes = 1.0;        /* ACN_GRID_PASSES: a learned grid gives a workgroup this many workgroup-loads of the input it expects (x 1/2: head room) */
    bool     learn_passes = true;      /* ACN_LEARN_PASSES=0: every level gets ACN_WALK_PASSES launches of k_walk, needed or not */
    bool     shade_fission = false;    /* ACN_SHADE_FISSION=1: the two sample loops of a shading point run as two launches, the direct-light half and its
                                          deferred shadow rays on a side stream (render_chunk).  Measured and OFF: 1080p 52.0 -> 56.7 ms, the 1/8 share
                                          13.7 -> 17 - 20 ms, c2 28.0 -> 31.9, paraffin_lamp 365 -> 400 - 450 (profiles/r04/ab_fission.txt): every lane then
                                          has two streams of persistent grids, and eight grids of 512 - 1024 workgroups take turns on one chip */
    bool     learn_sample = true;      /* ACN_LEARN_SAMPLE=0: no strided learning pass on a cold handle (learn_rates): the first chunks learn, as in round 3 */
    bool     cold_pipeline = true;     /* ACN_COLD_PIPELINE=0: a cold handle makes its lanes before the learning pass, not beside it (render_lanes) */
    bool     early_lanes = false;      /* ACN_EARLY_LANES=1: the lanes a whole frame of the scene's own raster will use are made during acn_scene_upload (a
                                          helper thread beside the upload's own work, while the device is idle), not by the first call that needs them.
                                          Measured and OFF (profiles/r04/ab_early_lanes_s42.txt): the streams cost the same ~10 ms each wherever they are made
                                          and do not overlap the handle's own first stream, so the upload grows by 50 - 90 ms while the first frame loses
                                          20 - 100 (1080p 114 - 174 -> 72 - 75 ms, c2 94 - 135 -> 48 - 49, paraffin_lamp 486 - 504 -> 466 - 479, hanging_lamp
                                          417 - 426 -> 390 - 401); upload + first frame: 1080p 247 - 299 -> 277 - 323 ms, c2 189 - 261 -> 195 - 204, the
                                          lamps +30.  For a host that uploads long before it renders */
    bool     count_work = false;       /* ACN_COUNT_WORK */
    bool     stage_timing = false;     /* ACN_STAGE_TIMING */
    void read()
    {
        if( const char* e = getenv( "ACN_WORKSPACE_MB" ) ) workspace_mb = ( size_t )atoll( e );
        if( const char* e = getenv( "ACN_CHUNK" ) ) chunk = ( size_t )atoll( e );
        if( const char* e = getenv( "ACN_LANES" ) ) lanes = atoi( e );
        if( const char* e = getenv( "ACN_GRID" ) ) grid = ( unsigned )atoi( e );
        if( const char* e = getenv( "ACN_SHADE_GRID" ) ) shade_grid = ( unsigned )atoi( e );
        if( const char* e = getenv( "ACN_WALK_GRID" ) ) walk_grid = ( unsigned )atoi( e );
        if( const char* e = getenv( "ACN_STACK_CAP" ) ) stack_cap = ( uint32_t )atoll( e );
        if( const char* e = getenv( "ACN_TEST_STACK_USE" ) ) stack_use = ( uint32_t )atoll( e );
        if( const char* e = getenv( "ACN_FETCH_WALK" ) ) fetch_walk = ( uint32_t )atoll( e );
        if( const char* e = getenv( "ACN_FETCH_HARD" ) ) fetch_hard = ( uint32_t )atoll( e );
        if( const char* e = getenv( "ACN_FETCH_SHADE" ) ) fetch_shade = ( uint32_t )atoll( e );
        if( const char* e = getenv( "ACN_CLASS0_MIN" ) ) class0_min = ( uint32_t )atoll( e );
        if( fetch_shade < 1 ) fetch_shade = 1;
        if( const char* e = getenv( "ACN_WALK_PASSES" ) ) walk_passes = ( uint32_t )atoll( e );
        if( const char* e = getenv( "ACN_PRIVATE_LIMIT" ) ) { private_limit = ( uint32_t )atoll( e ); private_limit_set = true; }
        if( walk_passes < 1 ) walk_passes = 1;
        if( walk_passes > ACN_MAX_WALK_PASSES ) walk_passes = ACN_MAX_WALK_PASSES;
        if( fetch_walk < 64 ) fetch_walk = 64;
        if( fetch_hard < 64 ) fetch_hard = 64;
        count_work = getenv( "ACN_COUNT_WORK" ) != nullptr;
        if( const char* e = getenv( "ACN_LEARN_PASSES" ) ) learn_passes = atoi( e ) != 0;
        if( const char* e = getenv( "ACN_LEARN_GRIDS" ) ) learn_grids = atoi( e );
        if( const char* e = getenv( "ACN_LEARN_SAMPLE" ) ) learn_sample = atoi( e ) != 0;
        if( const char* e = getenv( "ACN_COLD_PIPELINE" ) ) cold_pipeline = atoi( e ) != 0;
        if( const char* e = getenv( "ACN_EARLY_LANES" ) ) early_lanes = atoi( e ) != 0;
        if( const char* e = getenv( "ACN_SHADE_FISSION" ) ) shade_fission = atoi( e ) != 0;
        if( const char* e = getenv( "ACN_WS_UNIFORM" ) ) ws_uniform = atoi( e ) != 0;
        debug_chunks = getenv( "ACN_DEBUG_CHUNKS" ) != nullptr;
        if( const char* e = getenv( "ACN_GRID_PASSES" ) ) { grid_passes = atof( e ); if( !( grid_passes >= 0.25 && grid_passes <= 64.0 ) ) grid_passes = 1.0; }
        stage_timing = getenv( "ACN_STAGE_TIMING" ) != nullptr;
        if( lanes < 1 ) lanes = 1;
        if( lanes > 16 ) lanes = 16;
        if( stack_cap < 256 ) stack_cap = 256;
        if( stack_use == 0 || stack_use > stack_cap ) stack_use = stack_cap;
    }
};

/* the queue workspace of one pipeline run */
struct Workspace
{
    DTask*      tasks = nullptr;
    uint32_t*   idx[ ACN_NCLASS ] = { nullptr, nullptr, nullptr, nullptr };
    HitRec*     children = nullptr;
    HardShadow* hard_shadow = nullptr;
    HardPath*   hard_path = nullptr;
    RayTask*    rays[ 2 ] = { nullptr, nullptr };
    RayTask*    stacks = nullptr;   size_t stack_waves = 0;
    uint32_t    cap[ 5 ] = { 0, 0, 0, 0, 0 };   /* records per queue, WQ_* */
    size_t      bytes = 0;          /* device memory of the queues and stacks */
    uint64_t    allocs = 0;         /* times this workspace was (re)allocated */
    bool        trimmed = false;    /* it was already re-allocated smaller once */
    uint32_t    sized_calls = 0;    /* calls of ensure_workspace with learned rates (the trim window, see there) */
};
/* the queues of a pipeline run.  Each is sized from its OWN demand per sample position (learned, below): on the wine glass a
 * position leaves 15 deferred shadow rays but 2 shading points, and one common capacity -- the former layout -- made every
 * queue as large as the fullest one needs (64 GiB for a 1080p frame of which 7 % were used). */
enum { WQ_TASKS = 0, WQ_CHILDREN, WQ_HARD_SHADOW, WQ_HARD_PATH, WQ_RAYS, WQ_N };
static const size_t wq_bytes[ WQ_N ] = { sizeof( DTask ) + ACN_NCLASS * sizeof( uint32_t ), sizeof( HitRec ), sizeof( HardShadow ), sizeof( HardPath ), 2 * sizeof( RayTask ) };

struct acn_scene_handle
{
    int device = 0;
    DevScene dev{};
    GNode*   d_nodes = nullptr;
    GMat*    d_mats = nullptr;
    int32_t* d_elems = nullptr;
    acn_texture* d_textures = nullptr;
    SCEntry* d_sc_table = nullptr;
    double* d_sc_spheres = nullptr;            /* ( pos, radius ) of the sphere leaves of d_sc_table */
    double* d_env_tab = nullptr;               /* envelopes of the compound slices, element by element (acn_device.h: root_candidates) */
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;         /* the direct-light half of a fissioned level and its deferred shadow rays (render_chunk) */
    hipEvent_t ev_fork = nullptr, ev_path = nullptr, ev_join = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int cur_stage = 0;
    bool stage_timing = false;                 /* ACN_OPT_STAGE_TIMING of the current call */
    int max_csg_depth = 0;
    Tunables tun;
    unsigned cus = 256;                        /* compute units of the device */
    unsigned grid = 1024, shade_grid = 1024;   /* workgroups of the persistent kernels / of k_shade */
    unsigned walk_grid = 1024;                 /* ... of k_walk */
    int n_levels = 1;                          /* path levels of the scene's trace_depth */
    size_t n_lights = 1;                       /* elements of the light root */
    /* workspace of the wavefront pipeline */
    Workspace ws;
    uint32_t* d_counts = nullptr;              /* ACN_MAX_PATH_LEVELS + 1 counter blocks of QC_N words */
    uint32_t* h_counts = nullptr;              /* pinned copy */
    unsigned long long* d_accum = nullptr;  size_t accum_cap = 0;
    unsigned long long* d_counters = nullptr;
    std::vector< StageEvents > events;  size_t events_used = 0;
    size_t lds_bytes = 0;                      /* > 0: the node array fits the LDS staging budget */
    size_t lds_stack_bytes = 0;                /* > 0: the machine kernels keep their CSG stacks in LDS */
    bool prune = false;                        /* some root element has an interval-prune program: launch the PRUNE kernel variants */
    bool leaf_lights = true;                   /* every light element is a plane / sphere */
    bool count_work = false;                   /* ACN_OPT_COUNT_WORK of the current call */
    uint32_t shard_rank = 0, shard_world = 1;  /* ACN_SHARD_SAMPLES of the current call */
    uint64_t launches[ 4 ] = { 0, 0, 0, 0 };   /* walk, shade, finalize, hard-ray kernels */
    uint64_t hard_rays = 0, walk_steps = 0, walk_rays = 0, shade_hit_recs = 0, host_syncs = 0, private_rays = 0, probe_rays = 0;
    uint32_t flags_seen = 0;                   /* ACN_FLAG_* bits of the last call */
    uint32_t rate_cnt = 0;                     /* positions of the chunk the rates were taken from */
    acn_chunk_ctl ctl = { 0.7, 0, 0 };         /* acn_chunkplan.h.  fill_target: fraction of its capacity the fullest queue of a chunk is
                                                  planned to reach: lowered by every overflow (a redone chunk is lost work), raised slowly
                                                  by chunks that fit */
    double rate[ 5 ] = { 0, 0, 0, 0, 0 };      /* learned: records per sample position a chunk leaves in each queue (WQ_*); 0: not known yet */
    size_t workspace_budget = 0;               /* bytes this handle's queues may take (all lanes together) */
    uint64_t chunks = 0, retries = 0, levels = 0;
    uint64_t peak_tasks = 0, peak_children = 0;
    /* learned: the input of every launch of the last chunk's chain and the positions of that chunk ( 0: nothing known ) */
    uint32_t seen_cnt = 0;
    uint32_t seen_class[ ACN_MAX_PATH_LEVELS + 1 ][ ACN_NCLASS ] = {};
    uint32_t seen_hs[ ACN_MAX_PATH_LEVELS + 1 ] = {}, seen_hp[ ACN_MAX_PATH_LEVELS + 1 ] = {}, seen_hits[ ACN_MAX_PATH_LEVELS + 1 ] = {};
    uint32_t seen_gen[ ACN_MAX_PATH_LEVELS + 1 ][ ACN_MAX_WALK_PASSES + 2 ] = {};
    uint32_t walk_passes_seen[ ACN_MAX_PATH_LEVELS + 1 ] = { 0, 0, 0, 0, 0, 0 };   /* learned: passes of a level that had input in the last chunk (0: not known yet) */
    unsigned long long* d_counters_keep = nullptr;   /* the work counters as they were before the current chunk (restored when it is redone) */
    /* concurrent lanes (render_lanes): clones of this handle that share the resident scene and own a stream and a
     * workspace each */
    bool is_lane = false;
    size_t budget_div = 1;                     /* workspace budget of a lane = the handle's budget / lanes */
    std::vector< acn_scene_handle* > lanes;
    /* lanes made during acn_scene_upload on a helper thread (early_lanes_begin), taken over by the first call that runs on lanes */
    std::thread early_maker;
    std::vector< acn_scene_handle* > early_made;
    int early_status = 0; std::string early_message;
    LaneWorker* worker = nullptr;              /* of a lane */
    size_t scene_bytes[ 4 ] = { 0, 0, 0, 0 };
    double* d_lane_pos = nullptr; double* d_lane_out = nullptr; size_t lane_buf_cap = 0;   /* a lane's gathered positions / results */
    double* d_shard_pos = nullptr; size_t shard_pos_cap = 0;                                /* acn_render_main_pass_shard_dev: the rank's positions */
    std::string lane_error;
    bool used_lanes = false;                   /* the last render call ran through the lanes: statistics are their sums */
    int  lanes_used = 0;                       /* ... the first lanes_used of them */
    bool one_lane = false;                     /* the last call would have used lanes but did not fit the workspace bound that way */
};
#define ACN_LEVEL_BLOCKS ( ACN_MAX_PATH_LEVELS + 1 )

/* ------------------------------------------------------------------------------------------------------------------ */
/* kernels */

/* camera basis with the oracle's expressions (scene.c:963-973), one lane */
__global__ void k_camera_setup( DevScene sc, M3* out_rot, double* out_unit_f )
{
    uint64_t unit_sz = ( sc.prm.image_height >> 1 );
    *out_unit_f = 1.0 / unit_sz;
    V3 ry = v_of_length( ld3( sc.prm.camera_view_direction ), 1 );
    V3 rz = v_of_length( ld3( sc.prm.camera_top_direction ), 1 );
    rz = v_von( ry, rz );
    V3 rx = v_mlx( ry, rz );
    M3 r; r.x = rx; r.y = ry; r.z = rz;
    *out_rot = m_transposed( r );
}

/* fixed point -> f64 (+ optional cl_s_sat) for positions [ base, base + n ) */
__global__ void k_finalize( const unsigned long long* __restrict__ accum, uint32_t n, double gamma, int linear,
                            double* __restrict__ out_rgb )
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n ) return;
    V3 c = mk( ( double )( long long )accum[ ( size_t )i * 3 + 0 ] * ACN_FIX_INV,
               ( double )( long long )accum[ ( size_t )i * 3 + 1 ] * ACN_FIX_INV,
               ( double )( long long )accum[ ( size_t )i * 3 + 2 ] * ACN_FIX_INV );
    if( !linear ) c = cl_sat( c, gamma );
    out_rgb[ ( size_t )i * 3 + 0 ] = c.x;
    out_rgb[ ( size_t )i * 3 + 1 ] = c.y;
    out_rgb[ ( size_t )i * 3 + 2 ] = c.z;
}

/* the pixel sums of the positions in slots [ base, base + cnt ) start over (a chunk is redone after a queue overflow) */
__global__ void k_clear_slots( unsigned long long* __restrict__ accum, uint32_t base, uint32_t cnt, TileOrder order )
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= cnt ) return;
    uint32_t p = order.position( base + i );
    if( p >= order.n ) return;
    accum[ ( size_t )p * 3 + 0 ] = 0; accum[ ( size_t )p * 3 + 1 ] = 0; accum[ ( size_t )p * 3 + 2 ] = 0;
}

/* cl_s_sat + cps_from_cl after the (cross-GPU) accumulation */
__global__ void k_resolve( const double* __restrict__ lin, size_t n, double gamma, double* __restrict__ out_rgb,
                           unsigned char* __restrict__ out_rgb8 )
{
    size_t i = ( size_t )blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n ) return;
    V3 c = cl_sat( mk( lin[ i * 3 ], lin[ i * 3 + 1 ], lin[ i * 3 + 2 ] ), gamma );
    if( out_rgb ) { out_rgb[ i * 3 ] = c.x; out_rgb[ i * 3 + 1 ] = c.y; out_rgb[ i * 3 + 2 ] = c.z; }
    if( out_rgb8 )
    {
        out_rgb8[ i * 3 + 0 ] = c.x > 0.0 ? c.x < 1.0 ? ( unsigned char )( c.x * 256 ) : 255 : 0;
        out_rgb8[ i * 3 + 1 ] = c.y > 0.0 ? c.y < 1.0 ? ( unsigned char )( c.y * 256 ) : 255 : 0;
        out_rgb8[ i * 3 + 2 ] = c.z > 0.0 ? c.z < 1.0 ? ( unsigned char )( c.z * 256 ) : 255 : 0;
    }
}

/* obj_ray_exit + obj_estimate_envelope (objects.c:286-363), one lane */
__global__ void k_estimate_envelope( DevScene sc, int node, uint64_t samples, uint32_t rseed, double radius_factor,
                                     V3* scratch, double* out )
{
    Cnt< false > cnt;
    NodeP hdr = &sc.nodes[ node ];
    uint64_t size = 0;
    V3 sum = mk( 0, 0, 0 );
    uint64_t rv = rseed;
    V3 rp = ld3( hdr->pos );
    for( uint64_t i = 0; i < samples; i++ )
    {
        V3 rd = v_random_sphere_belt( &rv, 1.0 );
        /* obj_ray_exit */
        double exit_a = F3_INF;
        {
            V3 nor = mk( 0, 0, 0 );
            double a = obj_ray_hit_dev( sref( sc ), node, rp, rd, true, &nor, &cnt );
            if( a < F3_INF )
            {
                V3 lp = rp;
                double s = 0;
                while( a < F3_INF )
                {
                    a += F3_EPS * 2;
                    s += a;
                    lp = ray_pos( lp, rd, a );
                    a = obj_ray_hit_dev( sref( sc ), node, lp, rd, true, &nor, &cnt );
                }
                if( v_mlv( nor, rd ) > 0 ) exit_a = s;
            }
        }
        if( exit_a < F3_INF )
        {
            V3 pos = ray_pos( rp, rd, exit_a );
            scratch[ size++ ] = pos;
            sum = v_add( sum, ray_pos( rp, rd, exit_a ) );
            rp = v_mlf( sum, ( 1.0 / size ) );
            rp.x += F3_EPS * f3_rnd0( &rv );
            rp.y += F3_EPS * f3_rnd0( &rv );
            rp.z += F3_EPS * f3_rnd0( &rv );
        }
    }
    double radius = F3_MAG;
    if( size > 0 )
    {
        double max_r2 = 0;
        for( uint64_t i = 0; i < size; i++ )
        {
            double r = v_diff_sqr( rp, scratch[ i ] );
            max_r2 = r > max_r2 ? r : max_r2;
        }
        radius = acn_sqrt( max_r2 ) * radius_factor;
    }
    out[ 0 ] = rp.x; out[ 1 ] = rp.y; out[ 2 ] = rp.z; out[ 3 ] = radius;
}

__global__ void k_detmath( int op, const double* x, const double* y, double* out, size_t n )
{
    size_t i = ( size_t )blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n ) return;
    double a = x[ i ], b = y ? y[ i ] : 0.0, r = 0;
    switch( op )
    {
        case 0: r = acn_sin( a ); break;
        case 1: r = acn_cos( a ); break;
        case 2: r = acn_tan( a ); break;
        case 3: r = acn_acos( a ); break;
        case 4: r = acn_log( a ); break;
        case 5: r = acn_exp( a ); break;
        case 6: r = acn_pow( a, b ); break;
        case 7: r = acn_sqrt( a ); break;
        case 8: r = a / b; break;
        case 9: r = ( double )acn_f64_bits( a ); break;
        case 10: r = acn_frexp_mant( a ); break;
        default: break;
    }
    out[ i ] = r;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* validation: what the reference would abort on, plus the device limits */
static int csg_depth( const acn_flat_scene* sc, int node, int d, std::string& err )
{
    if( d > 4096 ) { err = "cyclic node graph"; return -1; }
    const acn_node* n = &sc->nodes[ node ];
    int m = 0;
    switch( n->type )
    {
        case ACN_PAIR_INSIDE: case ACN_PAIR_OUTSIDE:
        {
            int a = csg_depth( sc, n->child0, d + 1, err ), b = csg_depth( sc, n->child1, d + 1, err );
            if( a < 0 || b < 0 ) return -1;
            m = 1 + ( a > b ? a : b );
            break;
        }
        case ACN_NEG: case ACN_SCALE:
        {
            int a = csg_depth( sc, n->child0, d + 1, err );
            if( a < 0 ) return -1;
            m = 1 + a;
            break;
        }
        default: break;
    }
    return m;
}

static int compound_depth( const acn_flat_scene* sc, int node, int d, int* max_csg, std::string& err )
{
    if( d > 256 ) { err = "compound nesting too deep / cyclic"; return -1; }
    const acn_node* n = &sc->nodes[ node ];
    int m = 1;
    for( int k = 0; k < n->child1; k++ )
    {
        int e = sc->elems[ n->child0 + k ];
        if( sc->nodes[ e ].type == ACN_COMPOUND )
        {
            int c = compound_depth( sc, e, d + 1, max_csg, err );
            if( c < 0 ) return -1;
            if( c + 1 > m ) m = c + 1;
        }
        else
        {
            int c = csg_depth( sc, e, 0, err );
            if( c < 0 ) return -1;
            if( c > *max_csg ) *max_csg = c;
        }
    }
    return m;
}

static int validate( const acn_flat_scene* sc, int* max_csg )
{
    if( !sc || !sc->nodes ) return fail( ACN_ERR_ARG, "null scene" );
    if( sc->abi_version != ACN_ABI_VERSION ) return fail( ACN_ERR_ARG, "abi_version mismatch" );
    if( sc->n_nodes == 0 || sc->light_root < 0 || sc->matter_root < 0 || ( uint32_t )sc->light_root >= sc->n_nodes ||
        ( uint32_t )sc->matter_root >= sc->n_nodes ) return fail( ACN_ERR_ARG, "bad root index" );
    if( sc->n_elems && !sc->elems ) return fail( ACN_ERR_ARG, "null elems" );
    if( sc->params.experimental_level != 0 ) return fail( ACN_ERR_UNSUPPORTED, "Unsupported experimental level" );   /* scene.c:1004-1007 */
    if( sc->params.image_height < 2 || sc->params.image_width < 1 ) return fail( ACN_ERR_ARG, "image size" );
    if( sc->params.trace_depth > 10 * ACN_MAX_PATH_LEVELS + 10 ) return fail( ACN_ERR_UNSUPPORTED, "trace_depth exceeds device path-level limit" );
    for( uint32_t i = 0; i < sc->n_nodes; i++ )
    {
        const acn_node* n = &sc->nodes[ i ];
        if( n->texture != -1 )
        {
            if( n->texture < 0 || ( uint32_t )n->texture >= sc->n_textures || !sc->textures ) return fail( ACN_ERR_ARG, "bad texture index" );
            const acn_texture* t = &sc->textures[ n->texture ];
            if( t->kind != ACN_TXM_PLAIN && t->kind != ACN_TXM_CHESS ) return fail( ACN_ERR_ARG, "unknown texture kind" );
            if( t->kind == ACN_TXM_CHESS && n->type != ACN_PLANE && n->type != ACN_SPHERE && n->type != ACN_DISTANCE )
                return fail( ACN_ERR_UNSUPPORTED, "object has no projection-function for a chess texture (objects.c:240-245)" );
        }
        switch( n->type )
        {
            case ACN_PLANE: case ACN_SPHERE: case ACN_SQUAROID: break;
            case ACN_DISTANCE:
                if( n->sdf_kind != ACN_SDF_SPHERE && n->sdf_kind != ACN_SDF_TORUS ) return fail( ACN_ERR_UNSUPPORTED, "unknown distance function" );
                break;
            case ACN_PAIR_INSIDE: case ACN_PAIR_OUTSIDE:
                if( n->child1 < 0 || ( uint32_t )n->child1 >= sc->n_nodes || sc->nodes[ n->child1 ].type == ACN_COMPOUND ) return fail( ACN_ERR_ARG, "bad pair child" );
                /* fallthrough */
            case ACN_NEG: case ACN_SCALE:
                if( n->child0 < 0 || ( uint32_t )n->child0 >= sc->n_nodes || sc->nodes[ n->child0 ].type == ACN_COMPOUND ) return fail( ACN_ERR_ARG, "bad child" );
                break;
            case ACN_COMPOUND:
                if( n->child1 < 0 || n->child0 < 0 || ( uint64_t )n->child0 + ( uint64_t )n->child1 > sc->n_elems ) return fail( ACN_ERR_ARG, "bad compound slice" );
                for( int k = 0; k < n->child1; k++ )
                {
                    int e = sc->elems[ n->child0 + k ];
                    if( e < 0 || ( uint32_t )e >= sc->n_nodes ) return fail( ACN_ERR_ARG, "bad element index" );
                }
                break;
            default: return fail( ACN_ERR_ARG, "unknown node type" );
        }
    }
    const acn_node* light = &sc->nodes[ sc->light_root ];
    if( light->type != ACN_COMPOUND || sc->nodes[ sc->matter_root ].type != ACN_COMPOUND ) return fail( ACN_ERR_ARG, "roots must be compounds" );
    for( int k = 0; k < light->child1; k++ )
    {
        int t = sc->nodes[ sc->elems[ light->child0 + k ] ].type;
        if( t == ACN_COMPOUND ) return fail( ACN_ERR_ARG, "light elements must be objects (scene.c:547)" );
        if( t != ACN_PLANE && t != ACN_SPHERE && t != ACN_PAIR_INSIDE && t != ACN_PAIR_OUTSIDE )
            return fail( ACN_ERR_NO_FOV, "light object has no fov-function (objects.c:254-258)" );
    }
    std::string err;
    *max_csg = 0;
    int dl = compound_depth( sc, sc->light_root, 0, max_csg, err );
    int dm = dl < 0 ? -1 : compound_depth( sc, sc->matter_root, 0, max_csg, err );
    if( dl < 0 || dm < 0 ) return fail( ACN_ERR_ARG, err );
    if( dm > ACN_CMP_MAX_DEPTH || dl > ACN_CMP_MAX_DEPTH ) return fail( ACN_ERR_UNSUPPORTED, "compound nesting exceeds device limit" );
    if( *max_csg > ACN_CSG_MAX_DEPTH ) return fail( ACN_ERR_UNSUPPORTED, "CSG nesting exceeds device limit" );
    return ACN_OK;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* ABI */
extern "C" int acn_device_count( void )
{
    int n = 0;
    if( hipGetDeviceCount( &n ) != hipSuccess ) return 0;
    return n;
}

static int lane_objects( int device, bool side_stream, bool debug, acn_scene_handle** out );
static int lanes_for_counts( int tun_lanes, size_t n, uint64_t path_samples );

/* Lanes made during the upload.  A stream that gets its own hardware queue costs ~10 ms of host time, and making one while kernels
 * run stretches those kernels too (the learning pass of a cold handle: 19 ms alone, 56 - 150 ms beside six streams being made,
 * profiles/r04/upload_timeline_s41.txt) -- so the lanes a whole frame of the scene's own raster will use are made here, on a helper
 * thread beside the upload's host work and copies, while nothing of this handle runs on the device (ACN_EARLY_LANES=1; off by
 * default: the runtime makes streams one after the other, so the time only moves from the first call into the upload).  A handle that would render its
 * raster on one lane (small rasters, path_samples >= 256 on a cold handle: render_positions) makes none.  Failures are not reported
 * from here: the first call that needs the lanes makes what is missing and reports its own. */
static void early_lanes_begin( acn_scene_handle* h, size_t n, uint64_t path_samples )
{
    const int lanes = lanes_for_counts( h->tun.lanes, n, path_samples );
    if( lanes <= 1 || path_samples >= 256 ) return;
    const int device = h->device; const bool side = h->tun.shade_fission, debug = h->tun.debug_chunks;
    h->early_maker = std::thread( [ h, lanes, device, side, debug ]()
    {
        for( int k = 0; k < lanes; k++ )
        {
            acn_scene_handle* l = nullptr;
            if( lane_objects( device, side, debug, &l ) != ACN_OK ) break;
            h->early_made.push_back( l );
        }
    } );
}
static void early_lanes_join( acn_scene_handle* h ) { if( h->early_maker.joinable() ) h->early_maker.join(); }

extern "C" int acn_scene_upload( const acn_flat_scene* scene, int device, acn_scene_handle** out )
{
    if( !out ) return fail( ACN_ERR_ARG, "null out" );
    *out = nullptr;
    int max_csg = 0;
    int st = validate( scene, &max_csg );
    if( st != ACN_OK ) return st;
    int ndev = acn_device_count();
    if( ndev <= 0 ) return fail( ACN_ERR_DEVICE, "no HIP device (libactinon_hip has no CPU fallback)" );
    if( device < 0 || device >= ndev ) return fail( ACN_ERR_ARG, "bad device index" );
    HIP_TRY( hipSetDevice( device ) );
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [ & ]() { return std::chrono::duration< double, std::milli >( std::chrono::steady_clock::now() - t_begin ).count(); };
    acn_scene_handle* h = new acn_scene_handle();
    h->device = device;
    h->max_csg_depth = max_csg;
    h->tun.read();
    if( h->tun.early_lanes ) early_lanes_begin( h, ( size_t )scene->params.image_width * ( size_t )scene->params.image_height, scene->params.path_samples );
    double t_up[ 4 ] = { 0, 0, 0, 0 };   /* ACN_DEBUG_CHUNKS: stream + events, host-side tables, device copies, the camera kernel */
    {
        /* Workspace BOUND of the handle: ACN_WORKSPACE_MB, or 64 GiB / a quarter of the free device memory (288 GB per
         * MI355X).  It is a bound, not an allocation: the queues are sized from measured demand (ensure_workspace) and take
         * what ONE chunk per lane needs, if the bound allows -- every further chunk of a lane is another chain of ~45
         * dependent launches (1080p wine_glass, 4 lanes: 20 GB and 71 ms with one chunk per lane; bound 8 GiB: 9 chunks, 91 ms;
         * 4 GiB: 18 chunks, 125 ms).  Scenes whose demand per position is huge (path_samples 256 .. 1024: thousands of
         * second-level hits per pixel) use the whole bound: many_spheres p256 at 24 GiB 532 chunks, 39 s; at 64 GiB ... */
        size_t free_b = 0, total_b = 0;
        if( hipMemGetInfo( &free_b, &total_b ) != hipSuccess ) free_b = ( size_t )32 << 30;
        h->workspace_budget = h->tun.workspace_mb ? h->tun.workspace_mb * 1024 * 1024 : ( ( size_t )64 << 30 );
        if( !h->tun.workspace_mb && h->workspace_budget > free_b / 4 ) h->workspace_budget = free_b / 4;
    }
    {
        int cus = 0;
        if( hipDeviceGetAttribute( &cus, hipDeviceAttributeMultiprocessorCount, device ) != hipSuccess || cus <= 0 ) cus = 256;
        /* persistent grids.  A call that runs alone on its stream: 4 workgroups of 256 lanes per CU, what fits of the
         * 128-VGPR kernels.  The concurrent lanes of a call (create_lane): 2 per CU each -- what is resident of k_walk
         * (256 VGPRs); four lanes keep the chip full and leave room for each other's kernels (1080p: 83.4 ms against
         * 85.2 with 4 per CU; a lone lane with 2 per CU: 106 ms against 88) */
        h->cus = ( unsigned )cus;
        h->grid = h->tun.grid ? h->tun.grid : ( unsigned )cus * 4u;
        h->shade_grid = h->tun.shade_grid ? h->tun.shade_grid : ( unsigned )cus * 4u;
        h->walk_grid = h->tun.walk_grid ? h->tun.walk_grid : h->grid;
        /* path levels: level L shades hits at depth trace_depth - 10 L and spawns the next one while that is > 10 (scene.c:584) */
        uint64_t td = scene->params.trace_depth;
        h->n_levels = scene->params.path_samples && td > 10 ? 1 + ( int )( ( td - 10 + 9 ) / 10 ) : 1;
    }
    auto bail = [ & ]( int code ) { acn_scene_free( h ); return code; };
#define HIP_TRY_H( expr ) do { hipError_t e_ = ( expr ); if( e_ != hipSuccess ) \
    return bail( fail( ACN_ERR_DEVICE, std::string( #expr ) + ": " + hipGetErrorString( e_ ) ) ); } while( 0 )
    const double t_stream0 = since();
    HIP_TRY_H( hipStreamCreate( &h->stream ) );
    const double t_stream1 = since();   /* (the first stream a process makes: 85 - 100 ms on this runtime; later ones ~10) */
    /* (a stream costs ~10 ms of host time to make: the second one only where it is used) */
    if( h->tun.shade_fission ) HIP_TRY_H( hipStreamCreateWithFlags( &h->side_stream, hipStreamNonBlocking ) );
    HIP_TRY_H( hipEventCreate( &h->ev0 ) );
    HIP_TRY_H( hipEventCreate( &h->ev1 ) );
    HIP_TRY_H( hipEventCreateWithFlags( &h->ev_fork, hipEventDisableTiming ) );
    HIP_TRY_H( hipEventCreateWithFlags( &h->ev_path, hipEventDisableTiming ) );
    HIP_TRY_H( hipEventCreateWithFlags( &h->ev_join, hipEventDisableTiming ) );

    /* ABI layout -> device layout: geometry (GNode) and shading properties (GMat) split */
    std::vector< GNode > nodes( scene->n_nodes );
    std::vector< GMat > mats( scene->n_nodes );
    for( uint32_t i = 0; i < scene->n_nodes; i++ )
    {
        const acn_node& a = scene->nodes[ i ];
        GNode& g = nodes[ i ];
        memset( &g, 0, sizeof( g ) );
        g.type = a.type; g.flags = a.flags; g.child0 = a.child0; g.child1 = a.child1;
        if( ( a.type == ACN_PAIR_INSIDE || a.type == ACN_PAIR_OUTSIDE ) && !getenv( "ACN_NO_LEAF_PAIRS" ) )
        {
            auto simple = [ & ]( int32_t c )
            {
                const acn_node* x = &scene->nodes[ c ];
                if( x->type == ACN_NEG ) x = &scene->nodes[ x->child0 ];
                return x->type == ACN_PLANE || x->type == ACN_SPHERE || x->type == ACN_SQUAROID;
            };
            auto level1 = [ & ]( int32_t c )
            {
                const acn_node& x = scene->nodes[ c ];
                return ( x.type == ACN_PAIR_INSIDE || x.type == ACN_PAIR_OUTSIDE ) && simple( x.child0 ) && simple( x.child1 );
            };
            if( simple( a.child0 ) && simple( a.child1 ) ) g.flags |= ACN_GFLAG_LEAF_PAIR;
            else if( ( simple( a.child0 ) || level1( a.child0 ) ) && ( simple( a.child1 ) || level1( a.child1 ) ) && !getenv( "ACN_NO_PAIR2" ) ) g.flags |= ACN_GFLAG_PAIR2;
        }
        memcpy( g.prm, a.prm, sizeof( g.prm ) );
        memcpy( g.pos, a.pos, sizeof( g.pos ) );
        memcpy( g.env_pos, a.env_pos, sizeof( g.env_pos ) );
        g.env_radius = a.env_radius;
        memcpy( g.rax, a.rax, sizeof( g.rax ) );
        g.surface_roughness = a.surface_roughness;
        g.sdf_kind = a.sdf_kind; g.cycles = a.cycles;
        GMat& m = mats[ i ];
        memcpy( m.color, a.color, sizeof( m.color ) );
        m.radiance = a.radiance; m.refractive_index = a.refractive_index;
        m.fresnel_reflectivity = a.fresnel_reflectivity; m.chromatic_reflectivity = a.chromatic_reflectivity;
        m.diffuse_reflectivity = a.diffuse_reflectivity; m.sigma = a.sigma;
        memcpy( m.transparency, a.transparency, sizeof( m.transparency ) );
        m.texture = a.texture; m.pad_ = 0;
    }
    /* surely_outside (acn_device.h) descends a pair tree up to ACN_PRUNE_DEPTH levels to test the envelopes it finds.  How many
     * levels of a node are worth reading is known here: ACN_GFLAG_PRUNE_LEVELS( flags ) = 0 if neither the node nor any pair
     * operand within three levels below it has an envelope, else 1 + the depth of the deepest such envelope -- the descent stops
     * where nothing is left to test instead of reading operands for nothing (a chain of dependent scalar loads per level). */
    if( !getenv( "ACN_NO_PRUNE_LEVELS" ) )
    {
        std::function< int( int32_t, int ) > deepest = [ & ]( int32_t i, int left ) -> int   /* depth of the deepest envelope within `left` levels, -1: none */
        {
            const acn_node& a = scene->nodes[ i ];
            int best = ( a.flags & ACN_NODE_HAS_ENVELOPE ) ? 0 : -1;
            if( left > 0 && ( a.type == ACN_PAIR_INSIDE || a.type == ACN_PAIR_OUTSIDE ) )
                for( int32_t c : { a.child0, a.child1 } ) { int d = deepest( c, left - 1 ); if( d >= 0 && d + 1 > best ) best = d + 1; }
            return best;
        };
        for( uint32_t i = 0; i < scene->n_nodes; i++ ) nodes[ i ].flags |= ( uint32_t )( deepest( ( int32_t )i, 3 ) + 1 ) << ACN_GFLAG_PRUNE_LEVELS_SHIFT;
    }
    else for( uint32_t i = 0; i < scene->n_nodes; i++ ) nodes[ i ].flags |= 4u << ACN_GFLAG_PRUNE_LEVELS_SHIFT;
    h->scene_bytes[ 0 ] = sizeof( GNode ) * scene->n_nodes; h->scene_bytes[ 1 ] = sizeof( GMat ) * scene->n_nodes;
    h->scene_bytes[ 3 ] = sizeof( acn_texture ) * ( scene->n_textures ? scene->n_textures : 1 );
    t_up[ 0 ] = since();
    HIP_TRY_H( hipMalloc( &h->d_nodes, sizeof( GNode ) * scene->n_nodes ) );
    HIP_TRY_H( hipMalloc( &h->d_mats, sizeof( GMat ) * scene->n_nodes ) );
    /* elems[ 0 .. n ) as given; elems[ n .. 2n ) the same slices with each compound's elements ordered by estimated
     * test cost (any-hit occlusion queries are an OR over the elements, so their order is free; closest-hit queries
     * keep the given order because ties go to the first element, compound.c:225-243) */
    std::vector< int32_t > elems2( 2 * ( size_t )scene->n_elems + 1, 0 );
    std::vector< SCEntry > sc_table;   /* pre-order tables of the simple compounds (acn_device.h: simple_compound_hit) */
    std::vector< double > sc_spheres;
    {
        std::vector< double > cost( scene->n_nodes, -1.0 );
        std::function< double( int32_t ) > node_cost = [ & ]( int32_t i ) -> double
        {
            if( cost[ i ] >= 0 ) return cost[ i ];
            const acn_node& a = scene->nodes[ i ];
            double c = 1;
            switch( a.type )
            {
                case ACN_PLANE: c = 0.5; break;
                case ACN_SPHERE: c = 1; break;
                case ACN_SQUAROID: c = 1.5; break;
                case ACN_DISTANCE: c = 60; break;            /* sphere tracing, up to `cycles` evaluations */
                case ACN_NEG: case ACN_SCALE: c = 1 + node_cost( a.child0 ); break;
                case ACN_PAIR_INSIDE: case ACN_PAIR_OUTSIDE: c = 2 + 1.5 * ( node_cost( a.child0 ) + node_cost( a.child1 ) ); break;
                case ACN_COMPOUND: c = 1; for( int32_t k = 0; k < a.child1; k++ ) c += node_cost( scene->elems[ a.child0 + k ] ); break;
                default: break;
            }
            return cost[ i ] = c;
        };
        for( uint32_t k = 0; k < scene->n_elems; k++ ) elems2[ k ] = elems2[ scene->n_elems + k ] = scene->elems[ k ];
        for( uint32_t i = 0; i < scene->n_nodes; i++ )
        {
            const acn_node& a = scene->nodes[ i ];
            if( a.type != ACN_COMPOUND || a.child1 < 2 ) continue;
            int32_t* first = elems2.data() + scene->n_elems + a.child0;
            std::stable_sort( first, first + a.child1, [ & ]( int32_t x, int32_t y ) { return node_cost( x ) < node_cost( y ); } );
        }
    }
    /* the envelope of every entry of elems[ 0 .. 2n ), in the same order: the broad phase of the root loops (root_candidates) */
    std::vector< double > env_tab( 8 * ( size_t )scene->n_elems + 4, -1.0 );
    for( size_t k = 0; k < 2 * ( size_t )scene->n_elems; k++ )
    {
        const acn_node& a = scene->nodes[ elems2[ k ] ];
        if( a.flags & ACN_NODE_HAS_ENVELOPE )
        {
            for( int c = 0; c < 3; c++ ) env_tab[ 4 * k + c ] = a.env_pos[ c ];
            env_tab[ 4 * k + 3 ] = a.env_radius;
        }
    }
    /* elems[ 2n .. 2n + n_nodes ): per node the offset of its interval-prune program (acn_device.h: prune_run) or -1,
     * followed by the programs.  Only root elements of compounds that are CSG composites with at least
     * ACN_PRUNE_MIN nodes get one (small trees are cheaper to walk than to pre-test). */
    h->dev.prune_base = 2 * scene->n_elems;
    {
        size_t min_nodes = 32;
        if( const char* e = getenv( "ACN_PRUNE_MIN" ) ) min_nodes = ( size_t )atoll( e );
        elems2.resize( 2 * ( size_t )scene->n_elems );
        elems2.resize( 2 * ( size_t )scene->n_elems + scene->n_nodes, -1 );
        std::vector< int32_t > size( scene->n_nodes, -1 );
        std::function< int32_t( int32_t ) > subtree = [ & ]( int32_t i ) -> int32_t
        {
            if( size[ i ] >= 0 ) return size[ i ];
            const acn_node& a = scene->nodes[ i ];
            int32_t c = 1;
            if( a.type == ACN_NEG || a.type == ACN_SCALE ) c += subtree( a.child0 );
            else if( a.type == ACN_PAIR_INSIDE || a.type == ACN_PAIR_OUTSIDE ) c += subtree( a.child0 ) + subtree( a.child1 );
            return size[ i ] = c;
        };
        std::vector< uint32_t > prog;
        int max_depth = 0;
        /* postfix code for node i; returns the interval-stack depth it needs.  The child that needs the deeper
         * stack is emitted first (the combining ops are symmetric), which keeps balanced trees within the budget. */
        std::function< int( int32_t, int ) > gen = [ & ]( int32_t i, int depth ) -> int
        {
            const acn_node& a = scene->nodes[ i ];
            int need = 1;
            switch( a.type )
            {
                case ACN_PLANE:    prog.push_back( ACN_PO( ACN_PO_PLANE, i ) ); break;
                case ACN_SPHERE:   prog.push_back( ACN_PO( ACN_PO_SPHERE, i ) ); break;
                case ACN_SQUAROID: prog.push_back( ACN_PO( ACN_PO_QUAD, i ) ); break;
                case ACN_NEG:
                {
                    const acn_node& c = scene->nodes[ a.child0 ];
                    need = gen( a.child0, depth );
                    bool bare_plane = c.type == ACN_PLANE && !( c.flags & ACN_NODE_HAS_ENVELOPE ) && a.child0 != 0;
                    prog.push_back( ACN_PO( ACN_PO_NEG, bare_plane ? a.child0 : 0 ) );
                }
                break;
                case ACN_PAIR_INSIDE: case ACN_PAIR_OUTSIDE:
                {
                    if( depth >= max_depth ) { prog.push_back( ACN_PO( ACN_PO_ALL, 0 ) ); break; }
                    /* n-ary view: chains of the same pair type without envelopes in between are one intersection /
                     * union (the sets H and S do not depend on how the reference's tree is balanced); operands are
                     * combined one after the other, the one needing the deepest stack first */
                    std::vector< int32_t > items, todo{ a.child1, a.child0 };
                    while( !todo.empty() )
                    {
                        int32_t c = todo.back(); todo.pop_back();
                        const acn_node& cn = scene->nodes[ c ];
                        if( cn.type == a.type && !( cn.flags & ACN_NODE_HAS_ENVELOPE ) ) { todo.push_back( cn.child1 ); todo.push_back( cn.child0 ); }
                        else items.push_back( c );
                    }
                    size_t mark = prog.size();
                    std::vector< std::pair< int, std::vector< uint32_t > > > code;
                    for( int32_t c : items )
                    {
                        int d = gen( c, depth + 1 );
                        code.emplace_back( d, std::vector< uint32_t >( prog.begin() + mark, prog.end() ) );
                        prog.resize( mark );
                    }
                    std::stable_sort( code.begin(), code.end(), []( const std::pair< int, std::vector< uint32_t > >& x, const std::pair< int, std::vector< uint32_t > >& y ) { return x.first > y.first; } );
                    need = code[ 0 ].first;
                    for( size_t k = 0; k < code.size(); k++ )
                    {
                        prog.insert( prog.end(), code[ k ].second.begin(), code[ k ].second.end() );
                        if( k > 0 )
                        {
                            prog.push_back( ACN_PO( a.type == ACN_PAIR_INSIDE ? ACN_PO_AND : ACN_PO_OR, 0 ) );
                            if( 1 + code[ k ].first > need ) need = 1 + code[ k ].first;
                        }
                    }
                    if( need > ACN_PRUNE_STACK ) { prog.resize( mark ); prog.push_back( ACN_PO( ACN_PO_ALL, 0 ) ); need = 1; }
                }
                break;
                default: prog.push_back( ACN_PO( ACN_PO_ALL, 0 ) ); break;
            }
            if( a.flags & ACN_NODE_HAS_ENVELOPE ) prog.push_back( ACN_PO( ACN_PO_ENV, i ) );
            return need;
        };
        const size_t max_ops = 256;
        for( uint32_t i = 0; i < scene->n_nodes; i++ )
        {
            const acn_node& c = scene->nodes[ i ];
            if( c.type != ACN_COMPOUND ) continue;
            for( int32_t k = 0; k < c.child1; k++ )
            {
                int32_t e = scene->elems[ c.child0 + k ];
                const acn_node& a = scene->nodes[ e ];
                if( !( a.type == ACN_PAIR_INSIDE || a.type == ACN_PAIR_OUTSIDE ) ) continue;
                if( ( size_t )subtree( e ) < min_nodes || elems2[ h->dev.prune_base + e ] >= 0 ) continue;
                for( max_depth = 12; max_depth >= 1; max_depth-- )   /* the deepest expansion that fits the budget */
                {
                    prog.clear();
                    gen( e, 0 );
                    if( prog.size() < max_ops ) break;
                }
                if( max_depth < 1 ) continue;
                prog.push_back( ACN_PO( ACN_PO_END, 0 ) );
                elems2[ h->dev.prune_base + e ] = ( int32_t )elems2.size();
                h->prune = true;
                for( uint32_t w : prog ) elems2.push_back( ( int32_t )w );
            }
        }
        /* simple compounds (acn_device.h: simple_compound_hit): pre-order ( node, skip ) tables for root elements that
         * are compounds over nothing but compounds and simple leaves; the same per-node offset table locates them */
        if( !getenv( "ACN_NO_SIMPLE_COMPOUNDS" ) )
        {
            std::vector< SCEntry >& sct = sc_table;
            std::vector< int8_t > simple( scene->n_nodes, -1 );
            std::function< bool( int32_t ) > is_simple = [ & ]( int32_t i ) -> bool
            {
                if( simple[ i ] >= 0 ) return simple[ i ] != 0;
                const acn_node& a = scene->nodes[ i ];
                bool ok = a.type == ACN_PLANE || a.type == ACN_SPHERE || a.type == ACN_SQUAROID;
                if( a.type == ACN_COMPOUND )
                {
                    ok = true;
                    for( int32_t k = 0; k < a.child1 && ok; k++ ) ok = is_simple( scene->elems[ a.child0 + k ] );
                }
                simple[ i ] = ok ? 1 : 0;
                return ok;
            };
            const bool no_cull = getenv( "ACN_NO_SC_CULL" ) != nullptr;
            size_t n_bounding = 0, n_reversed = 0;
            const bool no_rev = getenv( "ACN_NO_SC_REVERSED" ) != nullptr;
            std::vector< int32_t > sph_of( scene->n_nodes, -1 );      /* sphere record of a leaf, flags of an entry: the reversed table reuses them */
            std::vector< uint32_t > flags_of( scene->n_nodes, 0u );
            double order_dir[ 3 ] = { 0, 0, 0 };                      /* along which the children of the compounds at hand come later, summed over the compounds */
            auto centre = [ & ]( const acn_node& x, int c ) { return ( x.flags & ACN_NODE_HAS_ENVELOPE ) ? x.env_pos[ c ] : x.pos[ c ]; };
            std::function< void( int32_t, bool ) > emit = [ & ]( int32_t c, bool reversed )   /* children of compound c, depth first */
            {
                const acn_node& a = scene->nodes[ c ];
                if( !reversed && a.child1 > 1 )
                {
                    double mean[ 3 ] = { 0, 0, 0 };
                    for( int32_t k = 0; k < a.child1; k++ ) for( int x = 0; x < 3; x++ ) mean[ x ] += centre( scene->nodes[ scene->elems[ a.child0 + k ] ], x ) / a.child1;
                    for( int32_t k = 0; k < a.child1; k++ ) for( int x = 0; x < 3; x++ )
                        order_dir[ x ] += ( k - 0.5 * ( a.child1 - 1 ) ) * ( centre( scene->nodes[ scene->elems[ a.child0 + k ] ], x ) - mean[ x ] );
                }
                for( int32_t kk = 0; kk < a.child1; kk++ )
                {
                    const int32_t k = reversed ? a.child1 - 1 - kk : kk;
                    int32_t e = scene->elems[ a.child0 + k ];
                    const acn_node& en = scene->nodes[ e ];
                    size_t at = sct.size();
                    SCEntry rec;
                    memcpy( rec.env_pos, en.env_pos, sizeof( rec.env_pos ) );
                    rec.env_radius = en.env_radius; rec.node = e; rec.skip = 0; rec.type = en.type; rec.flags = en.flags & ACN_NODE_HAS_ENVELOPE;
                    sct.push_back( rec );
                    if( en.type == ACN_COMPOUND ) emit( e, reversed );
                    sct[ at ].skip = ( int32_t )sct.size();   /* the entry behind e's subtree */
                    if( reversed )
                    {
                        sct[ at ].flags = flags_of[ e ];
                        if( en.type == ACN_SPHERE ) sct[ at ].skip = sph_of[ e ];
                        continue;
                    }
                    if( ( rec.flags & ACN_NODE_HAS_ENVELOPE ) && !no_cull )   /* does the envelope contain every leaf below? (simple_compound_hit: CULL) */
                    {
                        bool inside = true;
                        for( size_t j = at; j < sct.size() && inside; j++ )
                        {
                            const acn_node& ln = scene->nodes[ sct[ j ].node ];
                            if( ln.type == ACN_COMPOUND ) continue;
                            if( ln.type != ACN_SPHERE ) { inside = false; break; }
                            double d2 = 0;
                            for( int x = 0; x < 3; x++ ) d2 += ( ln.pos[ x ] - en.env_pos[ x ] ) * ( ln.pos[ x ] - en.env_pos[ x ] );
                            inside = sqrt( d2 ) + fabs( ln.prm[ 0 ] ) <= fabs( en.env_radius ) * ( 1.0 - 1E-9 );
                        }
                        if( inside ) { sct[ at ].flags |= ACN_SC_BOUNDING; n_bounding++; }
                    }
                    if( en.type == ACN_SPHERE )   /* a leaf never follows its link: it names the sphere's record instead */
                    {
                        sph_of[ e ] = ( int32_t )( sc_spheres.size() / 4 );
                        sct[ at ].skip = sph_of[ e ];
                        sct[ at ].flags |= ACN_SC_SPHERE | ( en.surface_roughness > 0 ? ACN_SC_ROUGH : 0u );
                        for( int x = 0; x < 3; x++ ) sc_spheres.push_back( en.pos[ x ] );
                        sc_spheres.push_back( en.prm[ 0 ] );
                    }
                    flags_of[ e ] = sct[ at ].flags;
                }
            };
            for( int root : { scene->light_root, scene->matter_root } )
            {
                const acn_node& r = scene->nodes[ root ];
                for( int32_t k = 0; k < r.child1; k++ )
                {
                    int32_t e = scene->elems[ r.child0 + k ];
                    if( scene->nodes[ e ].type != ACN_COMPOUND || !is_simple( e ) || elems2[ h->dev.prune_base + e ] >= 0 ) continue;
                    elems2[ h->dev.prune_base + e ] = ( int32_t )elems2.size();
                    elems2.push_back( ( int32_t )sct.size() );     /* first entry */
                    size_t first = sct.size();
                    order_dir[ 0 ] = order_dir[ 1 ] = order_dir[ 2 ] = 0;
                    emit( e, false );
                    const size_t count = sct.size() - first;
                    elems2.push_back( ( int32_t )count );          /* entry count */
                    /* the same subtree with the children of every compound in reverse order, for rays that run against the order of the
                     * first (simple_compound_hit: the walk culls more the sooner it meets the near leaves); -1: none */
                    const double len = sqrt( order_dir[ 0 ] * order_dir[ 0 ] + order_dir[ 1 ] * order_dir[ 1 ] + order_dir[ 2 ] * order_dir[ 2 ] );
                    if( !no_rev && !no_cull && count >= 64 && len > 0 )
                    {
                        elems2.push_back( ( int32_t )sct.size() );
                        elems2.push_back( ( int32_t )( sc_spheres.size() / 4 ) );
                        for( int x = 0; x < 3; x++ ) sc_spheres.push_back( order_dir[ x ] / len );
                        sc_spheres.push_back( 0.0 );
                        emit( e, true );
                        n_reversed += count;
                    }
                    else { elems2.push_back( -1 ); elems2.push_back( 0 ); }
                    nodes[ e ].flags |= ACN_GFLAG_SIMPLE_COMPOUND;
                    h->prune = true;   /* the extras kernel variants */
                }
            }
            if( getenv( "ACN_VERBOSE" ) && sct.size() ) fprintf( stderr, "actinon_hip: simple compounds: %zu entries, %zu with a verified bounding envelope, %zu again in reversed order\n", sct.size() - n_reversed, n_bounding, n_reversed );
        }
        elems2.push_back( 0 );
    }
    h->scene_bytes[ 2 ] = sizeof( int32_t ) * elems2.size();
    HIP_TRY_H( hipMalloc( &h->d_elems, sizeof( int32_t ) * elems2.size() ) );
    HIP_TRY_H( hipMalloc( &h->d_sc_table, sizeof( SCEntry ) * ( sc_table.size() ? sc_table.size() : 1 ) ) );
    t_up[ 1 ] = since();
    if( sc_table.size() ) HIP_TRY_H( hipMemcpy( h->d_sc_table, sc_table.data(), sizeof( SCEntry ) * sc_table.size(), hipMemcpyHostToDevice ) );
    h->dev.sc_table = h->d_sc_table;
    HIP_TRY_H( hipMalloc( &h->d_sc_spheres, sizeof( double ) * ( sc_spheres.size() ? sc_spheres.size() : 4 ) ) );
    if( sc_spheres.size() ) HIP_TRY_H( hipMemcpy( h->d_sc_spheres, sc_spheres.data(), sizeof( double ) * sc_spheres.size(), hipMemcpyHostToDevice ) );
    h->dev.sc_spheres = h->d_sc_spheres;
    HIP_TRY_H( hipMalloc( &h->d_env_tab, sizeof( double ) * env_tab.size() ) );
    HIP_TRY_H( hipMemcpy( h->d_env_tab, env_tab.data(), sizeof( double ) * env_tab.size(), hipMemcpyHostToDevice ) );
    h->dev.env_tab = ( CDblP )h->d_env_tab;
    /* Width of a shading task (size_class in acn_pipeline.h).  Narrow groups waste less of a sample loop's last round;
     * a whole wavefront per point keeps the rays of a round on one origin, which pays when a sample's traversal is long
     * and divergent (nested compounds, CSG objects with prune programs: the scenes of the "extras" kernel variants).
     * Measured, 4 lanes: wine_glass 1080p (200 / 64 samples) 79.8 ms narrow, 85.2 wide from 33 samples; many_spheres
     * 1080p p256 every 16th pixel 3.94 s narrow, 2.90 s wide; diamond 1080p p512 4.58 s narrow, 4.05 s wide. */
    h->dev.class0_min = h->tun.class0_min ? h->tun.class0_min : ( h->prune ? 32u : 255u );
    HIP_TRY_H( hipMalloc( &h->d_textures, sizeof( acn_texture ) * ( scene->n_textures ? scene->n_textures : 1 ) ) );
    if( scene->n_textures ) HIP_TRY_H( hipMemcpy( h->d_textures, scene->textures, sizeof( acn_texture ) * scene->n_textures, hipMemcpyHostToDevice ) );
    HIP_TRY_H( hipMalloc( &h->d_counters, sizeof( unsigned long long ) * ACN_CNT_SLOTS ) );
    HIP_TRY_H( hipMemset( h->d_counters, 0, sizeof( unsigned long long ) * ACN_CNT_SLOTS ) );
    HIP_TRY_H( hipMalloc( &h->d_counters_keep, sizeof( unsigned long long ) * ACN_CNT_SLOTS ) );
    HIP_TRY_H( hipMalloc( &h->d_counts, sizeof( uint32_t ) * QC_N * ACN_LEVEL_BLOCKS ) );
    HIP_TRY_H( hipHostMalloc( &h->h_counts, sizeof( uint32_t ) * QC_N * ACN_LEVEL_BLOCKS ) );
    HIP_TRY_H( hipMemcpy( h->d_nodes, nodes.data(), sizeof( GNode ) * scene->n_nodes, hipMemcpyHostToDevice ) );
    HIP_TRY_H( hipMemcpy( h->d_mats, mats.data(), sizeof( GMat ) * scene->n_nodes, hipMemcpyHostToDevice ) );
    HIP_TRY_H( hipMemcpy( h->d_elems, elems2.data(), sizeof( int32_t ) * elems2.size(), hipMemcpyHostToDevice ) );
    h->dev.nodes = ( NodeP )h->d_nodes; h->dev.gnodes = ( NodeP )h->d_nodes;
    h->dev.mats = ( MatP )h->d_mats;
    h->dev.elems = ( ElemP )h->d_elems;
    h->dev.textures = ( TexP )h->d_textures;
    h->dev.light_root = scene->light_root;
    h->dev.matter_root = scene->matter_root;
    h->dev.n_nodes = scene->n_nodes;
    h->dev.n_elems = scene->n_elems;
    h->dev.prm = scene->params;
    {
        const acn_node& lr = scene->nodes[ scene->light_root ];
        h->n_lights = lr.child1 > 0 ? ( size_t )lr.child1 : 1;
        for( int k = 0; k < lr.child1; k++ )
        {
            int t = scene->nodes[ scene->elems[ lr.child0 + k ] ].type;
            if( t != ACN_PLANE && t != ACN_SPHERE ) h->leaf_lights = false;
        }
    }
    {
        /* LDS plan of the machine kernels (160 KB per CU, 4 blocks of 256 lanes wanted per CU => 40 KB per block):
         *   nodes + stacks   when the node array is small (<= 8 KB: wine_glass 6 KB);
         *   nodes only       up to 40 KB (diamond): staging the per-lane node reads pays more than the stacks;
         *   stacks only      beyond (the node array stays in global memory / L2).
         * ACN_LDS_MAX (bytes of nodes that may be staged) and ACN_LDS_STACK=0|1 override. */
        size_t lds_max = 40960;
        /* Round 4: nodes are staged only for scenes whose roots hold GENERIC nested compounds (hanging_lamps_in_row: compounds of
         * CSG objects) -- the one traversal left that reads nodes per lane (compound_ray_hit_dev).  The lock-step machines read
         * every node through scalar loads from global memory whatever is staged, and the leaves a root loop tests in line are
         * better off with scalar loads too: a staged node comes back through ds_read into VGPRs (1080p wine_glass 51.8 -> 50.6 ms
         * without staging, profiles/r04); the diamond's 40 KB of nodes had cost it the LDS stacks of its CSG machines. */
        bool generic_compound = false;
        for( int root : { scene->light_root, scene->matter_root } )
        {
            const acn_node& r = scene->nodes[ root ];
            for( int32_t k = 0; k < r.child1; k++ )
            {
                const int32_t e = scene->elems[ r.child0 + k ];
                if( scene->nodes[ e ].type == ACN_COMPOUND && !( nodes[ e ].flags & ACN_GFLAG_SIMPLE_COMPOUND ) ) generic_compound = true;
            }
        }
        if( !generic_compound ) lds_max = 0;
        if( const char* e = getenv( "ACN_LDS_MAX" ) ) lds_max = ( size_t )atoll( e );
        size_t need = sizeof( GNode ) * ( size_t )scene->n_nodes;
        /* every machine kernel owns the stacks AND the ray pool of its workgroup (pooled_machine_hit); nodes are staged in front
         * of them only if the three fit 40 KB (four workgroups per CU) */
        h->lds_bytes = need <= lds_max && need + ACN_LDS_STACK_BYTES + ACN_LDS_POOL_BYTES + ACN_LDS_ORG_BYTES <= 40960 ? need : 0;
        h->lds_stack_bytes = ACN_LDS_STACK_BYTES + ACN_LDS_POOL_BYTES + ACN_LDS_ORG_BYTES;
    }
    h->dev.flags = h->d_counts + QC_FLAGS;
    h->dev.lds_stack = h->lds_stack_bytes ? 0u : ACN_NO_LDS_STACK;   /* the kernels that own a stack area set the offset */
    HIP_TRY_H( hipMemset( h->d_counts, 0, sizeof( uint32_t ) * QC_N * ACN_LEVEL_BLOCKS ) );
    /* camera basis on the device so that it shares the device's arithmetic */
    {
        M3* d_rot = nullptr; double* d_uf = nullptr;
        HIP_TRY_H( hipMalloc( &d_rot, sizeof( M3 ) ) );
        HIP_TRY_H( hipMalloc( &d_uf, sizeof( double ) ) );
        t_up[ 2 ] = since();
        early_lanes_join( h );   /* before the first kernel: nothing of this handle runs while hardware queues are being made */
        t_up[ 3 ] = since();
        hipLaunchKernelGGL( k_camera_setup, dim3( 1 ), dim3( 1 ), 0, h->stream, h->dev, d_rot, d_uf );
        HIP_TRY_H( hipGetLastError() );
        HIP_TRY_H( hipStreamSynchronize( h->stream ) );
        HIP_TRY_H( hipMemcpy( &h->dev.camera_rotation, d_rot, sizeof( M3 ), hipMemcpyDeviceToHost ) );
        HIP_TRY_H( hipMemcpy( &h->dev.unit_f, d_uf, sizeof( double ), hipMemcpyDeviceToHost ) );
        hipFree( d_rot ); hipFree( d_uf );
    }
    if( h->tun.debug_chunks )
        fprintf( stderr, "[acn upload] %u nodes: the handle's stream %.2f ms, events %.2f, tables on the host %.2f, device copies %.2f, waited for %d early lanes %.2f, first kernel of the library (camera set-up) %.2f\n",
                 ( unsigned )scene->n_nodes, t_stream1 - t_stream0, t_up[ 0 ] - t_stream1 + t_stream0, t_up[ 1 ] - t_up[ 0 ], t_up[ 2 ] - t_up[ 1 ], ( int )h->early_made.size(), t_up[ 3 ] - t_up[ 2 ], since() - t_up[ 3 ] );
    *out = h;
    return ACN_OK;
}

static void free_workspace( acn_scene_handle* h )
{
    Workspace& w = h->ws;
    if( w.tasks ) hipFree( w.tasks );
    for( int k = 0; k < ACN_NCLASS; k++ ) if( w.idx[ k ] ) hipFree( w.idx[ k ] );
    if( w.children ) hipFree( w.children );
    if( w.hard_shadow ) hipFree( w.hard_shadow );
    if( w.hard_path ) hipFree( w.hard_path );
    for( int k = 0; k < 2; k++ ) if( w.rays[ k ] ) hipFree( w.rays[ k ] );
    if( w.stacks ) hipFree( w.stacks );
    w = Workspace();
}

extern "C" void acn_scene_free( acn_scene_handle* h )
{
    if( !h ) return;
    hipSetDevice( h->device );
    early_lanes_join( h );
    for( acn_scene_handle* l : h->early_made ) acn_scene_free( l );
    h->early_made.clear();
    for( acn_scene_handle* l : h->lanes ) acn_scene_free( l );
    h->lanes.clear();
    if( h->worker ) { h->worker->stop(); delete h->worker; h->worker = nullptr; }
    free_workspace( h );
    if( h->d_counts ) hipFree( h->d_counts );
    if( h->h_counts ) hipHostFree( h->h_counts );
    if( h->d_accum ) hipFree( h->d_accum );
    if( h->d_lane_pos ) hipFree( h->d_lane_pos );
    if( h->d_lane_out ) hipFree( h->d_lane_out );
    if( h->d_shard_pos ) hipFree( h->d_shard_pos );
    if( !h->is_lane )   /* a lane borrows the resident scene of its parent */
    {
        if( h->d_nodes ) hipFree( h->d_nodes );
        if( h->d_mats ) hipFree( h->d_mats );
        if( h->d_elems ) hipFree( h->d_elems );
        if( h->d_textures ) hipFree( h->d_textures );
        if( h->d_sc_table ) hipFree( h->d_sc_table );
        if( h->d_sc_spheres ) hipFree( h->d_sc_spheres );
        if( h->d_env_tab ) hipFree( h->d_env_tab );
    }
    if( h->d_counters ) hipFree( h->d_counters );
    if( h->d_counters_keep ) hipFree( h->d_counters_keep );
    for( auto& e : h->events ) { hipEventDestroy( e.a ); hipEventDestroy( e.b ); }
    if( h->ev0 ) hipEventDestroy( h->ev0 );
    if( h->ev1 ) hipEventDestroy( h->ev1 );
    if( h->ev_fork ) hipEventDestroy( h->ev_fork );
    if( h->ev_path ) hipEventDestroy( h->ev_path );
    if( h->ev_join ) hipEventDestroy( h->ev_join );
    if( h->side_stream ) hipStreamDestroy( h->side_stream );
    if( h->stream ) hipStreamDestroy( h->stream );
    delete h;
}

/* Queue capacities.  Only one chunk of positions is in flight per pipeline run, so the queues are sized for a chunk, not
 * for the call, and each queue for its own demand:
 *   - rates unknown (first call on a handle): a small uniform starter set; the first chunk of the call is small, teaches
 *     the rates (render_chunk) and launch_render comes back here;
 *   - rates known: room for as many positions as the call has (at most ACN_CHUNK_TARGET) at 1 / 0.7 of the learned rates,
 *     scaled down to the handle's budget (ACN_WORKSPACE_MB; default 8 GiB or a quarter of the free device memory) if that is
 *     less.  The chunk size follows the capacities (launch_render), so a small workspace costs more chunks, not
 *     correctness; if hipMalloc refuses, the request is halved until it fits. */
#define ACN_CHUNK_TARGET ( ( size_t )1 << 22 )
#define ACN_STARTER_RECORDS ( ( size_t )1 << 20 )
static double f_max_host( double a, double b ) { return a > b ? a : b; }
static bool rates_known( const acn_scene_handle* h ) { return h->rate[ WQ_TASKS ] > 0 || h->rate[ WQ_RAYS ] > 0 || h->rate[ WQ_HARD_SHADOW ] > 0; }

/* positions a chunk may have so that every queue stays below 70 % of its capacity */
static size_t chunk_for_caps( const acn_scene_handle* h )
{
    double chunk = 2.0e9;
    for( int q = 0; q < WQ_N; q++ )
    {
        const double r = h->rate[ q ] > 1e-3 ? h->rate[ q ] : 1e-3;
        const double c = h->ctl.fill_target * ( double )h->ws.cap[ q ] / r;
        if( c < chunk ) chunk = c;
    }
    return chunk < 64 ? 64 : ( size_t )chunk;
}

static int ensure_workspace( acn_scene_handle* h, size_t n )
{
    Workspace& w = h->ws;
    const size_t budget = h->workspace_budget / h->budget_div;
    const size_t stack_waves = ( size_t )( h->walk_grid > h->grid ? h->walk_grid : h->grid ) * 4;
    const size_t stack_bytes = stack_waves * h->tun.stack_cap * sizeof( RayTask );
    size_t want[ WQ_N ];
    bool trim = false;
    if( h->tun.ws_uniform )
    {
        size_t per_rec = wq_bytes[ WQ_HARD_SHADOW ];
        for( int q = 0; q < WQ_N; q++ ) per_rec += wq_bytes[ q ];
        size_t recs = budget > stack_bytes ? ( budget - stack_bytes ) / per_rec : 65536;
        if( recs > 0x7FFFFF00ull ) recs = 0x7FFFFF00ull;
        for( int q = 0; q < WQ_N; q++ ) want[ q ] = recs;
        want[ WQ_HARD_SHADOW ] = 2 * recs;
    }
    else if( !rates_known( h ) )
    {
        /* starter set: 2^20 records per queue (the deferred-shadow queue twice that), less for a call of a few positions */
        const size_t s = h->dev.prm.path_samples ? h->dev.prm.path_samples : 1;
        const size_t per_pos = ( s + 2 ) * ( s > 16 ? s / 16 : 1 ) + ( size_t )h->dev.prm.direct_samples * h->n_lights;
        size_t recs = n * per_pos + 65536;
        if( recs > ACN_STARTER_RECORDS ) recs = ACN_STARTER_RECORDS;
        size_t per_rec = wq_bytes[ WQ_HARD_SHADOW ];
        for( int q = 0; q < WQ_N; q++ ) per_rec += wq_bytes[ q ];
        const size_t max_recs = budget > stack_bytes ? ( budget - stack_bytes ) / per_rec : 0;
        if( recs > max_recs ) recs = max_recs;
        for( int q = 0; q < WQ_N; q++ ) want[ q ] = recs;
        want[ WQ_HARD_SHADOW ] = 2 * recs;
    }
    else
    {
        double positions = ( double )( n < ACN_CHUNK_TARGET ? n : ACN_CHUNK_TARGET );
        double bytes = 0;
        /* 40 % above what the rates ask for: the rates move a little from frame to frame, and a queue that is a few per
         * cent short turns one chunk per lane into two (a second chain of launches: c2 36 -> 50 ms) or, worse, makes the
         * lane re-allocate in the middle of a frame (hipFree synchronises the device: paraffin_lamp 440 -> 700 ms) */
        const double slack = 1.4;
        for( int q = 0; q < WQ_N; q++ ) bytes += ( slack * h->rate[ q ] * positions / 0.7 + 65536.0 ) * ( double )wq_bytes[ q ];
        const double room = budget > stack_bytes ? ( double )( budget - stack_bytes ) : 0.0;
        if( bytes > room ) positions *= room / bytes;
        for( int q = 0; q < WQ_N; q++ )
        {
            double c = slack * h->rate[ q ] * positions / 0.7 + 65536.0;
            want[ q ] = c > 4.0e9 ? 0xFFFFFF00ull : ( size_t )c;
        }
    }
    for( int q = 0; q < WQ_N; q++ ) { if( want[ q ] < 65536 ) want[ q ] = 65536; if( want[ q ] > 0xFFFFFF00ull ) want[ q ] = 0xFFFFFF00ull; }
    /* keep what is there while it holds what the rates ask for (the slack is for growth, not a reason to re-allocate) */
    bool fits = w.stack_waves >= stack_waves;
    for( int q = 0; q < WQ_N; q++ ) if( ( double )w.cap[ q ] < ( double )want[ q ] / 1.4 ) fits = false;
    /* ... and give back what the first, small chunks of a handle over-estimated (their dead slots do not scale): once, when
     * the rates come from a large chunk and the queues hold 40 % more than those ask for (slack included) */
    /* ... in a WINDOW: the first few sizing steps after the rates were learned (the second and third call of a handle).  Rates
     * decay slowly towards what the chunks really leave, so without the window the condition could first become true ten frames
     * later and put 100 ms of hipFree + hipMalloc into an arbitrary frame (round 4, session 10: the 1080p bench line read 68.6 ms
     * instead of 51.8 because the trim fell into its ten timed steps) */
    if( rates_known( h ) && h->rate_cnt >= 32768 ) w.sized_calls++;
    if( fits && rates_known( h ) && !h->tun.ws_uniform && h->rate_cnt >= 32768 && !w.trimmed && w.sized_calls <= 3 )
    {
        size_t have = 0, need = 0;
        for( int q = 0; q < WQ_N; q++ ) { have += ( size_t )w.cap[ q ] * wq_bytes[ q ]; need += want[ q ] * wq_bytes[ q ]; }
        if( ( double )have > 1.25 * ( double )need && have - need > ( ( size_t )1 << 29 ) ) { fits = false; trim = true; }
    }
    if( fits ) return ACN_OK;
    const uint64_t allocs_before = w.allocs;
    free_workspace( h );
    w.allocs = allocs_before + 1;
    for( ;; )
    {
        hipError_t e = hipSuccess;
        size_t total = 0;
        auto grab = [ & ]( void** p, size_t bytes ) { if( e == hipSuccess ) { e = hipMalloc( p, bytes ); total += bytes; } };
        grab( ( void** )&w.children, sizeof( HitRec ) * want[ WQ_CHILDREN ] );
        grab( ( void** )&w.tasks, sizeof( DTask ) * want[ WQ_TASKS ] );
        for( int k = 0; k < ACN_NCLASS; k++ ) grab( ( void** )&w.idx[ k ], sizeof( uint32_t ) * want[ WQ_TASKS ] );
        grab( ( void** )&w.hard_shadow, sizeof( HardShadow ) * want[ WQ_HARD_SHADOW ] );
        grab( ( void** )&w.hard_path, sizeof( HardPath ) * want[ WQ_HARD_PATH ] );
        for( int k = 0; k < 2; k++ ) grab( ( void** )&w.rays[ k ], sizeof( RayTask ) * want[ WQ_RAYS ] );
        grab( ( void** )&w.stacks, stack_bytes );
        if( e == hipSuccess ) { w.bytes = total; break; }
        ( void )hipGetLastError();
        free_workspace( h );
        w.allocs = allocs_before + 1;
        bool floor = true;
        for( int q = 0; q < WQ_N; q++ ) { if( want[ q ] > 65536 ) floor = false; want[ q ] = want[ q ] / 2 < 65536 ? 65536 : want[ q ] / 2; }
        if( floor ) return fail( ACN_ERR_DEVICE, std::string( "queue workspace: " ) + hipGetErrorString( e ) );
    }
    for( int q = 0; q < WQ_N; q++ ) w.cap[ q ] = ( uint32_t )want[ q ];
    w.stack_waves = stack_waves;
    w.trimmed = trim;
    return ACN_OK;
}

/* per-launch HIP events (stage times of acn_last_stage_ms) cost ~0.7 % of a frame and more of a small one: only
 * with ACN_OPT_STAGE_TIMING; launch counts and pipeline statistics are kept either way */
static int stage_begin( acn_scene_handle* h, int stage, hipStream_t stream )
{
    if( !h->stage_timing ) { h->cur_stage = stage; return ACN_OK; }
    if( h->events_used == h->events.size() )
    {
        StageEvents e{};
        HIP_TRY( hipEventCreate( &e.a ) );
        HIP_TRY( hipEventCreate( &e.b ) );
        h->events.push_back( e );
    }
    h->events[ h->events_used ].stage = stage;
    HIP_TRY( hipEventRecord( h->events[ h->events_used ].a, stream ) );
    return ACN_OK;
}

static int stage_end( acn_scene_handle* h, hipStream_t stream )
{
    if( !h->stage_timing ) { h->launches[ h->cur_stage ]++; return ACN_OK; }
    HIP_TRY( hipEventRecord( h->events[ h->events_used ].b, stream ) );
    h->launches[ h->events[ h->events_used ].stage ]++;
    h->events_used++;
    return ACN_OK;
}

static SceneArgs scene_args( const acn_scene_handle* h )
{
    SceneArgs s;
    s.dev = h->dev; s.nodes = h->d_nodes; s.mats = h->d_mats; s.elems = h->d_elems; s.textures = h->d_textures;
    return s;
}
static KernelFlags kernel_flags( const acn_scene_handle* h )
{
    KernelFlags f;
    f.count = h->count_work; f.leaf_lights = h->leaf_lights; f.lds_nodes = h->lds_bytes != 0; f.prune = h->prune;
    return f;
}
/* the workspace as the kernels of path level `level` see it */
static LevelQ level_queues( const acn_scene_handle* h, int level )
{
    const Workspace& w = h->ws;
    LevelQ q;
    q.tasks = w.tasks; for( int k = 0; k < ACN_NCLASS; k++ ) q.idx[ k ] = w.idx[ k ];
    q.task_cap = w.cap[ WQ_TASKS ]; q.child_cap = w.cap[ WQ_CHILDREN ]; q.hs_cap = w.cap[ WQ_HARD_SHADOW ]; q.hard_cap = w.cap[ WQ_HARD_PATH ];
    q.ray_cap = w.cap[ WQ_RAYS ];
    q.children = w.children; q.hard_shadow = w.hard_shadow; q.hard_path = w.hard_path;
    q.rays[ 0 ] = w.rays[ 0 ]; q.rays[ 1 ] = w.rays[ 1 ];
    q.stacks = w.stacks; q.stack_cap = h->tun.stack_cap; q.stack_use = h->tun.stack_use;
    q.counts = h->d_counts + ( size_t )level * QC_N;
    q.prev_children = h->d_counts + ( size_t )( level > 0 ? level - 1 : 0 ) * QC_N + QC_CHILDREN;
    q.grid = h->grid; q.shade_grid = h->shade_grid;
    q.fetch_walk = h->tun.fetch_walk; q.fetch_hard = h->tun.fetch_hard; q.private_limit = h->tun.private_limit; q.fetch_shade = h->tun.fetch_shade;
    /* the outermost sample loops are those of level 0 */
    const bool sharded = level == 0 && h->shard_world > 1;
    q.shard_rank = sharded ? h->shard_rank : 0u; q.shard_world = sharded ? h->shard_world : 1u;
    q.emit_terms = sharded && h->shard_rank != 0 ? 0u : 1u;
    return q;
}
static size_t machine_lds_bytes( const acn_scene_handle* h ) { return h->lds_bytes + h->lds_stack_bytes; }

/* launches of k_walk for path level `level`: ACN_WALK_PASSES, but no more than the hits of the level have depth left */
static uint32_t walk_passes_of_level( const acn_scene_handle* h, int level )
{
    const uint64_t depth_left = h->dev.prm.trace_depth > 10ull * ( uint64_t )level ? h->dev.prm.trace_depth - 10ull * ( uint64_t )level : 1;
    uint32_t passes = h->tun.walk_passes;
    if( passes > depth_left + 1 ) passes = ( uint32_t )depth_left + 1;
    /* The chunks of a call see the same mix of pixels (TileOrder), so the passes that had input in the last chunk, plus
     * one, are the passes this chunk needs: the last launch of a level finishes whatever is left on the private stacks in
     * any case, so a guess that is too low costs time, never rays.  (A frame without specular surfaces: 2 launches per
     * level instead of 12.) */
    const uint32_t seen = h->tun.learn_passes ? h->walk_passes_seen[ level ] : 0u;
    if( seen && seen + 1 < passes ) passes = seen + 1;
    return passes;
}

/* Workgroups for a launch whose input had `seen` items in the last chunk of `seen_cnt` positions, scaled to this chunk's `cnt`
 * positions: enough workgroups for twice that input at `per_wg` items each -- what ONE workgroup takes on at a time (256
 * records, 256 / lanes-per-task shading tasks), times ACN_GRID_PASSES -- at least 16, at most the persistent grid.  The
 * kernels are persistent and fetch their work through cursors, so ANY grid finishes ANY input: a guess that is too small
 * costs time, never work.
 * OFF by default since the closing measurements of round 3 (profiles/r03/learned_grids_*.txt, four same-box sessions): the
 * 1080p frame gains 0.2 - 0.7 ms of 70 and C1 0.07 of 1.41 ms, but paraffin_lamp 400x600 loses 15 - 20 % (420 -> 510 ms, in
 * mode 2 as well, i.e. through launches whose input was empty in the chunk before), hanging_lamp 600x800 2 - 5 % and the 1/8
 * share of the 1080p frame 2 - 3 %.  What it buys: most launches of a chain are small (a generation of a few thousand rays, the
 * shading tasks of a size class nothing falls into), and a launch of 512 workgroups that has nothing to do still has to
 * get every one of them onto a chip that the other lanes keep busy -- 0.3 - 1 ms each in the kernel trace of round 2. */
static unsigned learned_grid( const acn_scene_handle* h, uint32_t seen, uint32_t cnt, uint32_t per_wg, unsigned full )
{
    if( !h->tun.learn_grids || h->seen_cnt == 0 ) return full;
    if( h->tun.learn_grids == 2 ) return seen == 0 ? 16u : full;
    const double items = 2.0 * ( double )seen * ( double )cnt / ( double )h->seen_cnt + 1.0;
    double g = items / ( ( double )per_wg * h->tun.grid_passes );
    if( g < 16.0 ) g = 16.0;
    return g >= ( double )full ? full : ( unsigned )g;
}

#define ACN_LAUNCH( h, stage, stream, call ) do { int st_ = stage_begin( h, stage, stream ); if( st_ != ACN_OK ) return st_; call; \
    HIP_TRY( hipGetLastError() ); if( ( st_ = stage_end( h, stream ) ) != ACN_OK ) return st_; } while( 0 )

/* One chunk of positions [ base, base + cnt ).  The whole chain -- per path level: ( k_shade_hits -> ) the passes of
 * k_walk -> k_shade x 4 size classes -> k_hard_shadow -> k_hard_path -- is enqueued blind: every kernel takes
 * its input count from the counter block of its level on the device, and a level that turns out to be empty costs a few
 * launches of waves that exit at once.  The host synchronises ONCE, at the end, to read the counter blocks: overflow
 * flags (the chunk is then redone smaller) and statistics. */
static int render_chunk( acn_scene_handle* h, const double* d_pos_xy, size_t first_pixel, uint32_t base, uint32_t cnt, TileOrder order,
                         hipStream_t stream, int* overflow, uint32_t* fill, double* dead_share )
{
    *overflow = 0;
    *dead_share = 0;
    for( int q = 0; q < WQ_N; q++ ) fill[ q ] = 0;
    const int levels = h->n_levels;
    const KernelFlags f = kernel_flags( h );
    const SceneArgs s = scene_args( h );
    const size_t lds = machine_lds_bytes( h );
    HIP_TRY( hipMemsetAsync( h->d_counts, 0, sizeof( uint32_t ) * QC_N * levels, stream ) );
    for( int level = 0; level < levels; level++ )
    {
        const LevelQ q = level_queues( h, level );
        /* the path-sample hits of the level before are shaded (level >= 1), then the specular rays walked: generation
         * passes while the generations are large, the rest on the waves' private stacks (k_walk); a level has at most as
         * many generations as its hits have depth left */
        LevelQ qg = q;   /* the level's queues with the grid of the launch at hand (learned_grid) */
        /* a small chunk (<= 2^17 positions) of a frame without path tracing finishes every generation that is no larger than
         * itself on the private stacks: its generations are not worth a launch each (C1, 120 000 pixels on one lane: 1.41 ->
         * 1.15 ms).  With path samples the rule was measured and dropped: the 1/8 share of the 1080p frame 15.2 -> 14.8 ms and
         * hanging_lamp 600x800 -3 %, but paraffin_lamp 400x600 +8 % -- the rays of a CSG scene are worth redistributing
         * (profiles/r03/private_limit_small_frames.txt) */
        if( !h->tun.private_limit_set && h->dev.prm.path_samples == 0 && cnt <= ( 1u << 17 ) && cnt > qg.private_limit ) qg.private_limit = cnt;
        if( level > 0 )
        {
            qg.grid = learned_grid( h, h->seen_hits[ level ], cnt, 256u, h->grid );
            ACN_LAUNCH( h, 0, stream, acn_launch_shade_hits( f.count, qg, stream, s, h->d_accum, h->d_counters ) );
        }
        const uint32_t passes = walk_passes_of_level( h, level );
        for( uint32_t pass = 0; pass < passes; pass++ )
        {
            /* the last launch of a level finishes whatever is left on the private stacks: the input of all later generations */
            uint32_t seen = level == 0 && pass == 0 ? h->seen_cnt : h->seen_gen[ level ][ pass ];
            if( pass + 1 == passes ) for( uint32_t g = pass + 1; g <= ACN_MAX_WALK_PASSES; g++ ) seen += h->seen_gen[ level ][ g ];
            /* (k_walk keeps its full grid unless the pass had no input at all: its rays multiply on the private stacks, and
             * hanging_lamp 600x800 lost 7 % with grids sized to the input) */
            qg.grid = ( level == 0 && pass == 0 ) || seen != 0 ? h->walk_grid : learned_grid( h, 0, cnt, 512u, h->walk_grid );
            ACN_LAUNCH( h, 0, stream, acn_launch_walk( f, pass, pass + 1 == passes, qg, lds, stream, s, d_pos_xy, first_pixel, base,
                                                       level == 0 && pass == 0 ? cnt : 0u, order, h->d_accum, h->d_counters ) );
        }
        /* ACN_SHADE_FISSION=1 (off by default: measured slower, see Tunables).  The two sample loops of a shading point share
         * nothing but the task record, so the level can fork: the direct-light loops and the shadow rays they defer on the side
         * stream, the path loop and the path rays it defers on the main one.  The critical path of a level is then
         * walk -> max( direct + hard_shadow, path + hard_path ) instead of their sum.  k_hard_shadow also takes the probes the path
         * loop appends, so it waits for that launch (ev_path); the queues are the level's, so the next level waits for both.
         * The last level of a frame casts no path rays (depth <= 10) and a frame without path samples has one level: no fork. */
        const bool fork = h->tun.shade_fission && level + 1 < levels && h->side_stream != nullptr;
        hipStream_t direct_stream = fork ? h->side_stream : stream;
        if( fork )
        {
            HIP_TRY( hipEventRecord( h->ev_fork, stream ) );
            HIP_TRY( hipStreamWaitEvent( h->side_stream, h->ev_fork, 0 ) );
        }
        for( int part = fork ? ACN_SHADE_DIRECT : ACN_SHADE_BOTH; part <= ( fork ? ACN_SHADE_PATH : ACN_SHADE_BOTH ); part++ )
        {
            hipStream_t part_stream = part == ACN_SHADE_DIRECT ? direct_stream : stream;
            qg.shade_grid = learned_grid( h, h->seen_class[ level ][ 0 ], cnt, 4u, h->shade_grid );
            ACN_LAUNCH( h, 1, part_stream, acn_launch_shade64( f, qg, part_stream, s, h->d_accum, h->d_counters, part ) );
            qg.shade_grid = learned_grid( h, h->seen_class[ level ][ 1 ], cnt, 16u, h->shade_grid );
            ACN_LAUNCH( h, 1, part_stream, acn_launch_shade16( f, qg, part_stream, s, h->d_accum, h->d_counters, part ) );
            qg.shade_grid = learned_grid( h, h->seen_class[ level ][ 2 ], cnt, 64u, h->shade_grid );
            ACN_LAUNCH( h, 1, part_stream, acn_launch_shade4( f, qg, part_stream, s, h->d_accum, h->d_counters, part ) );
            qg.shade_grid = learned_grid( h, h->seen_class[ level ][ 3 ], cnt, 256u, h->shade_grid );
            ACN_LAUNCH( h, 1, part_stream, acn_launch_shade1( f, qg, part_stream, s, h->d_accum, h->d_counters, part ) );
        }
        if( fork )
        {
            HIP_TRY( hipEventRecord( h->ev_path, stream ) );
            HIP_TRY( hipStreamWaitEvent( h->side_stream, h->ev_path, 0 ) );
        }
        qg.grid = learned_grid( h, h->seen_hs[ level ], cnt, 256u, h->grid );
        ACN_LAUNCH( h, 3, direct_stream, acn_launch_hard_shadow( f, qg, lds, direct_stream, s, h->d_accum, h->d_counters ) );
        if( level + 1 < levels )   /* the last level casts no path rays (depth <= 10) */
        {
            qg.grid = learned_grid( h, h->seen_hp[ level ], cnt, 256u, h->grid );
            ACN_LAUNCH( h, 3, stream, acn_launch_hard_path( f, qg, lds, stream, s, h->d_accum, h->d_counters ) );
        }
        if( fork )
        {
            HIP_TRY( hipEventRecord( h->ev_join, h->side_stream ) );
            HIP_TRY( hipStreamWaitEvent( stream, h->ev_join, 0 ) );
        }
    }
    HIP_TRY( hipMemcpyAsync( h->h_counts, h->d_counts, sizeof( uint32_t ) * QC_N * levels, hipMemcpyDeviceToHost, stream ) );
    HIP_TRY( hipStreamSynchronize( stream ) );
    h->host_syncs++;
    uint32_t flags = 0;
    for( int level = 0; level < levels; level++ )
    {
        const uint32_t* c = h->h_counts + ( size_t )level * QC_N;
        flags |= c[ QC_FLAGS ];
        if( c[ QC_GEN + walk_passes_of_level( h, level ) ] ) flags |= ACN_FLAG_CHILD_OVERFLOW;   /* rays left over by the last pass */
    }
    h->flags_seen |= flags & ACN_FLAG_CLAMPED;
    /* what the chunk put into each queue (high-water marks of reserved slots, dead slots included; of a chunk that
     * overflowed: at least this much): the next chunk's size and the queue capacities are derived from it */
    for( int level = 0; level < levels; level++ )
    {
        const uint32_t* c = h->h_counts + ( size_t )level * QC_N;
        auto up = [ & ]( int q, uint32_t v ) { if( v > fill[ q ] ) fill[ q ] = v; };
        up( WQ_TASKS, c[ QC_TASKS ] );
        for( int k = 0; k < ACN_NCLASS; k++ ) up( WQ_TASKS, c[ QC_CLASS0 + k ] );
        up( WQ_CHILDREN, c[ QC_CHILDREN ] );
        up( WQ_HARD_SHADOW, c[ QC_HARD_SHADOW ] );
        up( WQ_HARD_PATH, c[ QC_HARD_PATH ] );
        for( int g = 0; g <= ACN_MAX_WALK_PASSES; g++ ) up( WQ_RAYS, c[ QC_GEN + g ] );
    }
    {
        /* how much of the marks are dead slots (the ends of the waves' reservations): known exactly for the deferred-shadow
         * queue, whose records are counted; the share does not scale with the chunk, so a small chunk's marks over-state
         * the demand per position by 1 / ( 1 - share ) */
        uint32_t mark = 0, recs = 0;
        for( int level = 0; level < levels; level++ )
        {
            const uint32_t* c = h->h_counts + ( size_t )level * QC_N;
            if( c[ QC_HARD_SHADOW ] > mark ) { mark = c[ QC_HARD_SHADOW ]; recs = c[ QS_HARD_SHADOW ] + c[ QS_PROBES ]; }
        }
        if( mark > 0 && recs < mark ) *dead_share = ( double )( mark - recs ) / ( double )mark;
    }
    /* a lost chunk first: it is redone smaller, and a stack overflow that is real shows again in the retry */
    if( flags & ( ACN_FLAG_TASK_OVERFLOW | ACN_FLAG_CHILD_OVERFLOW ) ) { *overflow = 1; return ACN_OK; }
    if( flags & ACN_FLAG_STACK_OVERFLOW ) return fail( ACN_ERR_UNSUPPORTED, "device CSG / compound stack overflow (or a walk that did not end)" );
    for( int level = 0; level < levels; level++ )
    {
        const uint32_t* c = h->h_counts + ( size_t )level * QC_N;
        if( c[ QC_TASKS ] == 0 && c[ QS_WALK_RAYS ] == 0 ) break;
        h->levels++;
        h->walk_rays += c[ QS_WALK_RAYS ];
        h->walk_steps += c[ QS_WALK_STEPS ];
        h->hard_rays += ( uint64_t )c[ QS_HARD_SHADOW ] + c[ QS_HARD_PATH ];
        h->shade_hit_recs += c[ QS_CHILDREN ];
        if( c[ QC_TASKS ] > h->peak_tasks ) h->peak_tasks = c[ QC_TASKS ];
        if( c[ QC_CHILDREN ] > h->peak_children ) h->peak_children = c[ QC_CHILDREN ];
        h->private_rays += c[ QS_PRIVATE_RAYS ];
        h->probe_rays += c[ QS_PROBES ];
    }
    if( cnt >= 4096 )   /* a chunk large enough to stand for the next one */
    {
        uint32_t seen[ ACN_MAX_PATH_LEVELS + 1 ];
        for( int level = 0; level < levels; level++ )
        {
            const uint32_t* c = h->h_counts + ( size_t )level * QC_N;
            const uint32_t launched = walk_passes_of_level( h, level );
            uint32_t used = 1;   /* pass 0 of level 0 has the camera rays; a level without rays keeps one launch */
            for( uint32_t g = 0; g < launched; g++ ) if( c[ QC_GEN + g ] ) used = g + 1;
            /* the last launch ran in private mode: if it still had input the level may need more passes than were launched */
            seen[ level ] = ( used == launched && launched > 1 ) ? used + 2 : used;
        }
        for( int level = 0; level < levels; level++ ) h->walk_passes_seen[ level ] = seen[ level ];
        h->seen_cnt = cnt;
        for( int level = 0; level < levels; level++ )
        {
            const uint32_t* c = h->h_counts + ( size_t )level * QC_N;
            for( int k = 0; k < ACN_NCLASS; k++ ) h->seen_class[ level ][ k ] = c[ QC_CLASS0 + k ];
            h->seen_hs[ level ] = c[ QC_HARD_SHADOW ]; h->seen_hp[ level ] = c[ QC_HARD_PATH ];
            h->seen_hits[ level ] = level > 0 ? h->h_counts[ ( size_t )( level - 1 ) * QC_N + QC_CHILDREN ] : 0u;
            for( int g = 0; g <= ACN_MAX_WALK_PASSES; g++ ) h->seen_gen[ level ][ g ] = c[ QC_GEN + g ];
        }
    }
    return ACN_OK;
}

/* the rates of a handle from the queue marks of one chunk of cnt positions */
static void set_rates( acn_scene_handle* h, uint32_t cnt, const uint32_t* fill, double dead_share )
{
    /* (a chunk of a few positions: mostly dead slots, 64 positions of hanging_lamp p1024 mark 15 600 deferred rays per position
     * where 3 500 is the rate -- the counted share of the deferred-shadow queue corrects all five) */
    const double live = cnt <= 4096 && dead_share > 0 && dead_share < 0.95 ? 1.0 - dead_share : 1.0;
    for( int q = 0; q < WQ_N; q++ ) h->rate[ q ] = f_max_host( live * ( double )fill[ q ] / ( double )cnt, 1e-3 );
    h->rate_cnt = cnt;
}

/* Cold handle: the queue demand per position is learned from a SAMPLE of the call's own positions -- every ( n / m )-th of them,
 * m = 512 .. 4096 -- rendered once on the starter queues and thrown away, before anything is sized.  Round 3 let the first
 * chunks of the call learn: the first tile of the order is a corner of the picture (C4: 1 shading task and 161 deferred path rays
 * per position where the frame's average is 300 / 1 800), so the queues were found one by one by halving -- 12 redone chunks on
 * the C3 / C4 frames, a first frame of 718 ms on paraffin_lamp where the second takes 465 -- and every lane of a call learned
 * for itself and re-sized its queues in the middle of the frame.  A sample over the whole frame costs one short chain of
 * launches (a few ms; nothing next to a frame whose queues must be allocated anyway) and is trusted like a large chunk: the
 * queues are then sized ONCE, while the device is idle (launch_render; render_lanes for all lanes of a call). */
static int learn_rates( acn_scene_handle* h, const double* d_pos_xy, size_t first, size_t n, hipStream_t stream, size_t plan_positions, unsigned plan_grid )
{
    if( rates_known( h ) || !h->tun.learn_sample || h->tun.chunk || h->tun.ws_uniform || n < 16384 ) return ACN_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [ & ]() { return std::chrono::duration< double, std::milli >( std::chrono::steady_clock::now() - t_begin ).count(); };
    int st = ensure_workspace( h, 4096 );   /* the starter set */
    if( st != ACN_OK ) return st;
    if( h->accum_cap < n )
    {
        if( h->d_accum ) hipFree( h->d_accum );
        h->d_accum = nullptr; h->accum_cap = 0;
        HIP_TRY( hipMalloc( &h->d_accum, sizeof( unsigned long long ) * 3 * n ) );
        h->accum_cap = n;
    }
    if( h->tun.debug_chunks ) fprintf( stderr, "[acn sample] starter queues (%.2f GB) after %.2f ms\n", ( double )h->ws.bytes / 1e9, since() );
    /* as many positions as the starter queues hold by the guess launch_render makes for a first chunk, 4096 at most */
    const size_t s = h->dev.prm.path_samples ? h->dev.prm.path_samples : 1;
    size_t want = ( size_t )( ( double )h->ws.cap[ WQ_CHILDREN ] / ( ( double )( s + 2 ) * ( s > 64 ? ( double )s / 64.0 : 1.0 ) ) );
    const size_t by_shadow = ( size_t )( ( double )h->ws.cap[ WQ_HARD_SHADOW ] / ( 0.25 * ( double )( h->dev.prm.direct_samples * h->n_lights + s ) + 4.0 ) );
    if( want > by_shadow ) want = by_shadow;
    if( want > 4096 ) want = 4096;
    /* (the guess is ten times what the lamp scenes need at path_samples 1024, where it allowed 63 positions: no sample at all, and
     * the whole hanging_lamp frame at stated size began every band with ~20 redone chunks, halving down from 92 000 positions to 9.
     * A sample that does not fit is halved below.) */
    if( want < 256 ) want = 256;
    if( want > n / 4 ) want = n / 4;
    const bool count_work = h->count_work, stage_timing = h->stage_timing;
    h->count_work = false; h->stage_timing = false; h->shard_rank = 0; h->shard_world = 1;
    h->events_used = 0;
    for( ; want >= 64; want /= 2 )
    {
        TileOrder order;
        order.n = ( uint32_t )n; order.n_tiles = 1; order.mul = 1;
        order.sample_stride = ( uint32_t )( n / want );
        const uint32_t cnt = ( uint32_t )want;
        hipLaunchKernelGGL( k_clear_slots, dim3( ( cnt + 255 ) / 256 ), dim3( 256 ), 0, stream, h->d_accum, 0u, cnt, order );
        HIP_TRY( hipGetLastError() );
        int overflow = 0;
        uint32_t fill[ WQ_N ];
        double dead_share = 0;
        st = render_chunk( h, d_pos_xy, first, 0u, cnt, order, stream, &overflow, fill, &dead_share );
        if( st != ACN_OK ) break;
        if( h->tun.debug_chunks )
            fprintf( stderr, "[acn sample] chain done after %.2f ms\n", since() );
        if( h->tun.debug_chunks )
            fprintf( stderr, "[acn sample] %u positions (every %u-th) %s dead %.2f | per pos T %.1f C %.1f HS %.1f HP %.1f R %.1f\n", cnt, order.sample_stride, overflow ? "OVERFLOW" : "ok",
                     dead_share, fill[ 0 ] / ( double )cnt, fill[ 1 ] / ( double )cnt, fill[ 2 ] / ( double )cnt, fill[ 3 ] / ( double )cnt, fill[ 4 ] / ( double )cnt );
        if( overflow ) continue;
        /* the records the sample left in each queue, exactly (marks minus dead slots; the fullest level counts, the queues are
         * the levels' in turn), plus a quarter for what a sample of a few thousand positions does not see */
        double live[ WQ_N ] = { 0, 0, 0, 0, 0 };
        for( int level = 0; level < h->n_levels; level++ )
        {
            const uint32_t* c = h->h_counts + ( size_t )level * QC_N;
            auto up = [ & ]( int q, double v ) { if( v > live[ q ] ) live[ q ] = v; };
            up( WQ_TASKS, ( double )c[ QC_TASKS ] - ( double )c[ QS_DEAD_T ] );
            up( WQ_CHILDREN, ( double )c[ QC_CHILDREN ] - ( double )c[ QS_DEAD_C ] );
            up( WQ_HARD_SHADOW, ( double )c[ QS_HARD_SHADOW ] + ( double )c[ QS_PROBES ] );
            up( WQ_HARD_PATH, ( double )c[ QC_HARD_PATH ] - ( double )c[ QS_DEAD_HP ] );
            /* rays: no generation holds more than the largest mark, nor more than all the level's generations together */
            double sum = 0, top = 0;
            for( int g = 0; g <= ACN_MAX_WALK_PASSES; g++ ) { sum += c[ QC_GEN + g ]; if( c[ QC_GEN + g ] > top ) top = c[ QC_GEN + g ]; }
            sum -= ( double )c[ QS_DEAD_R ];
            up( WQ_RAYS, sum < top ? sum : top );
        }
        /* What a queue must hold is records PLUS the slots that die at the ends of the waves' reservations: up to 64 per wave,
         * queue and launch that appends to it, whatever the chunk's size (a chunk of 230 000 positions of the wine glass marks 1.1 M
         * task slots for 0.45 M tasks).  The planner's rates are marks per position, so the dead slots of a chunk of the size the
         * call will run -- plan_positions, on persistent grids of plan_grid workgroups -- are spread over its positions:
         * tasks, specular rays and probes are appended by k_shade_hits and ~4 walk passes, path-sample hits and the two deferred
         * queues by the four k_shade launches (and k_hard_path). */
        {
            const double per_launch = ( double )ACN_QCHUNK * 4.0 * ( double )plan_grid;
            const double walkers = 3.0 * per_launch, shaders = 3.0 * per_launch;   /* (not every wave of every launch leaves a full reservation behind) */
            const double dead[ WQ_N ] = { walkers, shaders, walkers + shaders, shaders, walkers };
            const double pp = ( double )( plan_positions < ACN_CHUNK_TARGET ? plan_positions : ACN_CHUNK_TARGET );
            /* a generation of specular rays is at most three children per shaded hit (path-sample hits of the level before, or the
             * camera rays' shading points): the sample's ray marks are mostly dead slots */
            const double ray_bound = 2.0 * live[ WQ_CHILDREN ] + live[ WQ_TASKS ];
            if( live[ WQ_RAYS ] > ray_bound && ray_bound > 0 ) live[ WQ_RAYS ] = ray_bound;
            for( int q = 0; q < WQ_N; q++ ) h->rate[ q ] = f_max_host( 1.2 * live[ q ] / ( double )cnt + dead[ q ] / pp, 1e-3 );
        }
        if( h->tun.debug_chunks ) fprintf( stderr, "[acn sample] rates T %.1f C %.1f HS %.1f HP %.1f R %.1f\n", h->rate[ 0 ], h->rate[ 1 ], h->rate[ 2 ], h->rate[ 3 ], h->rate[ 4 ] );
        h->rate_cnt = 8192;   /* a sample of the whole frame: trusted like a chunk that size (launch_render re-sizes for the whole rest at once) */
        break;
    }
    for( int level = 0; level <= ACN_MAX_PATH_LEVELS; level++ ) h->walk_passes_seen[ level ] = 0;
    h->seen_cnt = 0;
    h->count_work = count_work; h->stage_timing = stage_timing;
    return st;
}

static int launch_render( acn_scene_handle* h, const double* d_pos_xy, size_t first, size_t n, double* d_out_rgb,
                          const acn_render_opts* opts, hipStream_t stream )
{
    if( opts && opts->cancel && *opts->cancel ) return fail( ACN_ERR_CANCELLED, "cancelled" );
    if( n == 0 ) return ACN_OK;
    if( n > 0xFFFFFF00ull ) return fail( ACN_ERR_ARG, "too many positions in one call" );
    int linear = ( opts && ( opts->flags & ACN_OPT_LINEAR_OUT ) ) ? 1 : 0;
    h->count_work = ( opts && ( opts->flags & ACN_OPT_COUNT_WORK ) ) || h->tun.count_work;
    h->shard_rank = 0; h->shard_world = 1;
    if( opts && opts->shard_mode == ACN_SHARD_SAMPLES && opts->shard_world > 1 )
    {
        if( opts->shard_rank >= opts->shard_world ) return fail( ACN_ERR_ARG, "shard_rank >= shard_world" );
        h->shard_rank = opts->shard_rank; h->shard_world = opts->shard_world;
    }
    else if( opts && opts->shard_mode > ACN_SHARD_SAMPLES ) return fail( ACN_ERR_ARG, "unknown shard_mode" );
    h->stage_timing = ( opts && ( opts->flags & ACN_OPT_STAGE_TIMING ) ) || h->tun.stage_timing;
    int st = learn_rates( h, d_pos_xy, first, n, stream, n, h->walk_grid > h->grid ? h->walk_grid : h->grid );
    if( st != ACN_OK ) return st;
    /* (shard fields again: the learning pass renders unsharded) */
    if( opts && opts->shard_mode == ACN_SHARD_SAMPLES && opts->shard_world > 1 ) { h->shard_rank = opts->shard_rank; h->shard_world = opts->shard_world; }
    st = ensure_workspace( h, n );
    if( st != ACN_OK ) return st;
    if( h->accum_cap < n )
    {
        if( h->d_accum ) hipFree( h->d_accum );
        h->d_accum = nullptr; h->accum_cap = 0;
        HIP_TRY( hipMalloc( &h->d_accum, sizeof( unsigned long long ) * 3 * n ) );
        h->accum_cap = n;
    }
    h->events_used = 0;
    h->launches[ 0 ] = h->launches[ 1 ] = h->launches[ 2 ] = h->launches[ 3 ] = 0;
    h->hard_rays = 0; h->walk_steps = 0; h->walk_rays = 0; h->shade_hit_recs = 0; h->host_syncs = 0; h->flags_seen = 0; h->private_rays = 0; h->probe_rays = 0;
    h->chunks = h->retries = h->levels = 0;
    h->ctl.retry_bound = 0;   /* (a call that ended in the middle of a retry) */
    h->peak_tasks = h->peak_children = 0;
    HIP_TRY( hipMemsetAsync( h->d_counters, 0, sizeof( unsigned long long ) * ACN_CNT_SLOTS, stream ) );
    HIP_TRY( hipEventRecord( h->ev0, stream ) );
    HIP_TRY( hipMemsetAsync( h->d_accum, 0, sizeof( unsigned long long ) * 3 * n, stream ) );

    /* Positions per pipeline run.  How many records a position leaves in each queue differs by orders of magnitude between
     * scenes (wine_glass: 15 deferred shadow rays per pixel; a closed room at path_samples 1024: 260 000 second-level hits),
     * so the rates are learned: a cautious first chunk on a small starter workspace, then chunks that fill the fullest
     * queue to 70 %, and the queues themselves re-sized once the rates are known (ensure_workspace); an overflow halves
     * the chunk.  Rates and workspace stay with the handle for its next call. */
    size_t s = h->dev.prm.path_samples ? h->dev.prm.path_samples : 1;
    size_t chunk;
    if( rates_known( h ) ) chunk = chunk_for_caps( h );
    else
    {
        /* the starter queues are small: a first chunk of at most 32 768 positions, fewer by a guess that errs on the safe
         * side by factors, not orders of magnitude (an overflow costs one small chunk): path-sample hits ~ path_samples per
         * position, squared from 64 samples on (two nested levels); a quarter of the direct-light samples deferred */
        chunk = ( size_t )( ( double )h->ws.cap[ WQ_CHILDREN ] / ( ( double )( s + 2 ) * ( s > 64 ? ( double )s / 64.0 : 1.0 ) ) );
        const size_t by_shadow = ( size_t )( ( double )h->ws.cap[ WQ_HARD_SHADOW ] / ( 0.25 * ( double )( h->dev.prm.direct_samples * h->n_lights + s ) + 4.0 ) );
        if( chunk > by_shadow ) chunk = by_shadow;
        if( chunk > 32768 ) chunk = 32768;
    }
    if( h->tun.chunk ) chunk = h->tun.chunk;
    if( chunk < 64 ) chunk = 64;
    /* the order of work: tiles of 256 positions in a multiplicative stride over the call (TileOrder) */
    TileOrder order;
    order.n = ( uint32_t )n;
    order.n_tiles = ( uint32_t )( ( n + ( ( 1u << ACN_ORDER_SHIFT ) - 1 ) ) >> ACN_ORDER_SHIFT );
    order.mul = 1;
    order.sample_stride = 0;
    if( order.n_tiles > 2 )
    {
        auto gcd = []( uint64_t a, uint64_t b ) { while( b ) { uint64_t t = a % b; a = b; b = t; } return a; };
        uint64_t m = ( uint64_t )( 0.6180339887 * order.n_tiles ) | 1u;
        while( gcd( m, order.n_tiles ) != 1 ) m += 2;
        order.mul = ( uint32_t )( m % order.n_tiles );
    }
    const size_t n_slots = ( size_t )order.n_tiles << ACN_ORDER_SHIFT;
    size_t base = 0;
    while( base < n_slots )
    {
        if( opts && opts->cancel && *opts->cancel ) return fail( ACN_ERR_CANCELLED, "cancelled" );
        /* the planned chunk; a rest that is predicted to fill no queue beyond 85 % is taken whole (a second chunk would be
         * another whole chain of launches for a few positions); a retry is at most half of the chunk that overflowed
         * (acn_chunkplan.h) */
        uint32_t cnt = acn_ctl_next( &h->ctl, n_slots - base, chunk, h->tun.chunk != 0, rates_known( h ), h->rate, h->ws.cap );
        int overflow = 0;
        uint32_t fill[ WQ_N ];
        double dead_share = 0;
        /* the work counters of a chunk that has to be redone must not count twice */
        if( h->count_work ) HIP_TRY( hipMemcpyAsync( h->d_counters_keep, h->d_counters, sizeof( unsigned long long ) * ACN_CNT_SLOTS, hipMemcpyDeviceToDevice, stream ) );
        st = render_chunk( h, d_pos_xy, first, ( uint32_t )base, cnt, order, stream, &overflow, fill, &dead_share );
        if( st != ACN_OK ) return st;
        if( h->tun.debug_chunks )
            fprintf( stderr, "[acn chunk] base %zu cnt %u %s target %.2f dead %.2f | fill T %u C %u HS %u HP %u R %u | per pos T %.1f C %.1f HS %.1f HP %.1f R %.1f | rate T %.1f C %.1f HS %.1f HP %.1f R %.1f | cap T %u C %u HS %u HP %u R %u\n",
                     base, cnt, overflow ? "OVERFLOW" : "ok", h->ctl.fill_target, dead_share, fill[ 0 ], fill[ 1 ], fill[ 2 ], fill[ 3 ], fill[ 4 ],
                     fill[ 0 ] / ( double )cnt, fill[ 1 ] / ( double )cnt, fill[ 2 ] / ( double )cnt, fill[ 3 ] / ( double )cnt, fill[ 4 ] / ( double )cnt,
                     h->rate[ 0 ], h->rate[ 1 ], h->rate[ 2 ], h->rate[ 3 ], h->rate[ 4 ], h->ws.cap[ 0 ], h->ws.cap[ 1 ], h->ws.cap[ 2 ], h->ws.cap[ 3 ], h->ws.cap[ 4 ] );
        if( overflow )
        {
            if( cnt <= 1 ) return fail( ACN_ERR_DEVICE, "work queues overflow for a single position: raise ACN_WORKSPACE_MB" );
            if( h->count_work ) HIP_TRY( hipMemcpyAsync( h->d_counters, h->d_counters_keep, sizeof( unsigned long long ) * ACN_CNT_SLOTS, hipMemcpyDeviceToDevice, stream ) );
            h->retries++;
            /* scenes whose demand per position varies much between chunks (many_spheres p256: 49 of 220 chunks were redone at
             * a fixed 70 %) plan with more head room */
            chunk = acn_ctl_overflow( &h->ctl, cnt );
            /* the marks of an overflowed chunk are lower bounds of its demand */
            for( int q = 0; q < WQ_N; q++ ) { const double r = ( double )fill[ q ] / ( double )cnt; if( r > h->rate[ q ] ) h->rate[ q ] = r; }
            for( int level = 0; level <= ACN_MAX_PATH_LEVELS; level++ ) h->walk_passes_seen[ level ] = 0;   /* the full number of passes again */
            h->seen_cnt = 0;                                                                                 /* ... and full grids */
            hipLaunchKernelGGL( k_clear_slots, dim3( ( cnt + 255 ) / 256 ), dim3( 256 ), 0, stream, h->d_accum, ( uint32_t )base, cnt, order );
            HIP_TRY( hipGetLastError() );
            continue;
        }
        h->chunks++;
        base += cnt;
        acn_ctl_fit( &h->ctl );
        if( h->tun.chunk ) continue;
        /* Learn.  A chunk much larger than the one the rates came from replaces them (the dead slots at the ends of the
         * waves' queue reservations do not scale with the chunk, so small chunks over-estimate); otherwise the rates
         * follow upwards at once and forget slowly. */
        const bool known = rates_known( h );
        if( !known || cnt >= 4 * h->rate_cnt ) set_rates( h, cnt, fill, dead_share );
        else
        {
            /* (also after small chunks: where chunks are small the demand per position is large and the dead slots do not
             * matter; rates that only went up left many_spheres p256 with 532 chunks of 3 900 positions after one spike) */
            for( int q = 0; q < WQ_N; q++ ) h->rate[ q ] = f_max_host( f_max_host( ( double )fill[ q ] / ( double )cnt, 0.85 * h->rate[ q ] ), 1e-3 );
            if( cnt > h->rate_cnt ) h->rate_cnt = cnt;
        }
        const size_t remaining = n_slots - base;
        if( remaining && ( double )chunk_for_caps( h ) * ( 0.85 / h->ctl.fill_target ) < ( double )remaining )
        {
            /* more than one further chunk with these queues: re-size them (a no-op when they already are what the budget
             * allows).  Rates that come from a small chunk are trusted for a medium one only. */
            const size_t target = h->rate_cnt < 8192 ? ( remaining < 65536 ? remaining : ( size_t )65536 ) : remaining;
            if( ( st = ensure_workspace( h, target ) ) != ACN_OK ) return st;
        }
        chunk = chunk_for_caps( h );
    }
    if( ( st = stage_begin( h, 2, stream ) ) != ACN_OK ) return st;
    hipLaunchKernelGGL( k_finalize, dim3( ( unsigned )( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, stream,
                        ( const unsigned long long* )h->d_accum, ( uint32_t )n, h->dev.prm.gamma, linear, d_out_rgb );
    HIP_TRY( hipGetLastError() );
    if( ( st = stage_end( h, stream ) ) != ACN_OK ) return st;
    HIP_TRY( hipEventRecord( h->ev1, stream ) );
    h->timed = true;
    return ACN_OK;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* Concurrent lanes.  One pipeline run is a chain of ~60 dependent launches (a walk pass per specular generation,
 * shade, hard rays, per level), each ending in a tail where a few long rays keep the chip waiting, and each followed by
 * a host round trip for the queue counts.  Pixels are independent, so a call is cut into ACN_LANE_TILE-pixel tiles
 * dealt round-robin to K lanes; every lane is a clone of the handle (same resident scene, own stream, own workspace)
 * driven by its own host thread, and the lanes' kernels fill each other's tails and bubbles.  Measured on the 1080p
 * frame: 118 -> 87 ms with 4 lanes; on the share one of 8 GPUs gets: 21.0 -> 16.5 ms.  Results are unchanged: every
 * pixel is computed by exactly the same kernels from exactly the same inputs. */
#define ACN_LANE_TILE 256

/* positions of lane `lane` of `lanes`: tiles lane, lane + lanes, ... of the n positions of the call */
static size_t lane_count( size_t n, int lanes, int lane )
{
    size_t tiles = ( n + ACN_LANE_TILE - 1 ) / ACN_LANE_TILE, cnt = 0;
    if( tiles == 0 ) return 0;
    size_t full = tiles / lanes, rest = tiles % lanes;
    size_t my_tiles = full + ( ( size_t )lane < rest ? 1 : 0 );
    cnt = my_tiles * ACN_LANE_TILE;
    size_t last_tile = tiles - 1;
    if( last_tile % lanes == ( size_t )lane ) cnt -= tiles * ACN_LANE_TILE - n;   /* the last tile may be short */
    return cnt;
}

__device__ __forceinline__ size_t lane_global_index( size_t i, int lanes, int lane )
{
    return ( ( i / ACN_LANE_TILE ) * lanes + lane ) * ACN_LANE_TILE + ( i % ACN_LANE_TILE );
}

/* lane_pos[ i ] = position of the lane's i-th pixel (taken from pos_xy, or generated like acn_render_main_pass_dev) */
__global__ void k_lane_gather( const double* __restrict__ pos_xy, size_t first_pixel, uint64_t image_width, size_t n_lane,
                               int lanes, int lane, double* __restrict__ lane_pos )
{
    size_t i = ( size_t )blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n_lane ) return;
    size_t g = lane_global_index( i, lanes, lane );
    double mx, my;
    if( pos_xy ) { mx = pos_xy[ g * 2 ]; my = pos_xy[ g * 2 + 1 ]; }
    else
    {
        size_t pix = first_pixel + g;
        mx = ( double )( pix % image_width ) + 0.5;
        my = ( double )( pix / image_width ) + 0.5;
    }
    lane_pos[ i * 2 ] = mx; lane_pos[ i * 2 + 1 ] = my;
}

__global__ void k_lane_scatter( const double* __restrict__ lane_out, size_t n_lane, int lanes, int lane, double* __restrict__ out_rgb )
{
    size_t i = ( size_t )blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n_lane ) return;
    size_t g = lane_global_index( i, lanes, lane );
    out_rgb[ g * 3 ] = lane_out[ i * 3 ]; out_rgb[ g * 3 + 1 ] = lane_out[ i * 3 + 1 ]; out_rgb[ g * 3 + 2 ] = lane_out[ i * 3 + 2 ];
}

/* A lane = a clone of the handle that borrows the resident scene and owns two streams, its events, counter blocks and a host
 * thread.  Making a stream takes ~10 ms of host time (tools/bench_alloc: 12 streams 120 - 130 ms, one after the other whatever thread
 * asks; events, pinned memory and hipMalloc of any size are free beside that), so six lanes with two streams each were 60 - 100 ms of
 * a handle's first call, more than its learning pass on the wine glass.  The side stream is made only where it is used
 * (ACN_SHADE_FISSION), and so the HIP objects (lane_objects: nothing in it reads the parent) are made on a helper
 * thread while the learning pass runs on the device (render_lanes), and the parent's fields are copied afterwards (bind_lane). */
static int lane_objects( int device, bool side_stream, bool debug, acn_scene_handle** out )
{
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [ & ]() { return std::chrono::duration< double, std::milli >( std::chrono::steady_clock::now() - t_begin ).count(); };
    double t[ 5 ] = { 0, 0, 0, 0, 0 };
    acn_scene_handle* l = new acn_scene_handle();
    l->is_lane = true;
    l->device = device;
#define HIP_TRY_L( expr ) do { hipError_t e_ = ( expr ); if( e_ != hipSuccess ) { acn_scene_free( l ); return fail( ACN_ERR_DEVICE, hipGetErrorString( e_ ) ); } } while( 0 )
    HIP_TRY_L( hipSetDevice( device ) );
    t[ 0 ] = since();
    HIP_TRY_L( hipStreamCreateWithFlags( &l->stream, hipStreamNonBlocking ) );
    t[ 1 ] = since();
    if( side_stream ) HIP_TRY_L( hipStreamCreateWithFlags( &l->side_stream, hipStreamNonBlocking ) );
    HIP_TRY_L( hipEventCreate( &l->ev0 ) );
    HIP_TRY_L( hipEventCreate( &l->ev1 ) );
    HIP_TRY_L( hipEventCreateWithFlags( &l->ev_fork, hipEventDisableTiming ) );
    HIP_TRY_L( hipEventCreateWithFlags( &l->ev_path, hipEventDisableTiming ) );
    HIP_TRY_L( hipEventCreateWithFlags( &l->ev_join, hipEventDisableTiming ) );
    t[ 2 ] = since();
    HIP_TRY_L( hipMalloc( &l->d_counters, sizeof( unsigned long long ) * ACN_CNT_SLOTS ) );
    HIP_TRY_L( hipMalloc( &l->d_counters_keep, sizeof( unsigned long long ) * ACN_CNT_SLOTS ) );
    HIP_TRY_L( hipMalloc( &l->d_counts, sizeof( uint32_t ) * QC_N * ACN_LEVEL_BLOCKS ) );
    /* on the lane's own stream, where everything that uses them follows (the null stream would wait for the caller's) */
    HIP_TRY_L( hipMemsetAsync( l->d_counters, 0, sizeof( unsigned long long ) * ACN_CNT_SLOTS, l->stream ) );
    HIP_TRY_L( hipMemsetAsync( l->d_counts, 0, sizeof( uint32_t ) * QC_N * ACN_LEVEL_BLOCKS, l->stream ) );
    t[ 3 ] = since();
    HIP_TRY_L( hipHostMalloc( &l->h_counts, sizeof( uint32_t ) * QC_N * ACN_LEVEL_BLOCKS ) );
    t[ 4 ] = since();
#undef HIP_TRY_L
    l->worker = new LaneWorker();
    l->worker->start();
    if( debug ) fprintf( stderr, "[acn lane] set device %.2f ms, stream %.2f, events %.2f, counter blocks + memsets %.2f, pinned block %.2f, thread %.2f\n", t[ 0 ], t[ 1 ] - t[ 0 ], t[ 2 ] - t[ 1 ], t[ 3 ] - t[ 2 ], t[ 4 ] - t[ 3 ], since() - t[ 4 ] );
    *out = l;
    return ACN_OK;
}

static unsigned lane_grid( const acn_scene_handle* parent ) { return parent->tun.grid ? parent->tun.grid : parent->cus * 1u; }
static void bind_lane( const acn_scene_handle* parent, int lanes, acn_scene_handle* l )
{
    l->budget_div = ( size_t )lanes;
    l->dev = parent->dev;
    l->d_nodes = parent->d_nodes; l->d_mats = parent->d_mats; l->d_elems = parent->d_elems; l->d_textures = parent->d_textures;
    l->scene_bytes[ 0 ] = parent->scene_bytes[ 0 ]; l->scene_bytes[ 1 ] = parent->scene_bytes[ 1 ]; l->scene_bytes[ 2 ] = parent->scene_bytes[ 2 ]; l->scene_bytes[ 3 ] = parent->scene_bytes[ 3 ];
    l->max_csg_depth = parent->max_csg_depth;
    l->lds_bytes = parent->lds_bytes; l->lds_stack_bytes = parent->lds_stack_bytes;
    l->prune = parent->prune; l->leaf_lights = parent->leaf_lights;
    l->tun = parent->tun; l->cus = parent->cus; l->n_levels = parent->n_levels;
    l->workspace_budget = parent->workspace_budget; l->n_lights = parent->n_lights;
    /* Round 4: six lanes on grids of ONE workgroup per CU (k_shade: one and a half) instead of four lanes on two.  With k_walk at
     * four waves per SIMD a grid of 256 workgroups is resident at once, and six shorter chains fill each other's tails better than
     * four: 1080p 50.2 -> 49.1 ms, c2 26.4 -> 25.0, and the share one of 8 GPUs gets 12.25 -> 11.4 ms (profiles/r04/ab_lanes6_*).
     * A call that runs ALONE on the handle keeps four workgroups per CU (diamond on one lane: 2.3 s with them, 6.7 s with one). */
    l->grid = lane_grid( parent );
    l->shade_grid = parent->tun.shade_grid ? parent->tun.shade_grid : parent->cus * 3u / 2u;
    l->walk_grid = parent->tun.walk_grid ? parent->tun.walk_grid : l->grid;
    l->dev.flags = l->d_counts + QC_FLAGS;
}

/* number of lanes for a call of n positions: the handle's ACN_LANES, fewer while a lane would get less than 32 tiles or
 * less than ~10^6 path samples' worth of work (a frame without path tracing is over before a second lane has started) */
static int lanes_for_counts( int tun_lanes, size_t n, uint64_t path_samples )
{
    int lanes = tun_lanes;
    const size_t work = n * ( size_t )( path_samples + 1 );
    while( lanes > 1 && ( n < ( size_t )lanes * 32 * ACN_LANE_TILE || work < ( size_t )lanes << 20 ) ) lanes--;
    return lanes;
}
static int lanes_for( const acn_scene_handle* h, size_t n ) { return lanes_for_counts( h->tun.lanes, n, h->dev.prm.path_samples ); }

static int render_lanes( acn_scene_handle* h, int lanes, const double* d_pos_xy, size_t first, size_t n, double* d_out_rgb,
                         const acn_render_opts* opts, hipStream_t stream )
{
    /* ACN_DEBUG_CHUNKS: where a call's wall time goes before and after the lanes run (one line per call on stderr) */
    const auto t_begin = std::chrono::steady_clock::now();
    double t_mark[ 5 ] = { 0, 0, 0, 0, 0 };
    auto mark = [ & ]( int i ) { t_mark[ i ] = std::chrono::duration< double, std::milli >( std::chrono::steady_clock::now() - t_begin ).count(); };
    /* the lanes this call lacks: made on a helper thread while the learning pass of a cold handle runs (see lane_objects) */
    early_lanes_join( h );
    for( acn_scene_handle* l : h->early_made ) { bind_lane( h, lanes, l ); h->lanes.push_back( l ); }   /* made during the upload */
    h->early_made.clear();
    const int missing = lanes - ( int )h->lanes.size();
    std::vector< acn_scene_handle* > made;
    int made_status = ACN_OK; std::string made_message;
    auto make_missing = [ & ]()
    {
        for( int k = 0; k < missing && made_status == ACN_OK; k++ )
        {
            acn_scene_handle* l = nullptr;
            made_status = lane_objects( h->device, h->tun.shade_fission, h->tun.debug_chunks, &l );
            if( made_status == ACN_OK ) made.push_back( l ); else made_message = g_last_error;   /* thread-local where it was set */
        }
    };
    std::thread maker;
    const bool learn = h->lanes.empty() ? !rates_known( h ) : !rates_known( h->lanes[ 0 ] ) && !rates_known( h );
    const bool maker_used = missing > 0 && learn && h->tun.cold_pipeline;
    if( missing > 0 ) { if( maker_used ) maker = std::thread( make_missing ); else make_missing(); }
    /* what the caller queued on `stream` before this call must be done before the lanes read the positions */
    hipError_t drained = hipEventRecord( h->ev0, stream );
    if( drained == hipSuccess ) drained = hipEventSynchronize( h->ev0 );
    mark( 0 );
    /* a cold handle learns the scene's queue demand once, for all lanes, from a sample of the call (learn_rates) */
    int learned = ACN_OK;
    if( drained == hipSuccess && learn )
    {
        h->budget_div = 1;
        learned = learn_rates( h, d_pos_xy, first, n, stream, n / ( size_t )lanes, lane_grid( h ) );
        if( learned == ACN_OK ) drained = hipStreamSynchronize( stream );
    }
    if( maker.joinable() ) maker.join();
    for( acn_scene_handle* l : made ) { bind_lane( h, lanes, l ); h->lanes.push_back( l ); }
    if( drained != hipSuccess ) return fail( ACN_ERR_DEVICE, hipGetErrorString( drained ) );
    if( learned != ACN_OK ) return learned;
    if( made_status != ACN_OK ) return fail( made_status, made_message );
    /* what one arrangement learned about the scene (records per position) holds for the other */
    for( int k = 0; k < lanes; k++ )
    {
        acn_scene_handle* l = h->lanes[ k ];
        const acn_scene_handle* from = rates_known( h ) ? h : h->lanes[ 0 ];
        if( rates_known( l ) || !rates_known( from ) ) continue;
        for( int q = 0; q < WQ_N; q++ ) l->rate[ q ] = from->rate[ q ];
        l->rate_cnt = from->rate_cnt; l->ctl.fill_target = from->ctl.fill_target;
        for( int level = 0; level <= ACN_MAX_PATH_LEVELS; level++ ) l->walk_passes_seen[ level ] = 0;
    }
    if( learn && rates_known( h ) ) free_workspace( h );   /* the bound is the handle's, whoever uses it */
    /* The lanes' queues are (re-)sized here, while the device is idle: hipFree synchronises the device, so lanes that
     * re-size at the start of their chains wait for each other's chunks (second frame of paraffin_lamp 400x600, whose
     * queues are trimmed to the rates the first frame learned: 2.1 s instead of 0.45, profiles/r03/frames_paraffin_*.txt).
     * hipMalloc itself is not what a first call pays: 22 GB of queues take 1 - 5 ms (tools/bench_alloc: 0.02 ms per GB; a lane that
     * started as soon as its own queues existed gained nothing, profiles/r04/first_frames_s36_s38.txt). */
    mark( 1 );
    acn_render_opts lane_opts{};
    if( opts ) lane_opts = *opts;
    std::vector< int > status( lanes, ACN_OK );
    std::vector< std::string > message( lanes );
    auto post_lane = [ & ]( int k )
    {
        h->lanes[ k ]->worker->post( [ &, k ]()
        {
            acn_scene_handle* l = h->lanes[ k ];
            l->budget_div = ( size_t )lanes;
            size_t cnt = lane_count( n, lanes, k );
            auto run = [ & ]() -> int
            {
                HIP_TRY( hipSetDevice( h->device ) );
                if( cnt == 0 ) { l->events_used = 0; HIP_TRY( hipMemset( l->d_counters, 0, sizeof( unsigned long long ) * ACN_CNT_SLOTS ) ); return ACN_OK; }
                if( l->lane_buf_cap < cnt )
                {
                    if( l->d_lane_pos ) hipFree( l->d_lane_pos );
                    if( l->d_lane_out ) hipFree( l->d_lane_out );
                    l->d_lane_pos = l->d_lane_out = nullptr; l->lane_buf_cap = 0;
                    HIP_TRY( hipMalloc( &l->d_lane_pos, sizeof( double ) * 2 * cnt ) );
                    HIP_TRY( hipMalloc( &l->d_lane_out, sizeof( double ) * 3 * cnt ) );
                    l->lane_buf_cap = cnt;
                }
                hipLaunchKernelGGL( k_lane_gather, dim3( ( unsigned )( ( cnt + 255 ) / 256 ) ), dim3( 256 ), 0, l->stream,
                                    d_pos_xy, first, ( uint64_t )h->dev.prm.image_width, cnt, lanes, k, l->d_lane_pos );
                HIP_TRY( hipGetLastError() );
                int st = launch_render( l, l->d_lane_pos, 0, cnt, l->d_lane_out, &lane_opts, l->stream );
                if( st != ACN_OK ) return st;
                hipLaunchKernelGGL( k_lane_scatter, dim3( ( unsigned )( ( cnt + 255 ) / 256 ) ), dim3( 256 ), 0, l->stream,
                                    ( const double* )l->d_lane_out, cnt, lanes, k, d_out_rgb );
                HIP_TRY( hipGetLastError() );
                HIP_TRY( hipStreamSynchronize( l->stream ) );
                return ACN_OK;
            };
            status[ k ] = run();
            if( status[ k ] != ACN_OK ) message[ k ] = g_last_error;   /* thread-local in the worker */
        } );
    };
    for( int k = 0; k < lanes; k++ )
    {
        acn_scene_handle* l = h->lanes[ k ];
        l->budget_div = ( size_t )lanes;
        const size_t cnt = lane_count( n, lanes, k );
        if( cnt ) { int st = ensure_workspace( l, cnt ); if( st != ACN_OK ) return st; }
    }
    for( int k = 0; k < lanes; k++ ) post_lane( k );
    mark( 2 );
    for( int k = 0; k < lanes; k++ ) h->lanes[ k ]->worker->wait();
    mark( 3 );
    if( h->tun.debug_chunks )
        fprintf( stderr, "[acn call] %zu positions on %d lanes: %d lanes made%s, caller's stream drained after %.2f ms, learning pass %.2f, queues sized %.2f, lanes done %.2f\n",
                 n, lanes, missing > 0 ? missing : 0, maker_used ? " beside the learning pass" : "", t_mark[ 0 ], t_mark[ 1 ] - t_mark[ 0 ], t_mark[ 2 ] - t_mark[ 1 ], t_mark[ 3 ] - t_mark[ 2 ] );
    for( int k = 0; k < lanes; k++ ) if( status[ k ] != ACN_OK ) return fail( status[ k ], message[ k ] );
    HIP_TRY( hipEventRecord( h->ev1, stream ) );
    /* statistics of the call: sums / maxima over the lanes */
    h->events_used = 0;
    h->launches[ 0 ] = h->launches[ 1 ] = h->launches[ 2 ] = h->launches[ 3 ] = 0;
    h->hard_rays = h->walk_steps = h->walk_rays = h->shade_hit_recs = h->host_syncs = h->private_rays = h->probe_rays = 0; h->flags_seen = 0;
    h->chunks = h->retries = h->levels = 0;
    h->ctl.retry_bound = 0;   /* (a call that ended in the middle of a retry) */
    h->peak_tasks = h->peak_children = 0;
    for( int k = 0; k < lanes; k++ )
    {
        const acn_scene_handle* l = h->lanes[ k ];
        if( lane_count( n, lanes, k ) == 0 ) continue;
        for( int i = 0; i < 4; i++ ) h->launches[ i ] += l->launches[ i ];
        h->hard_rays += l->hard_rays; h->walk_steps += l->walk_steps; h->walk_rays += l->walk_rays; h->shade_hit_recs += l->shade_hit_recs;
        h->host_syncs += l->host_syncs; h->flags_seen |= l->flags_seen; h->private_rays += l->private_rays; h->probe_rays += l->probe_rays;
        h->chunks += l->chunks; h->retries += l->retries;
        if( l->levels > h->levels ) h->levels = l->levels;
        h->peak_tasks += l->peak_tasks; h->peak_children += l->peak_children;
    }
    h->used_lanes = true;
    h->lanes_used = lanes;
    h->timed = true;
    return ACN_OK;
}

/* the caller's options as far as the caller's header knew them (acn_render_opts.struct_size), the rest zero */
static acn_render_opts opts_of( const acn_render_opts* in )
{
    acn_render_opts o{};
    if( in )
    {
        /* 0: a caller that zero-initialises the struct (the memset idiom) and never heard of struct_size -- the word was a
         * reserved zero in the first published layout, which already had the shard members: the 40-byte base layout */
        size_t n = in->struct_size ? in->struct_size : ( size_t )ACN_RENDER_OPTS_BASE_SIZE;
        if( n > sizeof( o ) ) n = sizeof( o );
        memcpy( &o, in, n );
    }
    o.struct_size = ( uint32_t )sizeof( o );
    return o;
}
#define ACN_OPTS_VIEW const acn_render_opts opts_seen_ = opts_of( opts ); opts = &opts_seen_;

/* one pipeline run on the handle itself, or the concurrent lanes */
static int render_dispatch( acn_scene_handle* h, const double* d_pos_xy, size_t first, size_t n, double* d_out_rgb,
                            const acn_render_opts* opts, hipStream_t stream )
{
    int lanes = lanes_for( h, n );
    /* Lanes pay when a lane's share is ONE chunk: their chains overlap.  A call whose queues cannot hold it in one chunk
     * per lane within the workspace bound -- scenes with hundreds or thousands of path samples -- does better on one lane
     * with the whole bound: four times the chunk, a quarter of the chains, and each chunk fills the chip by itself
     * (diamond 1080p p512, every 16th pixel: 3.69 s on 4 lanes, 2.83 s on one; hanging_lamp 2160p p1024, every 64th:
     * 21.6 -> 15.4 s; wine_glass 1080p p64, which fits: 71 ms on 4 lanes, 95 on one). */
    if( lanes > 1 )
    {
        const acn_scene_handle* known = nullptr;
        if( !h->lanes.empty() && rates_known( h->lanes[ 0 ] ) ) known = h->lanes[ 0 ];
        else if( rates_known( h ) ) known = h;
        if( known )
        {
            double need = 0;
            for( int q = 0; q < WQ_N; q++ ) need += known->rate[ q ] * ( double )n / 0.7 * ( double )wq_bytes[ q ];
            /* (sticky by 30 %: rates move a little from call to call, and changing the arrangement re-allocates everything) */
            if( need > ( h->one_lane ? 0.7 : 1.0 ) * ( double )h->workspace_budget ) lanes = 1;
        }
        else if( h->dev.prm.path_samples >= 256 ) lanes = 1;
    }
    h->used_lanes = false;
    h->one_lane = lanes <= 1 && lanes_for( h, n ) > 1;
    /* what one arrangement learned about the scene (records per position) holds for the other */
    auto inherit = []( acn_scene_handle* to, const acn_scene_handle* from )
    {
        if( rates_known( to ) || !rates_known( from ) ) return;
        for( int q = 0; q < WQ_N; q++ ) to->rate[ q ] = from->rate[ q ];
        to->rate_cnt = from->rate_cnt; to->ctl.fill_target = from->ctl.fill_target;
        for( int level = 0; level <= ACN_MAX_PATH_LEVELS; level++ ) to->walk_passes_seen[ level ] = 0;
    };
    if( lanes <= 1 )
    {
        for( acn_scene_handle* l : h->lanes ) free_workspace( l );   /* the bound is the handle's, whoever uses it */
        if( !h->lanes.empty() ) inherit( h, h->lanes[ 0 ] );
        return launch_render( h, d_pos_xy, first, n, d_out_rgb, opts, stream );
    }
    free_workspace( h );
    return render_lanes( h, lanes, d_pos_xy, first, n, d_out_rgb, opts, stream );   /* (makes the lanes it lacks) */
}

extern "C" int acn_render_positions_dev( acn_scene_handle* h, const void* d_pos_xy, size_t n, void* d_out_rgb,
                                         const acn_render_opts* opts )
{
    ACN_OPTS_VIEW
    if( !h || ( n && ( !d_pos_xy || !d_out_rgb ) ) ) return fail( ACN_ERR_ARG, "null argument" );
    HIP_TRY( hipSetDevice( h->device ) );
    hipStream_t stream = ( opts && opts->stream ) ? ( hipStream_t )opts->stream : h->stream;
    int st = render_dispatch( h, ( const double* )d_pos_xy, 0, n, ( double* )d_out_rgb, opts, stream );
    if( st != ACN_OK ) return st;
    if( !( opts && opts->stream ) ) HIP_TRY( hipStreamSynchronize( stream ) );
    return ACN_OK;
}

extern "C" int acn_render_main_pass_dev( acn_scene_handle* h, size_t first, size_t count, void* d_out_rgb,
                                         const acn_render_opts* opts )
{
    ACN_OPTS_VIEW
    if( !h || ( count && !d_out_rgb ) ) return fail( ACN_ERR_ARG, "null argument" );
    if( first + count > h->dev.prm.image_width * h->dev.prm.image_height ) return fail( ACN_ERR_ARG, "pixel range outside the image" );
    HIP_TRY( hipSetDevice( h->device ) );
    hipStream_t stream = ( opts && opts->stream ) ? ( hipStream_t )opts->stream : h->stream;
    int st = render_dispatch( h, nullptr, first, count, ( double* )d_out_rgb, opts, stream );
    if( st != ACN_OK ) return st;
    if( !( opts && opts->stream ) ) HIP_TRY( hipStreamSynchronize( stream ) );
    return ACN_OK;
}

extern "C" int acn_render_positions( acn_scene_handle* h, const double* pos_xy, size_t n, double* out_rgb,
                                     const acn_render_opts* opts )
{
    ACN_OPTS_VIEW
    if( !h || ( n && ( !pos_xy || !out_rgb ) ) ) return fail( ACN_ERR_ARG, "null argument" );
    if( n == 0 ) return ACN_OK;
    HIP_TRY( hipSetDevice( h->device ) );
    double* d_pos = nullptr; double* d_out = nullptr;
    HIP_TRY( hipMalloc( &d_pos, sizeof( double ) * 2 * n ) );
    hipError_t e = hipMalloc( &d_out, sizeof( double ) * 3 * n );
    if( e != hipSuccess ) { hipFree( d_pos ); return fail( ACN_ERR_DEVICE, hipGetErrorString( e ) ); }
    int st = ACN_OK;
    acn_render_opts o{};
    if( opts ) o = *opts;
    o.stream = nullptr;
    if( hipMemcpy( d_pos, pos_xy, sizeof( double ) * 2 * n, hipMemcpyHostToDevice ) != hipSuccess ) st = fail( ACN_ERR_DEVICE, "H2D copy failed" );
    if( st == ACN_OK ) st = acn_render_positions_dev( h, d_pos, n, d_out, &o );
    if( st == ACN_OK && hipMemcpy( out_rgb, d_out, sizeof( double ) * 3 * n, hipMemcpyDeviceToHost ) != hipSuccess ) st = fail( ACN_ERR_DEVICE, "D2H copy failed" );
    hipFree( d_pos ); hipFree( d_out );
    return st;
}

/* ---- sharding of whole positions: tiles of ACN_SHARD_TILE, round-robin (plain arithmetic, no GPU) ---- */
extern "C" size_t acn_shard_tile_count( size_t n, uint32_t rank, uint32_t world )
{
    if( world <= 1 ) return rank == 0 ? n : 0;
    return rank < world ? lane_count( n, ( int )world, ( int )rank ) : 0;
}
extern "C" size_t acn_shard_tile_padded( size_t n, uint32_t world )
{
    if( world <= 1 ) return n;
    size_t tiles = ( n + ACN_SHARD_TILE - 1 ) / ACN_SHARD_TILE;
    return ( ( tiles + world - 1 ) / world ) * ACN_SHARD_TILE;
}
extern "C" size_t acn_shard_tile_index( size_t n, uint32_t rank, uint32_t world, size_t i )
{
    ( void )n;
    if( world <= 1 ) return i;
    return ( ( i / ACN_SHARD_TILE ) * world + rank ) * ACN_SHARD_TILE + ( i % ACN_SHARD_TILE );
}

__global__ void k_shard_unpack( const double* __restrict__ gathered, size_t n, uint32_t world, size_t padded, double* __restrict__ frame )
{
    size_t g = ( size_t )blockIdx.x * blockDim.x + threadIdx.x;
    if( g >= n ) return;
    size_t tile = g / ACN_SHARD_TILE;
    size_t r = tile % world, i = ( tile / world ) * ACN_SHARD_TILE + g % ACN_SHARD_TILE;
    const double* src = gathered + ( r * padded + i ) * 3;
    frame[ g * 3 ] = src[ 0 ]; frame[ g * 3 + 1 ] = src[ 1 ]; frame[ g * 3 + 2 ] = src[ 2 ];
}

extern "C" int acn_render_main_pass_shard_dev( acn_scene_handle* h, size_t first, size_t count, uint32_t rank, uint32_t world,
                                               void* d_part, const acn_render_opts* opts )
{
    ACN_OPTS_VIEW
    if( !h || ( count && !d_part ) || world == 0 || rank >= world ) return fail( ACN_ERR_ARG, "bad argument" );
    if( first + count > h->dev.prm.image_width * h->dev.prm.image_height ) return fail( ACN_ERR_ARG, "pixel range outside the image" );
    HIP_TRY( hipSetDevice( h->device ) );
    hipStream_t stream = ( opts && opts->stream ) ? ( hipStream_t )opts->stream : h->stream;
    const size_t mine = acn_shard_tile_count( count, rank, world ), padded = acn_shard_tile_padded( count, world );
    if( padded > mine ) HIP_TRY( hipMemsetAsync( ( double* )d_part + 3 * mine, 0, sizeof( double ) * 3 * ( padded - mine ), stream ) );
    if( mine )
    {
        if( h->shard_pos_cap < mine )
        {
            if( h->d_shard_pos ) hipFree( h->d_shard_pos );
            h->d_shard_pos = nullptr; h->shard_pos_cap = 0;
            HIP_TRY( hipMalloc( &h->d_shard_pos, sizeof( double ) * 2 * mine ) );
            h->shard_pos_cap = mine;
        }
        hipLaunchKernelGGL( k_lane_gather, dim3( ( unsigned )( ( mine + 255 ) / 256 ) ), dim3( 256 ), 0, stream,
                            ( const double* )nullptr, first, ( uint64_t )h->dev.prm.image_width, mine, ( int )world, ( int )rank, h->d_shard_pos );
        HIP_TRY( hipGetLastError() );
        int st = render_dispatch( h, h->d_shard_pos, 0, mine, ( double* )d_part, opts, stream );
        if( st != ACN_OK ) return st;
    }
    if( !( opts && opts->stream ) ) HIP_TRY( hipStreamSynchronize( stream ) );
    return ACN_OK;
}

extern "C" int acn_shard_unpack_dev( acn_scene_handle* h, const void* d_gathered, size_t count, uint32_t world, void* d_frame,
                                     const acn_render_opts* opts )
{
    ACN_OPTS_VIEW
    if( !h || ( count && ( !d_gathered || !d_frame ) ) || world == 0 ) return fail( ACN_ERR_ARG, "bad argument" );
    if( count == 0 ) return ACN_OK;
    HIP_TRY( hipSetDevice( h->device ) );
    hipStream_t stream = ( opts && opts->stream ) ? ( hipStream_t )opts->stream : h->stream;
    hipLaunchKernelGGL( k_shard_unpack, dim3( ( unsigned )( ( count + 255 ) / 256 ) ), dim3( 256 ), 0, stream,
                        ( const double* )d_gathered, count, world, acn_shard_tile_padded( count, world ), ( double* )d_frame );
    HIP_TRY( hipGetLastError() );
    if( !( opts && opts->stream ) ) HIP_TRY( hipStreamSynchronize( stream ) );
    return ACN_OK;
}

extern "C" int acn_resolve_dev( acn_scene_handle* h, const void* d_linear_rgb, size_t n, void* d_out_rgb, void* d_out_rgb8,
                                const acn_render_opts* opts )
{
    ACN_OPTS_VIEW
    if( !h || ( n && !d_linear_rgb ) ) return fail( ACN_ERR_ARG, "null argument" );
    if( n == 0 ) return ACN_OK;
    HIP_TRY( hipSetDevice( h->device ) );
    hipStream_t stream = ( opts && opts->stream ) ? ( hipStream_t )opts->stream : h->stream;
    hipLaunchKernelGGL( k_resolve, dim3( ( unsigned )( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, stream,
                        ( const double* )d_linear_rgb, n, h->dev.prm.gamma, ( double* )d_out_rgb, ( unsigned char* )d_out_rgb8 );
    HIP_TRY( hipGetLastError() );
    if( !( opts && opts->stream ) ) HIP_TRY( hipStreamSynchronize( stream ) );
    return ACN_OK;
}

extern "C" int acn_last_kernel_ms( acn_scene_handle* h, double* trace_ms )
{
    if( !h || !trace_ms || !h->timed ) return fail( ACN_ERR_ARG, "no timed launch" );
    HIP_TRY( hipSetDevice( h->device ) );
    HIP_TRY( hipEventSynchronize( h->ev1 ) );
    float ms = 0;
    HIP_TRY( hipEventElapsedTime( &ms, h->ev0, h->ev1 ) );
    *trace_ms = ms;
    return ACN_OK;
}

extern "C" int acn_last_stage_ms( acn_scene_handle* h, double* out, int n )
{
    if( !h || !out || n < 0 || n > 25 || !h->timed ) return fail( ACN_ERR_ARG, "no timed launch" );
    HIP_TRY( hipSetDevice( h->device ) );
    HIP_TRY( hipEventSynchronize( h->ev1 ) );
    double ms[ 4 ] = { 0, 0, 0, 0 };
    size_t queue_cap = h->ws.cap[ WQ_HARD_SHADOW ], ws_bytes = h->ws.bytes, ws_allocs = h->ws.allocs;
    std::vector< const acn_scene_handle* > src{ h };
    if( h->used_lanes ) { src.assign( h->lanes.begin(), h->lanes.begin() + h->lanes_used ); queue_cap = 0; ws_bytes = 0; ws_allocs = 0; }   /* stage times: summed over the concurrent lanes of the call */
    for( const acn_scene_handle* l : src )
    {
        if( h->used_lanes ) { queue_cap += l->ws.cap[ WQ_HARD_SHADOW ]; ws_bytes += l->ws.bytes; ws_allocs += l->ws.allocs; }
        for( size_t i = 0; i < l->events_used; i++ )
        {
            float t = 0;
            HIP_TRY( hipEventElapsedTime( &t, l->events[ i ].a, l->events[ i ].b ) );
            ms[ l->events[ i ].stage ] += t;
        }
    }
    float total = 0;
    HIP_TRY( hipEventElapsedTime( &total, h->ev0, h->ev1 ) );
    double v[ 25 ] = { ms[ 0 ], ms[ 1 ], ms[ 2 ], total, ( double )h->launches[ 0 ], ( double )h->launches[ 1 ], ( double )h->launches[ 2 ],
                       ( double )h->chunks, ( double )h->retries, ( double )h->levels, ( double )h->peak_tasks, ( double )h->peak_children,
                       ( double )queue_cap, ms[ 3 ], ( double )h->launches[ 3 ], ( double )h->hard_rays,
                       ( double )h->walk_rays, ( double )h->shade_hit_recs, ( double )h->host_syncs, ( double )h->walk_steps,
                       ( double )h->flags_seen, ( double )h->private_rays, ( double )h->probe_rays, ( double )ws_bytes, ( double )ws_allocs };
    for( int k = 0; k < n && k < 25; k++ ) out[ k ] = v[ k ];
    return ACN_OK;
}

extern "C" int acn_last_counters( acn_scene_handle* h, uint64_t* out, int n )
{
    if( !h || !out || n < 0 || n > ACN_CNT_SLOTS ) return fail( ACN_ERR_ARG, "bad argument" );
    HIP_TRY( hipSetDevice( h->device ) );
    unsigned long long c[ ACN_CNT_SLOTS ], sum[ ACN_CNT_SLOTS ];
    for( int k = 0; k < ACN_CNT_SLOTS; k++ ) sum[ k ] = 0;
    std::vector< const acn_scene_handle* > src{ h };
    if( h->used_lanes ) src.assign( h->lanes.begin(), h->lanes.begin() + h->lanes_used );
    for( const acn_scene_handle* l : src )
    {
        HIP_TRY( hipMemcpy( c, l->d_counters, sizeof( c ), hipMemcpyDeviceToHost ) );
        for( int k = 0; k < ACN_CNT_SLOTS; k++ ) sum[ k ] += c[ k ];
    }
    for( int k = 0; k < n; k++ ) out[ k ] = k < ACN_CNT_SLOTS ? sum[ k ] : 0;
    return ACN_OK;
}

extern "C" int acn_estimate_envelope( acn_scene_handle* h, int32_t node, uint64_t samples, uint32_t rseed,
                                      double radius_factor, double* out )
{
    if( !h || !out || node < 0 || ( uint32_t )node >= h->dev.n_nodes ) return fail( ACN_ERR_ARG, "bad argument" );
    HIP_TRY( hipSetDevice( h->device ) );
    V3* d_scratch = nullptr; double* d_out = nullptr;
    HIP_TRY( hipMalloc( &d_scratch, sizeof( V3 ) * ( samples ? samples : 1 ) ) );
    HIP_TRY( hipMalloc( &d_out, sizeof( double ) * 4 ) );
    DevScene est_scene = h->dev;
    est_scene.lds_stack = ACN_NO_LDS_STACK;   /* one lane, no dynamic LDS: the machine keeps its stacks in scratch */
    hipLaunchKernelGGL( k_estimate_envelope, dim3( 1 ), dim3( 1 ), 0, h->stream, est_scene, node, samples, rseed, radius_factor, d_scratch, d_out );
    hipError_t e = hipGetLastError();
    if( e == hipSuccess ) e = hipStreamSynchronize( h->stream );
    if( e == hipSuccess ) e = hipMemcpy( out, d_out, sizeof( double ) * 4, hipMemcpyDeviceToHost );
    hipFree( d_scratch ); hipFree( d_out );
    if( e != hipSuccess ) return fail( ACN_ERR_DEVICE, hipGetErrorString( e ) );
    return ACN_OK;
}

extern "C" int acn_detmath_eval( int device, int op, const double* x, const double* y, double* out, size_t n )
{
    if( !x || !out ) return fail( ACN_ERR_ARG, "null argument" );
    if( acn_device_count() <= 0 ) return fail( ACN_ERR_DEVICE, "no HIP device" );
    HIP_TRY( hipSetDevice( device ) );
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIP_TRY( hipMalloc( &dx, sizeof( double ) * n ) );
    HIP_TRY( hipMalloc( &dout, sizeof( double ) * n ) );
    HIP_TRY( hipMemcpy( dx, x, sizeof( double ) * n, hipMemcpyHostToDevice ) );
    if( y )
    {
        HIP_TRY( hipMalloc( &dy, sizeof( double ) * n ) );
        HIP_TRY( hipMemcpy( dy, y, sizeof( double ) * n, hipMemcpyHostToDevice ) );
    }
    hipLaunchKernelGGL( k_detmath, dim3( ( unsigned )( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, 0, op, dx, dy, dout, n );
    HIP_TRY( hipGetLastError() );
    HIP_TRY( hipDeviceSynchronize() );
    HIP_TRY( hipMemcpy( out, dout, sizeof( double ) * n, hipMemcpyDeviceToHost ) );
    hipFree( dx ); hipFree( dout ); if( dy ) hipFree( dy );
    return ACN_OK;
}
