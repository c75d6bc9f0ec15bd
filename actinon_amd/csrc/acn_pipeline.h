/* acn_pipeline.h -- the wavefront formulation of scene_s_lum (src/scene.c:420-667) for gfx950.
 *
 * The reference evaluates one pixel by a branching recursion: specular chains (Fresnel reflection / chromatic
 * reflection / refraction, depth-1 each) with, at every diffuse shading point, a direct-light loop over
 * direct_samples*I cap samples per light and -- while depth > 10 -- a path loop over path_samples*I hemisphere
 * samples whose hits recurse with depth-10.  Every term is linear in what the recursion returns, so each pending
 * piece of work carries a colour throughput T and adds T * value into its pixel.  That turns the recursion into
 * records in HBM-resident queues and a fixed chain of kernels per path level, none of which needs the host:
 *
 *   k_walk        persistent waves.  A wave owns a private LIFO stack of pending specular rays (global memory) and
 *                 works in steps of 64 rays: whatever the stack holds, topped up with fresh input (camera rays of the
 *                 chunk's sample positions, or the ray queue k_shade_hits filled) fetched through one atomic cursor.
 *                 Each ray is traced (scene_s_trans_hit) and its hit shaded: light / background terms go to the
 *                 pixel, Fresnel / chromatic / refraction children back onto the wave's own stack, the diffuse block
 *                 becomes a DTask in a size-class queue.  No wave ever waits for another one: the whole specular
 *                 walk of a level -- up to trace_depth generations -- is ONE launch with full waves throughout.
 *   k_shade       one WAVEFRONT per DTask (16 / 4 / 1 lanes per task for the small size classes), lanes = samples:
 *                 lane j jumps the shading point's LCG stream ahead by 2j draws (the reference consumes exactly two
 *                 draws per sample, scene.c:558,598), casts its cap sample at the light, Oren-Nayar weight, shadow
 *                 ray; then the path samples: hemisphere sample, transition hit against matter; misses take the
 *                 background, hits become HitRecs of the next level; rays that enter the envelope of a CSG / SDF /
 *                 compound root element are deferred to
 *   k_hard_shadow / k_hard_path   persistent waves, one lane per deferred ray, full traversal with the CSG machine.
 *   k_shade_hits  first step of levels >= 1: shade_hit on the stored path-sample hits; fills the ray queue of k_walk.
 *
 * Every kernel reads its input count from device memory (the counter block of its level) and fetches work through
 * atomic cursors, so the host enqueues the whole chain of a chunk blind and synchronises once at its end.
 *
 * Queue appends: a wave reserves ACN_QCHUNK consecutive slots with one atomic and hands them out by ballot / prefix;
 * the unused tail of an abandoned reservation is marked dead (pixel = ACN_INVALID) and skipped by the consumer.  The
 * reservation state lives in LDS per wave, so appends may be made under divergent control flow.
 *
 * Scene access: root-compound loops run in lock-step over all lanes, so node records are fetched with wave-uniform
 * indices (scalar cache -> SGPRs); only rays that enter a CSG envelope go through the per-lane hit machine.
 *
 * Pixel accumulation is order-independent and therefore bit-reproducible: contributions are added as 2^-40
 * fixed-point integers with 64-bit integer atomics (resolution 9.1e-13; one contribution is clamped to +-16384 --
 * cl_s_sat saturates at 1.0 -- and ACN_FLAG_CLAMPED reports when that happened; 2^9 clamped adds cannot wrap).
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
#define ACN_FIX_CLAMP 16384.0

DEV long long to_fixed( double x )
{
    x = x < ACN_FIX_CLAMP ? x : ACN_FIX_CLAMP;
    x = x > -ACN_FIX_CLAMP ? x : -ACN_FIX_CLAMP;   /* NaN falls through both and converts to 0 below */
    if( x != x ) return 0;
    return __double2ll_rn( x * ACN_FIX_SCALE );
}

/* `flags`: the device word for ACN_FLAG_* bits; a clamped contribution is reported, not an error */
DEV void pixel_add( unsigned long long* accum, uint32_t* flags, size_t i, V3 c )
{
    long long x = to_fixed( c.x ), y = to_fixed( c.y ), z = to_fixed( c.z );
    if( f_abs( c.x ) > ACN_FIX_CLAMP || f_abs( c.y ) > ACN_FIX_CLAMP || f_abs( c.z ) > ACN_FIX_CLAMP ) atomicOr( flags, ACN_FLAG_CLAMPED );
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
    uint32_t pixel;     /* ACN_INVALID: dead slot */
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
    uint32_t pixel;      /* ACN_INVALID: dead slot */
};

/* a shadow ray of k_shade that entered the envelope of a CSG / SDF / compound element: finished by k_hard_shadow */
struct HardShadow
{
    V3 pos, d;
    double limit;        /* distance of the light hit */
    V3 contrib;          /* what the sample adds to the pixel if it is not occluded */
    uint32_t pixel, pad; /* pixel == ACN_INVALID: dead slot.  pad bit 0: a probe of a specular ray (see probe_push): the light root counts too */
};

/* a path ray of k_shade that did: k_hard_path finishes the transition hit */
struct HardPath
{
    V3 pos, d;
    V3 T;
    double intensity;
    int depth;
    uint32_t pixel;      /* ACN_INVALID: dead slot */
};

#define ACN_INVALID 0xFFFFFFFFu
#define ACN_INVALID_SLOT 0xFFFFFFFFu
#define ACN_NCLASS 4
/* One counter block per path level (uint32 each), all blocks of a chunk zeroed by one memset before its first launch.
 * Queue counters are high-water marks of reserved slots (dead slots included); QS_* are exact statistics.
 * QC_GEN + g: rays waiting for walk pass g of the level (g = 0: filled by k_shade_hits; g > 0: by pass g - 1);
 * QC_CUR_GEN + g: the work-fetch cursor of pass g. */
#define ACN_MAX_WALK_PASSES 32
enum
{
    QC_TASKS = 0, QC_CLASS0 = 1, QC_CHILDREN = 5, QC_FLAGS = 6, QC_HARD_SHADOW = 7, QC_HARD_PATH = 8,
    QC_CUR_HS = 9, QC_CUR_HP = 10, QC_CUR_HITS = 11, QC_CUR_SHADE0 = 20,   /* .. QC_CUR_SHADE0 + 3: one per size class */
    QC_CUR_SHADEP0 = 24,   /* .. + 3: the same for the path half of a fissioned k_shade (ACN_SHADE_PATH), which walks the same task lists */
    /* reserved slots no record was written to (the unused ends of the waves' reservations): mark - dead = records, exactly.
     * Tasks, path-sample hits, deferred path rays, specular rays (all generations of the level together); the deferred-shadow
     * queue has QS_HARD_SHADOW + QS_PROBES.  The dead slots of a chunk do not scale with it -- ~64 per wave, queue and launch --
     * so the marks of a SMALL chunk overstate its demand several times (learn_rates) */
    QS_DEAD_T = 28, QS_DEAD_C = 29, QS_DEAD_HP = 30, QS_DEAD_R = 31,
    QS_WALK_RAYS = 12, QS_HARD_SHADOW = 13, QS_HARD_PATH = 14, QS_CHILDREN = 15, QS_TASKS = 16, QS_WALK_STEPS = 17, QS_PRIVATE_RAYS = 18,
    QS_PROBES = 19,   /* specular rays that were answered by an any-hit probe instead of a walk (probe_push) */
    QC_GEN = 32, QC_CUR_GEN = QC_GEN + ACN_MAX_WALK_PASSES + 1,
    QC_N = 104
};

/* Size classes of the shading tasks: 64 / 16 / 4 / 1 lanes per task, by the larger of the task's two sample counts.
 * Narrow groups lose less in the last, partly filled round of a sample loop (200 samples on 64 lanes: 4 rounds, 78 %
 * of the lanes busy; on 16 lanes: 13 rounds, 96 %), and the tasks that share a wave are neighbours with similar counts;
 * a whole wavefront per task keeps the rays of a round on one origin.  Where the 64-lane class begins is chosen per
 * scene at upload (DevScene.class0_min, see there; ACN_CLASS0_MIN overrides). */
#ifndef ACN_CLASS1_MIN
#define ACN_CLASS1_MIN 8
#endif
DEV int size_class( uint64_t n, uint32_t class0_min )
{
    return n > class0_min ? 0 : n > ACN_CLASS1_MIN ? 1 : n > 2 ? 2 : 3;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* Queue appends.  A wave reserves ACN_QCHUNK slots of a queue with ONE returning atomic and hands them out itself
 * (ballot + prefix): the ~1-2 us round trip of a device-scope atomic is paid once per 64 records instead of once per
 * append.  When a request does not fit the rest of the reservation, the rest is marked dead and a new one is taken;
 * consumers skip dead slots.  The state ( next free slot, end ) of a wave's reservation lives in LDS, one pair per
 * wave and queue, and is touched by one lane per append -- which makes appends legal under divergent control flow
 * (the sample loops of k_shade): the lanes that reach an append together form its ballot. */
#ifndef ACN_QCHUNK
#define ACN_QCHUNK 64
#endif
#ifndef ACN_QCHUNK_MAX
#define ACN_QCHUNK_MAX 512u
#endif
/* spare: a second reservation taken ahead of need (chunk_prefetch), ACN_INVALID if none */
struct ChunkState { uint32_t cur, end, spare, size; };   /* size: slots per reservation (ACN_QCHUNK unless the kernel knows better, see chunks_resize) */
typedef ChunkState ACN_LDS* ChunkP;
#define ACN_NCHUNKS 8     /* reservation states per wave */
/* LDS block of the reservation states of a 256-lane workgroup; 256 B keeps the dynamic LDS behind it 16-byte aligned */
#define ACN_CHUNK_STATES __shared__ __attribute__( ( aligned( 16 ) ) ) ChunkState acn_chunk_states[ 4 * ACN_NCHUNKS ];
/* (the wave's index through readfirstlane: the pointer is then a scalar.  As a per-lane value derived from threadIdx.x it was
 * spilled and re-loaded from scratch four times per append: ~80 of the ~110 scratch loads of a k_walk step, profiles/r04) */
#define ACN_CHUNKS_OF_WAVE ( ( ChunkP )acn_chunk_states + __builtin_amdgcn_readfirstlane( ( int )( threadIdx.x >> 6 ) ) * ACN_NCHUNKS )
/* how many lanes of `mask` lie below the calling lane: two VALU instructions, no lane index or lane mask kept in registers */
DEV uint32_t lanes_below( unsigned long long mask )
{
    return __builtin_amdgcn_mbcnt_hi( ( uint32_t )( mask >> 32 ), __builtin_amdgcn_mbcnt_lo( ( uint32_t )mask, 0u ) );
}

DEV void chunks_init( ChunkP cs )
{
    if( ( threadIdx.x & 63 ) < ACN_NCHUNKS ) { cs[ threadIdx.x & 63 ].cur = 0; cs[ threadIdx.x & 63 ].end = 0; cs[ threadIdx.x & 63 ].spare = ACN_INVALID_SLOT; cs[ threadIdx.x & 63 ].size = ACN_QCHUNK; }
}
/* Larger reservations for a wave that is going to append a lot.  Every reservation is one returning atomic on the queue's ONE
 * counter, and the device serves same-address atomics of a launch one after the other: k_shade_hits on many_spheres turns 2.7e8
 * path hits per frame (every 16th pixel) into as many tasks, one reservation per wave and batch of 64 -- and spent 152 ms doing it
 * with 4 % of its VALU slots busy; with 256 slots per reservation 64 ms (profiles/r04/ab_qchunk_s24.txt).  The price of a large
 * reservation is its unused tail (dead slots the consumers step over: a fixed 256 costs the wine glass 4 - 18 %), so the size
 * follows what the wave expects to append: an eighth of its share of the input, 64 ... 512. */
DEV void chunks_resize( ChunkP cs, uint32_t items_of_launch )
{
    const uint32_t per_wave = items_of_launch / ( gridDim.x * 4u );
    uint32_t size = ( per_wave / 8u ) & ~63u;
    size = size < ( uint32_t )ACN_QCHUNK ? ( uint32_t )ACN_QCHUNK : size > ACN_QCHUNK_MAX ? ACN_QCHUNK_MAX : size;
    if( ( threadIdx.x & 63 ) < ACN_NCHUNKS ) cs[ threadIdx.x & 63 ].size = size;
}

/* Reservations ahead of need.  A k_walk wave appends up to 64 records per step to each of its queues, so nearly every
 * step runs a reservation dry and pays the round trip of a returning atomic (1-2 us) per queue in the middle of its
 * shading.  At the start of a step (all lanes active, nothing else in flight) lane q looks at queue q of the wave: if its
 * reservation exists and would not survive one more full step, the lane starts the atomic for the next one; the result
 * is parked in LDS (ChunkState.spare) after the step's traversal, when it has long arrived, and chunk_alloc takes it
 * from there.  One VGPR for all queues; a queue the wave never appended to is never reserved ahead. */
struct ChunkPrefetch { uint32_t base; bool issued; };
DEV void chunk_prefetch_issue( ChunkP cs, uint32_t* counter, bool enabled, ChunkPrefetch& pf )   /* lane q: cs = state of queue q */
{
    pf.base = 0;
    const uint32_t cur = cs->cur, end = cs->end, spare = cs->spare;
#ifndef ACN_RESERVE_AHEAD   /* off by default: measured three times against the same kernels without it, it costs 0.5 - 0.7 ms of the
                              71 ms wine_glass frame (the extra LDS reads and the compare of every step) and gains nothing where
                              the steps are long (profiles/r03/NOTES.md) */
    enabled = false;
#endif
    pf.issued = enabled && end != 0u && end - cur < 64u && spare == ACN_INVALID_SLOT;
    if( pf.issued ) pf.base = atomicAdd( counter, cs->size );
}
DEV void chunk_prefetch_park( ChunkP cs, const ChunkPrefetch& pf )
{
    if( pf.issued ) cs->spare = pf.base;
}

/* every lane with `want` gets a distinct slot of the queue counted by *counter.  A request that does not fit the rest
 * of the wave's reservation uses that rest up and continues in a new one, so slots only die at the end of a kernel. */
DEV uint32_t chunk_alloc( ChunkP cs, uint32_t* counter, bool want )
{
    unsigned long long mask = __ballot( want );
    if( !want ) return ACN_INVALID;
    /* from here on the active lanes are those with `want`: the first of them (rank 0) leads, and readfirstlane reads it */
    const uint32_t rank = lanes_below( mask );
    uint32_t m = ( uint32_t )__popcll( mask );
    uint32_t base = 0, room = 0, base2 = 0;
    if( rank == 0 )
    {
        uint32_t cur = cs->cur, end = cs->end;
        base = cur; room = end - cur;
        if( m > room )
        {
            const uint32_t spare = cs->spare, size = cs->size;
            if( spare != ACN_INVALID_SLOT ) { base2 = spare; cs->spare = ACN_INVALID_SLOT; }
            else base2 = atomicAdd( counter, size );
            cs->end = base2 + size;
            cs->cur = base2 + ( m - room );
        }
        else cs->cur = cur + m;
    }
    base  = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )base );
    room  = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )room );
    base2 = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )base2 );
    return rank < room ? base + rank : base2 + ( rank - room );
}

/* end of a kernel (all lanes of the wave): the unused tail of the wave's last reservation is dead */
template< class KILL >
DEV void chunk_close( ChunkP cs, uint32_t cap, KILL kill, uint32_t* dead = nullptr )
{
    uint32_t cur = cs->cur, end = cs->end, spare = cs->spare, size = cs->size;
    for( uint32_t k = cur + ( threadIdx.x & 63 ); k < end; k += 64 ) if( k < cap ) kill( k );
    if( spare != ACN_INVALID_SLOT ) for( uint32_t k = spare + ( threadIdx.x & 63 ); k < spare + size; k += 64 ) if( k < cap ) kill( k );
    const uint32_t d = end - cur + ( spare != ACN_INVALID_SLOT ? size : 0u );
    if( dead && d && ( threadIdx.x & 63 ) == 0 ) atomicAdd( dead, d );
}

/* one atomic per wave: every lane with `want` gets a distinct slot (unreserved form, used where appends are rare) */
DEV uint32_t wave_alloc( uint32_t* counter, bool want )
{
    unsigned long long mask = __ballot( want );
    if( !want ) return ACN_INVALID;
    const uint32_t rank = lanes_below( mask );
    uint32_t base = 0;
    if( rank == 0 ) base = atomicAdd( counter, ( uint32_t )__popcll( mask ) );
    base = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )base );
    return base + rank;
}

DEV void wave_stat_add( uint32_t* counter, uint32_t v )   /* all lanes of the wave */
{
    for( int o = 32; o > 0; o >>= 1 ) v += __shfl_down( v, o, 64 );
    if( ( threadIdx.x & 63 ) == 0 && v ) atomicAdd( counter, v );
}

/* work fetch of the persistent kernels: the wave takes the next `want` items of [ 0, n ); returns how many it got */
DEV uint32_t wave_fetch( uint32_t* cursor, uint32_t want, uint32_t n, uint32_t* first )
{
    uint32_t b = 0;
    if( ( threadIdx.x & 63 ) == 0 ) b = atomicAdd( cursor, want );
    b = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )b );
    *first = b;
    if( b >= n ) return 0;
    return n - b < want ? n - b : want;
}

/* Batch size of a kernel's work fetch for an input of n items: the configured batch, but small enough that every wave of
 * the grid comes back for work about eight times -- with one batch per wave the kernel ends when the wave that drew the
 * expensive items is done (1/8 of a 1080p frame: k_shade 5.4 ms with batches of 16 steps, where the work is 1.2 ms).
 * unit: items of one step of a wave. */
DEV uint32_t balanced_batch( uint32_t n, uint32_t batch, uint32_t unit )
{
    const uint32_t waves = gridDim.x * ( blockDim.x >> 6 );
    uint32_t b = n / ( waves * 8u );
    b -= b % unit;
    b = b < unit ? unit : b;
    return b < batch ? b : batch;
}

/* the same in batches, one ahead: the wave owns the range [ cur, end ) of the input and, while it works that off, the
 * atomic that reserves its next batch is already in flight -- its 1-2 us round trip overlaps the work instead of
 * stalling the wave once per batch.  ( cur, end, more are the same in every lane; next is lane 0's. ) */
struct FetchRange { uint32_t cur, end, next; bool pending, more; };
DEV void range_init( FetchRange& r, bool any ) { r.cur = r.end = r.next = 0; r.pending = false; r.more = any; }
DEV uint32_t range_take( FetchRange& r, uint32_t* cursor, uint32_t batch, uint32_t n, uint32_t want, uint32_t* first )
{
    if( r.cur == r.end && r.more )
    {
        if( !r.pending && ( threadIdx.x & 63 ) == 0 ) r.next = atomicAdd( cursor, batch );
        uint32_t b = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )r.next );
        r.pending = false;
        if( b >= n ) { r.more = false; r.cur = r.end = 0; }
        else
        {
            r.cur = b;
            r.end = n - b < batch ? n : b + batch;
            if( b + batch >= n ) r.more = false;
            else
            {
                if( ( threadIdx.x & 63 ) == 0 ) r.next = atomicAdd( cursor, batch );   /* the batch after this one */
                r.pending = true;
            }
        }
    }
    uint32_t have = r.end - r.cur;
    uint32_t take = have < want ? have : want;
    *first = r.cur;
    r.cur += take;
    return take;
}

/* ---- where shade_hit puts what it produces ---- */

/* the queues of diffuse shading tasks: tasks[] + one index list per size class */
struct TaskQ
{
    DTask*    tasks;
    uint32_t* idx[ ACN_NCLASS ];
    uint32_t* counts;               /* counter block of the level */
    uint32_t  task_cap;
    HardShadow* probes;             /* the level's hard-shadow queue: where probe_push appends */
    uint32_t  probe_cap;
    uint32_t  emit_terms;           /* 0: pixel terms under no sharded sample loop are dropped (ACN_SHARD_SAMPLES, level 0, rank > 0) */
    uint32_t* flags;                /* the chunk's ACN_FLAG_* word */
};

/* a ray queue in global memory (k_shade_hits -> k_walk; generation g -> generation g + 1 of k_walk) */
struct RayQ
{
    RayTask*  rays;
    uint32_t* counter;
    uint32_t  cap;
    uint32_t* flags;
    ChunkP    cs;

    DEV void push( bool want, V3 p, V3 d, V3 T, double intensity, int depth, uint32_t pixel ) const
    {
        uint32_t slot = chunk_alloc( cs, counter, want );
        if( want )
        {
            if( slot < cap )
            {
                RayTask& c = rays[ slot ];
                c.p = p; c.d = d; c.T = T; c.intensity = intensity; c.depth = depth; c.pixel = pixel;
            }
            else atomicOr( flags, ACN_FLAG_CHILD_OVERFLOW );
        }
    }
    DEV void close( uint32_t* dead ) const
    {
        RayTask* r = rays;
        chunk_close( cs, cap, [ r ]( uint32_t k ) { r[ k ].pixel = ACN_INVALID; }, dead );
    }
};

/* Where a k_walk wave puts the specular children of its rays.  `priv` is the same in every wave of a launch:
 *   false  generation pass: children go to the global queue of the next generation (the next launch spreads them
 *          over the whole chip again);
 *   true   the input is small: children go onto the wave's private LIFO stack and are traced by the same wave in its
 *          next steps; what does not fit the stack joins the next generation's queue.
 * `top` is the same in all lanes (every lane of the wave calls push). */
struct WalkSink
{
    bool      priv;
    RayTask*  stack;
    uint32_t  cap;
    uint32_t  top;
    RayQ      out;

    DEV void push( bool want, V3 p, V3 d, V3 T, double intensity, int depth, uint32_t pixel )
    {
        if( !priv ) { out.push( want, p, d, T, intensity, depth, pixel ); return; }
        unsigned long long mask = __ballot( want );
        uint32_t slot = top + lanes_below( mask );
        uint32_t new_top = top + ( uint32_t )__popcll( mask );
        bool fits = slot < cap;
        if( want && fits )
        {
            RayTask& c = stack[ slot ];
            c.p = p; c.d = d; c.T = T; c.intensity = intensity; c.depth = depth; c.pixel = pixel;
        }
        top = new_top < cap ? new_top : cap;
        if( new_top > cap ) out.push( want && !fits, p, d, T, intensity, depth, pixel );   /* wave-uniform branch */
    }
};

/* Rays that only need a yes / no.  scene_s_lum returns zero at once when depth == 0 or intensity < trace_min_intensity
 * (scene.c:430), so for a child ray that is born that way -- the Fresnel reflection off a polished floor seen through one
 * diffuse bounce, say: 4 % of an intensity of 0.4 is below the 0.03 of the shipped scripts -- the reference's
 *     if( scene_s_trans_hit( ... ) < f3_inf ) lum_l = scene_s_lum( ... ) [ = 0 ];  else lum_l = background * intensity;
 * (scene.c:484-491, 507-514, 644-651) asks only WHETHER the ray hits anything.  Such rays do not join the ray queue of
 * k_walk (closest hit, normal, media transition, the CSG machine with normals); they become records of the level's
 * hard-shadow queue: an any-hit test over both roots with the cheap elements first, early exit, no normals -- and the
 * background term added if nothing is hit (k_hard_shadow).  Same value, same fixed-point sum.
 * limit: hits at or beyond it do not count; F3_BIG = the largest finite double stands for "any finite hit". */
#define F3_BIG 1.7976931348623157e308
DEV void probe_push( const TaskQ& tq, ChunkP pcs, bool want, V3 p, V3 d, double limit, V3 contrib, uint32_t pixel, uint32_t flags )
{
    uint32_t slot = chunk_alloc( pcs, &tq.counts[ QC_HARD_SHADOW ], want );
    if( want )
    {
        /* statistics: one add per wave and call */
        const unsigned long long m = __ballot( 1 );
        if( lanes_below( m ) == 0 ) atomicAdd( &tq.counts[ QS_PROBES ], ( uint32_t )__popcll( m ) );
        if( slot < tq.probe_cap )
        {
            HardShadow& h = tq.probes[ slot ];
            h.pos = p; h.d = d; h.limit = limit; h.contrib = contrib; h.pixel = pixel; h.pad = flags;
        }
        else atomicOr( tq.flags, ACN_FLAG_CHILD_OVERFLOW );
    }
}
/* a specular child of shade_hit: onto the ray queue if scene_s_lum would look at its hit, else a probe */
template< class RAYS, class CT >
DEV void spawn_child( const DevScene& sc, RAYS& rays, const TaskQ& tq, ChunkP tcs, bool f, V3 p, V3 d, V3 T, double intensity, int depth, uint32_t pixel, CT* cnt )
{
    const bool dead = f && ( depth == 0 || intensity < sc.prm.trace_min_intensity );
    rays.push( f && !dead, p, d, T, intensity, depth, pixel );
    if( dead ) cnt->add( CNT_TRANS_RAY, 2u );   /* the two compound_s_ray_trans_hit calls of scene_s_trans_hit this ray stands for */
    probe_push( tq, tcs + 6, dead && tq.emit_terms, p, d, F3_BIG, v_mld( T, v_mlf( ld3( sc.prm.background_color ), intensity ) ), pixel, 1u );
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* scene_s_lum for one hit, everything except the two sample loops (scene.c:420-537, 623-664).  Specular children
 * (Fresnel reflection, chromatic reflection, refraction) are pushed to `rays`; the diffuse block becomes a DTask.
 * Every lane of the wave must call this (the appends are wave-wide); lanes with nothing to shade pass depth 0. */
template< class RAYS, class CT >
DEV void shade_hit( const DevScene& sc, RAYS& rays, const TaskQ& tq, ChunkP tcs, V3 rp, V3 rd, double offs, const Trans& trans, int depth,
                    double intensity, V3 T, uint32_t pixel, V3& acc, CT* cnt )
{
    const double min_intensity = sc.prm.trace_min_intensity;
    bool go = !( depth == 0 || intensity < min_intensity );
    if( go ) { cnt->inc( CNT_LUM ); cnt->cost( ACN_F_LUM_FIXED ); }
    V3 pos = ray_pos( rp, rd, offs );
    MatP enter_obj = ( go && trans.enter_obj >= 0 ) ? &sc.mats[ trans.enter_obj ] : nullptr;
    MatP exit_obj  = ( go && trans.exit_obj  >= 0 ) ? &sc.mats[ trans.exit_obj  ] : nullptr;

    if( go && enter_obj && enter_obj->radiance > 0 )   /* :432-437 */
    {
        double diff_sqr = v_diff_sqr( pos, ld3( sc.nodes[ trans.enter_obj ].pos ) );
        double light_intensity = ( diff_sqr > 0 ) ? ( enter_obj->radiance / diff_sqr ) : F3_MAG;
        V3 c = v_mlf( obj_color_dev( sc, trans.enter_obj, pos ), light_intensity * intensity );
        acc.x += T.x * c.x; acc.y += T.y * c.y; acc.z += T.z * c.z;
        cnt->cost( ACN_F_EMISSION );
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
            cnt->cost( ACN_F_ABSORB, ACN_T_ABSORB );
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
        if( f ) { cnt->cost( ACN_F_FRESNEL_REFL ); reflectance = fresnel_reflection( rd, trans.exit_nor, trix, &out_d ) * fresnel_reflectivity; }
        spawn_child( sc, rays, tq, tcs, f, pos, out_d, T, reflectance * intensity, depth - 1, pixel, cnt );
        if( f ) intensity *= ( 1.0 - reflectance );
    }

    /* chromatic reflection :498-523 */
    {
        bool f = go && chromatic_reflectivity > 0 && intensity >= min_intensity;
        V3 out_d = rd;
        if( f ) { cnt->cost( ACN_F_REFLECTION ); out_d = v_reflection( rd, trans.exit_nor ); }
        spawn_child( sc, rays, tq, tcs, f, pos, out_d, v_mld( T, enter_color ), chromatic_reflectivity * intensity, depth - 1, pixel, cnt );
        if( f ) intensity *= ( 1.0 - chromatic_reflectivity );
    }

    /* diffuse reflection :526-537: hand the sample loops to k_shade */
    bool diffuse = go && intensity * diffuse_reflectivity >= min_intensity;
    double diffuse_intensity = intensity * diffuse_reflectivity;
    uint64_t n_direct = 0, n_path = 0;
    if( diffuse )
    {
        cnt->cost( ACN_F_SHADE_DIFFUSE + ACN_F_SEED, ACN_T_SHADE_DIFFUSE + ACN_T_SEED );
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
        /* task slots need no dead marks: tasks are only reached through the index lists */
        uint32_t slot = chunk_alloc( tcs, &tq.counts[ QC_TASKS ], emit );
        bool ok = emit && slot < tq.task_cap;
        if( emit && !ok ) atomicOr( tq.flags, ACN_FLAG_TASK_OVERFLOW );
        int cls = ok ? size_class( n_direct > n_path ? n_direct : n_path, sc.class0_min ) : -1;
        if( ok )
        {
            DTask& t = tq.tasks[ slot ];
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
            uint32_t* list = tq.idx[ k ];
            uint32_t is = chunk_alloc( tcs + 1 + k, &tq.counts[ QC_CLASS0 + k ], cls == k );
            if( cls == k )
            {
                if( is < tq.task_cap ) list[ is ] = slot;
                else atomicOr( tq.flags, ACN_FLAG_TASK_OVERFLOW );
            }
        }
    }
    if( diffuse ) intensity *= ( 1.0 - diffuse_reflectivity );

    /* refraction :633-653 */
    {
        bool f = go && transparent && intensity >= min_intensity;
        V3 out_d = rd;
        if( f ) { cnt->cost( ACN_F_FRESNEL_REFR ); out_d = fresnel_refraction( rd, trans.exit_nor, trix ); }
        spawn_child( sc, rays, tq, tcs, f, ray_pos( rp, rd, offs + 2.0 * F3_EPS ), out_d, T, intensity, depth - 1, pixel, cnt );
    }
}

/* the index lists' reservations of a wave end with the kernel, and that of its probes */
DEV void task_chunks_close( const TaskQ& tq, ChunkP tcs )
{
    /* (task slots need no dead marks -- tasks are reached through the index lists -- but the end of the wave's reservation counts) */
    chunk_close( tcs, 0u, []( uint32_t ) {}, &tq.counts[ QS_DEAD_T ] );
    {
        HardShadow* pr = tq.probes;
        chunk_close( tcs + 6, tq.probe_cap, [ pr ]( uint32_t j ) { pr[ j ].pixel = ACN_INVALID; } );
    }
    for( int k = 0; k < ACN_NCLASS; k++ )
    {
        uint32_t* list = tq.idx[ k ];
        chunk_close( tcs + 1 + k, tq.task_cap, [ list ]( uint32_t j ) { list[ j ] = ACN_INVALID; } );
    }
}

DEV void wave_add_counters( unsigned long long* global, const Cnt< true >& mine )
{
    for( int k = 0; k < CNT_N + 2; k++ )
    {
        unsigned long long v = k < CNT_N ? mine.c[ k ] : k == CNT_N ? mine.flop : mine.transc;   /* [ CNT_N ] flop, [ CNT_N + 1 ] transcendentals */
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
/* Kernel parameter convention: every scene buffer is passed as its own __restrict__ pointer and the DevScene view is
 * rebuilt inside (scene arrays are then read through the constant address space, see acn_device.h); p_counts is the
 * counter block of the kernel's level. */
#define ACN_SCENE_PARAMS  DevScene sc_in, const GNode* __restrict__ p_nodes, const GMat* __restrict__ p_mats, const int32_t* __restrict__ p_elems, const acn_texture* __restrict__ p_textures
#define ACN_SCENE_VIEW    DevScene sc = sc_in; sc.nodes = ( NodeP )p_nodes; sc.gnodes = ( NodeP )p_nodes; sc.mats = ( MatP )p_mats; sc.elems = ( ElemP )p_elems; sc.textures = ( TexP )p_textures; sc.flags = sc_in.flags; sc.lds_stack = ACN_NO_LDS_STACK;

/* One ACN_FLAG_* word per chunk (DevScene.flags: the word of the chunk's first counter block), whatever the path level of
 * the kernel that raises a flag.  A queue that overflowed means the host will redo the chunk smaller (launch_render):
 * every later kernel of the chain, enqueued blind, looks at the word first and leaves at once -- the rest of a lost
 * chunk costs launches, not work (scenes with thousands of hits per position redo 15 - 25 % of their chunks). */
/* (the word is loaded next to the kernel's input count, one wait for both: a load of its own at the top of every kernel cost
 * 0.5 ms of the 71 ms frame, ~200 launches of which each is on the chunk's critical path) */
#define ACN_CHUNK_FLAGS_LOAD const uint32_t chunk_flags_ = *( const uint32_t* )sc_in.flags;
#define ACN_LEAVE_IF_CHUNK_IS_LOST \
    if( __builtin_amdgcn_readfirstlane( ( int )chunk_flags_ ) & ( int )( ACN_FLAG_TASK_OVERFLOW | ACN_FLAG_CHILD_OVERFLOW ) ) return;
/* The same for the kernels whose waves depend on each other afterwards (ACN_STAGE_NODES: every wave copies its quarter of the
 * node array into LDS, then the block synchronises).  Other workgroups of the SAME launch set the word concurrently, so the waves
 * of one workgroup may read different values; a wave that left alone would leave its quarter of the staged nodes unwritten and
 * the others would traverse garbage (a spurious ACN_FLAG_STACK_OVERFLOW turned a recoverable lost chunk into a failed call:
 * ADVICE r03).  Wave 0's reading decides for the whole workgroup. */
#define ACN_LEAVE_IF_CHUNK_IS_LOST_BLOCK \
    { \
        __shared__ uint32_t chunk_lost_; \
        if( threadIdx.x == 0 ) chunk_lost_ = chunk_flags_ & ( ACN_FLAG_TASK_OVERFLOW | ACN_FLAG_CHILD_OVERFLOW ); \
        __syncthreads(); \
        if( chunk_lost_ ) return; \
    }

/* LDS staging of the node array (kernels whose node reads are per-lane: the CSG machines).  The block copies the
 * GNode array into dynamic shared memory once; per-lane node reads then are ds_read instead of global loads. */
#define ACN_STAGE_NODES( sc ) \
    { \
        const double* src_ = ( const double* )p_nodes; \
        uint32_t words_ = ( sc ).n_nodes * ( uint32_t )( sizeof( GNode ) / sizeof( double ) ); \
        for( uint32_t k_ = threadIdx.x; k_ < words_; k_ += blockDim.x ) acn_lds_raw[ k_ ] = src_[ k_ ]; \
        __syncthreads(); \
    }

#define ACN_TASKQ_PARAMS DTask* __restrict__ p_tasks, uint32_t* __restrict__ p_idx0, uint32_t* __restrict__ p_idx1, \
    uint32_t* __restrict__ p_idx2, uint32_t* __restrict__ p_idx3, uint32_t* __restrict__ p_counts, uint32_t task_cap, \
    HardShadow* __restrict__ p_probes, uint32_t probe_cap, uint32_t probe_emit
#define ACN_TASKQ_VIEW \
    TaskQ tq; \
    tq.tasks = p_tasks; tq.idx[ 0 ] = p_idx0; tq.idx[ 1 ] = p_idx1; tq.idx[ 2 ] = p_idx2; tq.idx[ 3 ] = p_idx3; \
    tq.counts = p_counts; tq.task_cap = task_cap; tq.probes = p_probes; tq.probe_cap = probe_cap; tq.emit_terms = probe_emit; tq.flags = sc_in.flags;

#ifndef ACN_TRACE_WAVES
#define ACN_TRACE_WAVES ACN_WALK_WAVES
#endif
/* ACN_POOLED=1: the machine kernels pool the rays of a workgroup's four waves per root element (acn_device.h: pooled_machine_hit).
 * Measured slower on every workload, off (defined in acn_device.h) */
#ifndef ACN_HPATH_WAVES
#define ACN_HPATH_WAVES ACN_WALK_WAVES
#endif
#ifndef ACN_WALK_MAX_STEPS
#define ACN_WALK_MAX_STEPS ( 1u << 20 )   /* safety bound of a wave's step loop (64 M rays per wave) */
#endif

/* The order in which the sample positions of a call are worked off.  A chunk is a contiguous range of SLOTS; slot s
 * stands for position  tile( s / 256 ) * 256 + s % 256  with  tile( t ) = t * mul mod n_tiles  (mul coprime to n_tiles:
 * a bijection).  Tiles of 256 consecutive positions keep neighbouring pixels in one workgroup; the multiplicative stride
 * spreads the tiles of any chunk over the whole frame, so that every chunk of a call sees the same mix of sky, floor
 * and glass -- the queue fill of one chunk then predicts the next one's (launch_render sizes chunks that way). */
#define ACN_ORDER_SHIFT 8   /* tiles of 256 positions: the four waves of a workgroup work on neighbouring pixels and visit the same nodes
                               (tiles of 64 mix a small chunk better, but hanging_lamp 2160p p1024 -- 550 KB of nodes -- ran 25 % slower
                               and the small scenes 1 - 3 %) */
struct TileOrder
{
    uint32_t n;         /* positions of the call */
    uint32_t n_tiles;   /* ceil( n / 256 ) */
    uint32_t mul;
    uint32_t sample_stride;   /* > 0: the learning pass of a cold handle (learn_rates): slot s stands for position s * sample_stride */
    DEV uint32_t position( uint32_t slot ) const
    {
        if( sample_stride ) { const uint64_t p = ( uint64_t )slot * sample_stride; return p < n ? ( uint32_t )p : 0xFFFFFFFFu; }
        uint32_t tile = ( uint32_t )( ( ( uint64_t )( slot >> ACN_ORDER_SHIFT ) * mul ) % n_tiles );
        return ( tile << ACN_ORDER_SHIFT ) + ( slot & ( ( 1u << ACN_ORDER_SHIFT ) - 1u ) );
    }
};

/* One pass of the specular walk of a path level: persistent waves, work fetched through the atomic cursor of the pass.
 * Input (generation `pass` of the level): rays_in == nullptr: the camera rays of the sample positions in slots
 * [ base, base + n_cam ) of the call's TileOrder (lum_machine_s_func, scene.c:976-1011); else the ray queue rays_in[ 0 .. min( p_counts[ QC_GEN + pass ], in_cap ) ).
 * Each ray is traced (scene_s_trans_hit) and its hit shaded: light / background terms go to the pixel, the diffuse block
 * becomes a DTask, and the Fresnel / chromatic / refraction children go
 *   - input larger than private_limit: to rays_out, the queue of generation pass + 1 -- the next launch deals them to
 *     the whole chip again (no wave follows a long chain alone while the others idle);
 *   - else: onto the wave's private LIFO stack (stacks: ( waves of the grid ) x stack_stride slots, stack_cap of them
 *     used); the wave works in steps of 64 rays -- the top of its stack, topped up with fresh input -- until both are
 *     empty, so the tail of a level, up to trace_depth generations of a few rays each, costs no further launch.  Rays
 *     that do not fit the stack join rays_out.
 * The host enqueues a fixed number of passes per level blind; passes whose input is empty exit at once.
 * emit_terms: 0 drops the pixel terms of this launch (ACN_SHARD_SAMPLES, level 0, rank > 0). */
template< bool COUNT, bool LDS, bool PRUNE >
__global__ __launch_bounds__( 256, ACN_TRACE_WAVES )
void k_walk( ACN_SCENE_PARAMS, ACN_TASKQ_PARAMS, const RayTask* __restrict__ rays_in, uint32_t in_cap, uint32_t pass,
             const double* __restrict__ pos_xy, size_t first_pixel, uint32_t base, uint32_t n_cam, TileOrder order,
             RayTask* __restrict__ rays_out, uint32_t out_cap, uint32_t private_limit,
             RayTask* __restrict__ stacks, uint32_t stack_stride, uint32_t stack_cap, uint32_t fetch_batch, uint32_t emit_terms,
             unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_CHUNK_STATES
    ACN_CHUNK_FLAGS_LOAD
    uint32_t n_in = n_cam;
    if( rays_in ) { n_in = p_counts[ QC_GEN + pass ]; n_in = n_in < in_cap ? n_in : in_cap; }
    n_in = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )n_in );
    ACN_LEAVE_IF_CHUNK_IS_LOST_BLOCK
    if( n_in == 0 ) return;
    ACN_SCENE_VIEW
    ACN_TASKQ_VIEW
    if( sc_in.lds_stack != ACN_NO_LDS_STACK ) sc.lds_stack = LDS ? sc.n_nodes * ( uint32_t )sizeof( GNode ) : 0u;   /* the CSG stacks follow the staged nodes */
    if constexpr( LDS ) ACN_STAGE_NODES( sc )
    const ChunkP cs = ACN_CHUNKS_OF_WAVE;
    chunks_init( cs );
    ACN_PHASE_INIT
    Cnt< COUNT > cnt;
    cnt.clear();
    const int lane = ( int )( threadIdx.x & 63 );
    const uint32_t wave = blockIdx.x * ( blockDim.x >> 6 ) + ( threadIdx.x >> 6 );
    WalkSink sink;
    sink.priv = n_in <= private_limit;
    sink.stack = stacks + ( size_t )wave * stack_stride; sink.cap = stack_cap; sink.top = 0;
    sink.out.rays = rays_out; sink.out.counter = p_counts + QC_GEN + pass + 1; sink.out.cap = out_cap; sink.out.flags = sc_in.flags; sink.out.cs = cs + 5;
    uint32_t* cursor = p_counts + QC_CUR_GEN + pass;
    FetchRange fr;
    range_init( fr, true );
    uint32_t traced = 0, steps = 0;
    bool finished = false;
    RayPool pool = ray_pool_of( sc );
    for( uint32_t step = 0; step < ACN_WALK_MAX_STEPS; step++ )
    {
        /* the step's 64 rays: the top of the private stack, topped up with fresh input */
        uint32_t n_pop = sink.top < 64u ? sink.top : 64u;
        uint32_t n_fresh = 0, fb = 0;
        if( n_pop < 64u ) n_fresh = range_take( fr, cursor, fetch_batch, n_in, 64u - n_pop, &fb );
#if ACN_POOLED
        /* the waves of a workgroup step together (they pool their rays per root element): a wave that is out of work keeps
         * stepping, with no rays, until all four are */
        const bool have_rays = n_pop + n_fresh != 0;
        if( !__syncthreads_or( have_rays ? 1 : 0 ) ) { finished = true; break; }
#else
        const bool have_rays = true;
        if( n_pop + n_fresh == 0 ) { finished = true; break; }
#endif
        /* reservations this step may run dry: their atomics travel while the step's rays are traced (lane q: queue q of the
         * wave -- 0 tasks, 1 .. 4 the class lists, 5 the next generation's rays, 6 probes) */
        ChunkPrefetch pf;
        {
            const int q = lane < 7 ? lane : 0;
            uint32_t* ctr = q == 0 ? &tq.counts[ QC_TASKS ] : q <= ACN_NCLASS ? &tq.counts[ QC_CLASS0 + q - 1 ] : q == 5 ? sink.out.counter : &tq.counts[ QC_HARD_SHADOW ];
            chunk_prefetch_issue( cs + q, ctr, lane < 7 && !( q == 5 && sink.priv ), pf );
        }
        sink.top -= n_pop;
        const RayTask* src = nullptr;
        bool live = false;
        uint32_t pixel = 0;
        V3 rp = mk( 0, 0, 0 ), rd = mk( 0, 0, 1 );
        if( ( uint32_t )lane < n_pop ) { src = sink.stack + sink.top + lane; live = true; }
        else if( ( uint32_t )lane < n_pop + n_fresh )
        {
            uint32_t i = fb + ( ( uint32_t )lane - n_pop );
            if( rays_in ) { src = rays_in + i; live = src->pixel != ACN_INVALID; }
            else
            {
                /* the camera ray of a sample position (slots past the call's last position are empty) */
                pixel = order.position( base + i );
                live = pixel < order.n;
                if( !live ) pixel = 0;
                double mx, my;
                if( pos_xy ) { mx = pos_xy[ ( size_t )pixel * 2 ]; my = pos_xy[ ( size_t )pixel * 2 + 1 ]; }
                else
                {
                    size_t pix = first_pixel + pixel;
                    mx = ( double )( pix % sc.prm.image_width ) + 0.5;
                    my = ( double )( pix / sc.prm.image_width ) + 0.5;
                }
                camera_ray( sc, mx, my, &rp, &rd );
                if( live ) cnt.cost( ACN_F_CAMERA_RAY );
            }
        }
        if( src && live ) { rp = src->p; rd = src->d; }
        if( live ) traced++;
        if( have_rays ) steps++;

        /* scene_s_trans_hit */
        Trans trans;
        trans.exit_nor = mk( 0, 0, 0 ); trans.exit_obj = -1; trans.enter_obj = -1;
        double offs = F3_INF;
#if ACN_POOLED
        if constexpr( LDS ) offs = scene_trans_hit_pooled( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, ( LdsNodeP )acn_lds_raw ), pool, live, rp, rd, &trans, &cnt );
        else                offs = scene_trans_hit_pooled( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, sc.nodes ), pool, live, rp, rd, &trans, &cnt );
#else
        if( live )
        {
            if constexpr( LDS ) offs = scene_trans_hit_dev( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, ( LdsNodeP )acn_lds_raw ), rp, rd, &trans, &cnt );
            else                offs = scene_trans_hit_dev( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, sc.nodes ), rp, rd, &trans, &cnt );
        }
#endif
        chunk_prefetch_park( cs + ( lane < 7 ? lane : 0 ), pf );
        /* what the ray carries is read only now, so that it does not occupy registers across the traversal */
        asm volatile( "" ::: "memory" );
        V3 T = mk( 1, 1, 1 );
        double intensity = 1.0;
        int depth = ( int )sc.prm.trace_depth;
        if( src && live ) { T = src->T; intensity = src->intensity; depth = src->depth; pixel = src->pixel; }
#ifdef ACN_WALK_REFETCH
        /* ... and origin and direction are read (camera rays: computed) a second time for the shading: twelve registers less
         * across the traversal, which copies them anyway */
        if( src ) { if( live ) { rp = src->p; rd = src->d; } }
        else
        {
            asm volatile( "" : "+v"( pixel ) );
            double mx, my;
            if( pos_xy ) { mx = pos_xy[ ( size_t )pixel * 2 ]; my = pos_xy[ ( size_t )pixel * 2 + 1 ]; }
            else
            {
                size_t pix = first_pixel + pixel;
                mx = ( double )( pix % sc.prm.image_width ) + 0.5;
                my = ( double )( pix / sc.prm.image_width ) + 0.5;
            }
            camera_ray( sc, mx, my, &rp, &rd );
        }
#endif
        bool hit = live && offs < F3_INF;
        V3 acc = mk( 0, 0, 0 );
        if( live && !hit ) acc = v_mld( T, v_mlf( ld3( sc.prm.background_color ), intensity ) );
        shade_hit( sc, sink, tq, cs, rp, rd, hit ? offs : 0.0, trans, hit ? depth : 0, intensity, T, pixel, acc, &cnt );
        /* emit_terms == 0: level 0 of a rank > 0 of a sample-sharded call -- emission / background reached through
         * specular chains are under no sharded loop and come from rank 0 alone */
        if( live && emit_terms ) pixel_add( accum, sc.flags, pixel, acc );
        /* the wave reads next what it wrote last: same wave, program order; the fence keeps the compiler from moving the
         * next step's loads above this step's stores */
        __builtin_amdgcn_fence( __ATOMIC_SEQ_CST, "wavefront" );
        ACN_LAP( PH_SHADE );
    }
    ACN_LAP( PH_TAIL );
    ACN_PHASE_FLUSH( counters, 0 )
    /* the step bound is a safety net against a loop that does not end; work would be lost, so the call fails */
    if( !finished && lane == 0 ) atomicOr( sc_in.flags, ACN_FLAG_STACK_OVERFLOW );
    sink.out.close( p_counts + QS_DEAD_R );
    task_chunks_close( tq, cs );
    wave_stat_add( p_counts + QS_WALK_RAYS, traced );
    if( sink.priv ) wave_stat_add( p_counts + QS_PRIVATE_RAYS, traced );
    if( lane == 0 && steps ) atomicAdd( p_counts + QS_WALK_STEPS, steps );
    wave_add_counters( counters, cnt );
}

/* first step of a level >= 1: one lane per path-sample hit of the previous level (the recursive scene_s_lum call of
 * scene.c:610); persistent waves fetch 64 records at a time.  n_ptr: the previous level's QC_CHILDREN. */
template< bool COUNT >
__global__ __launch_bounds__( 256, ACN_WALK_WAVES )
void k_shade_hits( ACN_SCENE_PARAMS, ACN_TASKQ_PARAMS, const HitRec* __restrict__ recs, const uint32_t* __restrict__ n_ptr, uint32_t rec_cap, uint32_t fetch_batch,
                   RayTask* __restrict__ rays_out, uint32_t ray_cap,
                   unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_CHUNK_STATES
    ACN_CHUNK_FLAGS_LOAD
    uint32_t n = *n_ptr;
    n = n < rec_cap ? n : rec_cap;
    n = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )n );
    ACN_LEAVE_IF_CHUNK_IS_LOST
    if( n == 0 ) return;
    fetch_batch = balanced_batch( n, fetch_batch, 64u );
    ACN_SCENE_VIEW
    ACN_TASKQ_VIEW
    const ChunkP cs = ACN_CHUNKS_OF_WAVE;
    chunks_init( cs );
    chunks_resize( cs, n );
    Cnt< COUNT > cnt;
    cnt.clear();
    RayQ rq;
    rq.rays = rays_out; rq.counter = p_counts + QC_GEN; rq.cap = ray_cap; rq.flags = sc_in.flags; rq.cs = cs + 5;
    FetchRange fr;
    range_init( fr, n > 0 );
    for( ;; )
    {
        uint32_t first = 0;
        uint32_t got = range_take( fr, p_counts + QC_CUR_HITS, fetch_batch, n, 64u, &first );
        if( got == 0 ) break;
        uint32_t i = first + ( threadIdx.x & 63 );
        bool live = ( threadIdx.x & 63 ) < got;
        HitRec r;
        r.p = mk( 0, 0, 0 ); r.d = mk( 0, 0, 1 ); r.offs = 0; r.exit_nor = mk( 0, 0, 0 ); r.T = mk( 0, 0, 0 ); r.intensity = 0;
        r.exit_obj = -1; r.enter_obj = -1; r.depth = 0; r.pixel = ACN_INVALID;
        if( live ) r = recs[ i ];
        live = live && r.pixel != ACN_INVALID;
        V3 acc = mk( 0, 0, 0 );
        Trans trans;
        trans.exit_nor = r.exit_nor; trans.exit_obj = r.exit_obj; trans.enter_obj = r.enter_obj;
        shade_hit( sc, rq, tq, cs, r.p, r.d, r.offs, trans, live ? r.depth : 0, r.intensity, r.T, r.pixel, acc, &cnt );
        if( live ) pixel_add( accum, sc.flags, r.pixel, acc );
    }
    rq.close( p_counts + QS_DEAD_R );
    task_chunks_close( tq, cs );
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

/* Task frames.  A shading point's sample loops run 13 (200 samples on 16 lanes) to hundreds of rounds over ~40 doubles that do
 * not change from round to round: the surface frame, the Oren-Nayar constants, the light's sampling frame, position, radiance and
 * colour, the throughput.  Held in registers next to a round's own temporaries they do not fit the 128 VGPRs k_shade runs best
 * with, and what the allocator spills is exactly these values: round 3's kernel reloaded ~16 of them from SCRATCH in every round
 * (1.95e8 vector-memory instructions per 1080p frame at ~590 cycles each, 17 GB of spill writes; profiles/r03/NOTES.md section 1).
 * All LPT lanes of a task hold the SAME values, so they live once per task in LDS instead: the lane with sub == 0 writes the
 * frame when the task (and each light) is set up, and a round reads a value where it needs it -- a broadcast ds_read, ~64 cycles,
 * no VMEM slot, no register held across the round.  The accesses are volatile: the compiler neither keeps a value in a register
 * across uses nor moves a read above the write of another lane (LDS operations of one wave execute in program order).
 * Narrow classes (LPT < 16: 64 or 256 tasks per workgroup) keep the frame in registers as before: they are 3 % of the time. */
enum
{
    TF_SURFACE_D = 0, TF_RAY_PRJ = 3, TF_SIN_I = 6, TF_COS_I, TF_TAN_I, TF_THETA_I, TF_ON_A, TF_ON_B, TF_DIFF_I,
    TF_CON = 13,          /* 9: rows of the transposed sampling frame (src_con of the light at hand, then out_con of the path loop) */
    TF_CYL_HGT = 22, TF_LIGHT_POS = 23, TF_RADIANCE = 26,
    TF_SCALE = 27,        /* 3: direct loop: Tc * light_color * ( 2 cyl_hgt / direct_samples ), what a deferred sample's c is multiplied with;
                                path loop: Tchild = Tc * ( 2 / path_samples ) */
    TF_N = 30
};
template< bool IN_LDS > struct TaskFrame;
template<> struct TaskFrame< true >
{
    volatile double ACN_LDS* p;
    DEV double get( int k ) const { return p[ k ]; }
    DEV V3 get3( int k ) const { return mk( p[ k ], p[ k + 1 ], p[ k + 2 ] ); }
    DEV void set( bool writer, int k, double v ) { if( writer ) p[ k ] = v; }
    DEV void set3( bool writer, int k, V3 v ) { if( writer ) { p[ k ] = v.x; p[ k + 1 ] = v.y; p[ k + 2 ] = v.z; } }
    /* the writes of the task's first lane are visible to the reads of the others: same wave, program order; the fence keeps the
     * compiler from moving memory operations across */
    DEV void publish() const { __builtin_amdgcn_fence( __ATOMIC_SEQ_CST, "wavefront" ); __builtin_amdgcn_wave_barrier(); }
};
template<> struct TaskFrame< false >
{
    double r[ TF_N ];
    DEV double get( int k ) const { return r[ k ]; }
    DEV V3 get3( int k ) const { return mk( r[ k ], r[ k + 1 ], r[ k + 2 ] ); }
    DEV void set( bool, int k, double v ) { r[ k ] = v; }
    DEV void set3( bool, int k, V3 v ) { r[ k ] = v.x; r[ k + 1 ] = v.y; r[ k + 2 ] = v.z; }
    DEV void publish() const {}
};
template< class F > DEV M3 frame_con( const F& f )
{
    M3 m;
    m.x = f.get3( TF_CON ); m.y = f.get3( TF_CON + 3 ); m.z = f.get3( TF_CON + 6 );
    return m;
}
template< class F > DEV void frame_set_con( F& f, bool w, const M3& m ) { f.set3( w, TF_CON, m.x ); f.set3( w, TF_CON + 3, m.y ); f.set3( w, TF_CON + 6, m.z ); }

/* LEAF_LIGHTS: every light is a plane / sphere / squaroid-free leaf, so the kernel contains no call into the CSG
 * machine at all (the usual case); otherwise the light hit goes through the generic element test.
 * The tasks are idx[ 0 .. min( p_counts[ QC_CLASS0 + cls ], task_cap ) ) (dead entries skipped); the persistent waves of
 * the grid fetch them through the cursor of the class.
 * PART: 0 both sample loops of a task (one pixel add per task); 1 the direct-light loops only; 2 the path loop only (its LCG
 * stream starts behind the 2 * direct_samples draws per light of the loops it does not run): the two halves of a fissioned
 * launch, which share nothing but the task record (see render_chunk). */
#define ACN_SHADE_BOTH   0
#define ACN_SHADE_DIRECT 1
#define ACN_SHADE_PATH   2
template< int LPT, bool COUNT, bool LEAF_LIGHTS, bool PRUNE, int PART = ACN_SHADE_BOTH >
__global__ __launch_bounds__( 256, ACN_SHADE_WAVES )
void k_shade( ACN_SCENE_PARAMS, const DTask* __restrict__ tasks, const uint32_t* __restrict__ idx, int cls, uint32_t task_cap, uint32_t fetch_batch,
              HitRec* __restrict__ p_children, uint32_t child_cap, HardShadow* __restrict__ p_hard_shadow,
              HardPath* __restrict__ p_hard_path, uint32_t hs_cap, uint32_t hard_cap, uint32_t* __restrict__ p_counts, uint32_t shard_rank, uint32_t shard_world,
              unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_CHUNK_STATES
    ACN_CHUNK_FLAGS_LOAD
    ACN_SCENE_VIEW
    const auto scp = scene_view< PRUNE >( sc, sc.nodes );   /* the scene as the two root-traversal fast paths see it */
    const ChunkP cs = ACN_CHUNKS_OF_WAVE;                   /* [0] hard shadow, [1] hard path, [2] children */
    chunks_init( cs );
    ACN_PHASE_INIT
    constexpr int G = 64 / LPT;
    constexpr bool FRAME_IN_LDS = LPT >= 16;
    __shared__ __attribute__( ( aligned( 16 ) ) ) double acn_task_frames[ FRAME_IN_LDS ? ( 256 / LPT ) * TF_N : 1 ];
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPT;
    const int grp = lane / LPT;
    const bool writer = sub == 0;
    TaskFrame< FRAME_IN_LDS > fr_;
    if constexpr( FRAME_IN_LDS ) fr_.p = ( volatile double ACN_LDS* )acn_task_frames + ( threadIdx.x / LPT ) * TF_N;
    const TaskFrame< FRAME_IN_LDS >& F = fr_;
    Cnt< COUNT > cnt;
    cnt.clear();
    uint32_t n_tasks = p_counts[ QC_CLASS0 + cls ];
    n_tasks = n_tasks < task_cap ? n_tasks : task_cap;
    n_tasks = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )n_tasks );
    ACN_LEAVE_IF_CHUNK_IS_LOST
    fetch_batch = balanced_batch( n_tasks, fetch_batch, ( uint32_t )( 64 / LPT ) );
    uint32_t n_hs = 0, n_hp = 0, n_ch = 0;   /* statistics: records written by this lane */
    auto kill_hs = [ p_hard_shadow ]( uint32_t k ) { p_hard_shadow[ k ].pixel = ACN_INVALID; };
    auto kill_hp = [ p_hard_path ]( uint32_t k ) { p_hard_path[ k ].pixel = ACN_INVALID; };
    auto kill_ch = [ p_children ]( uint32_t k ) { p_children[ k ].pixel = ACN_INVALID; };

    /* persistent waves: G tasks per step, fetched fetch_batch at a time through the class's cursor */
    FetchRange fr;
    range_init( fr, n_tasks > 0 );
    for( ;; )
    {
        uint32_t base = 0;
        uint32_t got = range_take( fr, p_counts + ( PART == ACN_SHADE_PATH ? QC_CUR_SHADEP0 : QC_CUR_SHADE0 ) + cls, fetch_batch, n_tasks, ( uint32_t )G, &base );
        if( got == 0 ) break;
        uint32_t ti = base + grp;
        if( ( uint32_t )grp >= got ) continue;
        uint32_t slot = ( ( ElemP )( const void* )idx )[ ti ];
        if( LPT == 64 ) slot = __builtin_amdgcn_readfirstlane( slot );
        if( slot == ACN_INVALID ) continue;
        const DTask ACN_CONST& t = ( ( const DTask ACN_CONST* )tasks )[ slot ];
        const V3 pos = ldc( t.pos );   /* the origin of every ray of the task: stays in registers */
        const double diffuse_intensity = t.diffuse_intensity;
        const bool oren_nayar = t.on_b > 0;
        {
            const V3 surface_d = ldc( t.surface_d );
            fr_.set3( writer, TF_SURFACE_D, surface_d );
            fr_.set3( writer, TF_RAY_PRJ, ldc( t.ray_projection ) );
            const double theta_i = t.theta_i;
            double sin_i = 0, cos_i = 1;   /* loop invariant half of the Oren-Nayar term */
            if( oren_nayar ) acn_sincos( theta_i, &sin_i, &cos_i );
            fr_.set( writer, TF_SIN_I, sin_i ); fr_.set( writer, TF_COS_I, cos_i );
            fr_.set( writer, TF_TAN_I, cos_i > 0 ? sin_i / cos_i : 0.0 );   /* direct-light loop only, and only where cos_i > weight > 0 */
            fr_.set( writer, TF_THETA_I, theta_i );
            fr_.set( writer, TF_ON_A, t.on_a ); fr_.set( writer, TF_ON_B, t.on_b );
            fr_.set( writer, TF_DIFF_I, diffuse_intensity );
        }
        uint64_t rv = t.rv;
        V3 lum = mk( 0, 0, 0 );   /* lum_l of scene.c:539, identical in all lanes of the group after each reduction */
        ACN_LAP( PH_FETCH );      /* diagnostic build: k_shade books 10 task fetch / set-up, 4 cap sample, 5 light hit, 6 Oren-Nayar,
                                     7 occlusion, 8 queue appends and sums, 9 path sample, 11 path transition hit */
        const uint64_t direct_samples = [ & ]() { uint64_t n = ( uint64_t )( sc.prm.direct_samples * diffuse_intensity ); return n == 0 ? ( uint64_t )1 : n; }();
        NodeP light = &sc.nodes[ sc.light_root ];
        const int n_lights = light->child1;

        /* ---- direct light, scene.c:542-581 ---- */
        if constexpr( PART != ACN_SHADE_PATH )
        for( int li = 0; li < n_lights; li++ )
        {
            int light_idx = __builtin_amdgcn_readfirstlane( sc.elems[ light->child0 + li ] );
            NodeP light_src = &sc.nodes[ light_idx ];
            MatP light_mat = &sc.mats[ light_idx ];
            if( sub == 0 ) cnt.cost( ACN_F_FOV + ACN_F_FRAME );   /* per task and light: booked by one lane of the group */
            uint64_t skip;
            V3 light_color;
            double norm;
            {
                V3 fov_d; double cos_rs;
                obj_fov_dev( light_src, pos, &fov_d, &cos_rs );
                const M3 src_frame = m_con_z( fov_d );
                const double cyl_hgt = 1 - cos_rs;
                /* root elements no shadow ray of this loop can reach (all of them lie in the light's sampling cone; src_frame.z is
                 * the axis the cap samples are drawn around) */
                skip = root_cone_cull( scp, sc.matter_root, pos, src_frame.z, 1.0 - cyl_hgt );
                const V3 light_pos = ld3( light_src->pos );
                light_color = obj_color_dev( sc, light_idx, light_pos );   /* scene.c:552 */
                norm = 2.0 * cyl_hgt / direct_samples;
                frame_set_con( fr_, writer, m_transposed( src_frame ) );
                fr_.set( writer, TF_CYL_HGT, cyl_hgt );
                fr_.set3( writer, TF_LIGHT_POS, light_pos );
                fr_.set( writer, TF_RADIANCE, light_mat->radiance );
                const V3 Tc = ldc( t.Tc );
                fr_.set3( writer, TF_SCALE, mk( Tc.x * ( light_color.x * norm ), Tc.y * ( light_color.y * norm ), Tc.z * ( light_color.z * norm ) ) );
                fr_.publish();
            }

            double s = 0;
            /* ACN_SHARD_SAMPLES (shard_world > 1, level 0 only): this rank's share of the loop; sample j keeps its
             * place in the LCG stream and the normalisation above keeps the whole loop's n */
            uint64_t j_lo = 0, j_hi = direct_samples;
            uint64_t rv0 = rv;
            if( shard_world > 1 )
            {
                j_lo = direct_samples * shard_rank / shard_world; j_hi = direct_samples * ( shard_rank + 1 ) / shard_world;
                rv0 = lcg00_jump( rv, 2 * j_lo );
            }
            uint64_t rvj = lcg_jump_lane( rv0, sub );
            for( uint64_t j = j_lo + sub; j < j_hi; j += LPT )
            {
                ACN_TALLY( 0, true );   /* diagnostic build: lanes in a round of the direct-light loop */
                uint64_t r = rvj;
                rvj = lcg_stride< LPT >( rvj );
                cnt.inc( CNT_CAP_SAMPLE );
                cnt.cost( ACN_F_CAP_SAMPLE, ACN_T_CAP_SAMPLE );
                const V3 cap = v_random_sphere_cap( &r, F.get( TF_CYL_HGT ) );
                const V3 out_d = m_mlv( frame_con( F ), cap );
                double weight = v_mlv( out_d, F.get3( TF_SURFACE_D ) );
                ACN_LAP( PH_M_LEAF );
                if( weight <= 0 ) continue;
                double a;
                if( LEAF_LIGHTS ) a = leaf_element_hit< false >( light_src, light_src->type, pos, out_d, nullptr, &cnt );
                else a = light_hit_call( sc, light_idx, pos, out_d, &cnt );
                ACN_LAP( PH_M_PAIR );
                if( a >= F3_INF ) continue;
                cnt.inc( CNT_SHADOW_RAY );
                ACN_LAP( PH_M_FRAME );
                int occ = root_occluded_fast( scp, sc.matter_root, pos, out_d, a, skip, &cnt );
                ACN_LAP( PH_M_SIDE );
                if( occ == 1 ) continue;
                /* what the sample adds if it is not occluded (scene.c:569-574); the weight of a direct-light sample only scales its
                 * colour: closed form (oren_nayar_weight_direct).  Computed behind the occlusion test: an occluded sample needs none
                 * of it, and the test above holds no register for it */
                if( oren_nayar )
                {
                    cnt.cost( ACN_F_OREN_NAYAR_DIRECT );
                    weight = oren_nayar_weight_direct( weight, F.get( TF_SIN_I ), F.get( TF_COS_I ), F.get( TF_TAN_I ), F.get( TF_ON_A ), F.get( TF_ON_B ), out_d,
                                                       F.get3( TF_SURFACE_D ), F.get3( TF_RAY_PRJ ) );
                }
                V3 hit_pos = ray_pos( pos, out_d, a );
                double diff_sqr = v_diff_sqr( hit_pos, F.get3( TF_LIGHT_POS ) );
                double local_intensity = ( diff_sqr > 0 ) ? ( F.get( TF_RADIANCE ) / diff_sqr ) : F3_MAG;
                double c = local_intensity * weight * F.get( TF_DIFF_I );
                if( occ == 0 ) { s += c; cnt.cost( ACN_F_DIRECT_TAIL ); }
                /* hard shadow rays: appended to the queue of k_hard_shadow, which adds the contribution itself if unoccluded */
                uint32_t hs = chunk_alloc( cs + 0, &p_counts[ QC_HARD_SHADOW ], occ == 2 );
                if( occ == 2 )
                {
                    if( hs < hs_cap )
                    {
                        HardShadow& h = p_hard_shadow[ hs ];
                        const V3 scale = F.get3( TF_SCALE );
                        h.pos = pos; h.d = out_d; h.limit = a;
                        h.contrib = mk( scale.x * c, scale.y * c, scale.z * c );
                        h.pixel = t.pixel; h.pad = 0;
                        n_hs++;
                    }
                    else
                    {
                        atomicOr( sc_in.flags, ACN_FLAG_CHILD_OVERFLOW );
                    }
                }
            }
            rv = lcg00_jump( rv, 2 * direct_samples );
            s = group_sum< LPT >( s );
            double f = s * norm;
            lum.x += light_color.x * f; lum.y += light_color.y * f; lum.z += light_color.z * f;
            ACN_LAP( PH_SHADE );
        }
        if constexpr( PART == ACN_SHADE_PATH ) rv = lcg00_jump( rv, 2 * direct_samples * ( uint64_t )n_lights );

        /* ---- path tracing, scene.c:584-621 ---- */
        if constexpr( PART != ACN_SHADE_DIRECT )
        if( sc.prm.path_samples && t.depth > 10 )
        {
            if( sub == 0 ) cnt.cost( ACN_F_FRAME );
            uint64_t path_samples = ( uint64_t )( sc.prm.path_samples * diffuse_intensity );
            path_samples = ( path_samples == 0 ) ? 1 : path_samples;
            const double norm = 2.0 / path_samples;
            const V3 bg = ld3( sc.prm.background_color );
            /* "a < max_path_length" as the any-hit limit "a <= path_limit": the largest double below it */
            const double path_limit = sc.prm.max_path_length < F3_INF ? acn_bits_f64( acn_f64_bits( sc.prm.max_path_length ) - 1ull ) : F3_BIG;
            {
                frame_set_con( fr_, writer, m_transposed( m_con_z( F.get3( TF_SURFACE_D ) ) ) );
                fr_.set3( writer, TF_SCALE, v_mlf( ldc( t.Tc ), norm ) );
                fr_.publish();
            }
            double bsum = 0;
            uint64_t j_lo = 0, j_hi = path_samples;
            uint64_t rv0 = rv;
            if( shard_world > 1 )
            {
                j_lo = path_samples * shard_rank / shard_world; j_hi = path_samples * ( shard_rank + 1 ) / shard_world;
                rv0 = lcg00_jump( rv, 2 * j_lo );
            }
            uint64_t rvj = lcg_jump_lane( rv0, sub );
            for( uint64_t j = j_lo + sub; j < j_hi; j += LPT )
            {
                ACN_TALLY( 2, true );   /* ... of the path loop */
                uint64_t r = rvj;
                rvj = lcg_stride< LPT >( rvj );
                cnt.inc( CNT_CAP_SAMPLE );
                cnt.cost( ACN_F_CAP_SAMPLE, ACN_T_CAP_SAMPLE );
                const V3 cap = v_random_sphere_cap( &r, 1.0 );
                const V3 out_d = m_mlv( frame_con( F ), cap );
                double weight = v_mlv( out_d, F.get3( TF_SURFACE_D ) );
                bool live = weight > 0;
                double a = F3_INF;
                Trans trans;
                trans.exit_nor = mk( 0, 0, 0 ); trans.exit_obj = -1; trans.enter_obj = -1;
                bool hard = false;
                if( live )
                {
                    if( oren_nayar )
                    {
                        cnt.cost( ACN_F_OREN_NAYAR, ACN_T_OREN_NAYAR );
                        weight = oren_nayar_weight_pre( weight, F.get( TF_THETA_I ), F.get( TF_SIN_I ), F.get( TF_COS_I ), F.get( TF_ON_A ), F.get( TF_ON_B ), out_d,
                                                        F.get3( TF_SURFACE_D ), F.get3( TF_RAY_PRJ ) );
                    }
                    cnt.cost( ACN_F_PATH_TAIL );
                    ACN_LAP( PH_COMPOUND );
                    a = root_trans_hit_fast( scp, sc.matter_root, pos, out_d, &trans, &hard, &cnt );
                    ACN_LAP( PH_TAIL );
                }
                const double child_intensity = weight * diffuse_intensity;
                bool hit = live && !hard && a < sc.prm.max_path_length;
                if( live && !hard && !hit ) bsum += child_intensity;
                /* scene_s_lum of a hit with depth - 10 == 0 or too little intensity is zero (scene.c:430): such a hit is not
                 * queued, and such a ray that needs the machine only needs a yes / no (see probe_push) */
                const bool dark = t.depth - 10 == 0 || child_intensity < sc.prm.trace_min_intensity;
                if( dark ) hit = false;
                const bool probe = hard && dark;
                if( probe ) hard = false;
                uint32_t ps = chunk_alloc( cs + 0, &p_counts[ QC_HARD_SHADOW ], probe );
                if( probe )
                {
                    cnt.inc( CNT_TRANS_RAY );   /* the compound_s_ray_trans_hit this probe stands for */
                    if( ps < hs_cap )
                    {
                        HardShadow& h = p_hard_shadow[ ps ];
                        h.pos = pos; h.d = out_d; h.limit = path_limit;
                        h.contrib = v_mld( F.get3( TF_SCALE ), v_mlf( bg, child_intensity ) );   /* k_hard_path's term for a miss */
                        h.pixel = t.pixel; h.pad = 0;
                        n_hs++;
                    }
                    else
                    {
                        atomicOr( sc_in.flags, ACN_FLAG_CHILD_OVERFLOW );
                    }
                }
                /* hard path rays: the transition hit is finished by k_hard_path */
                uint32_t hp = chunk_alloc( cs + 1, &p_counts[ QC_HARD_PATH ], hard );
                if( hard )
                {
                    if( hp < hard_cap )
                    {
                        HardPath& h = p_hard_path[ hp ];
                        h.pos = pos; h.d = out_d; h.T = F.get3( TF_SCALE ); h.intensity = child_intensity;
                        h.depth = t.depth - 10; h.pixel = t.pixel;
                        n_hp++;
                    }
                    else
                    {
                        atomicOr( sc_in.flags, ACN_FLAG_CHILD_OVERFLOW );
                    }
                }
                /* the surviving path rays: the next level's queue */
                uint32_t csl = chunk_alloc( cs + 2, &p_counts[ QC_CHILDREN ], hit );
                if( hit )
                {
                    if( csl < child_cap )
                    {
                        HitRec& c = p_children[ csl ];
                        c.p = pos; c.d = out_d; c.offs = a; c.exit_nor = trans.exit_nor; c.T = F.get3( TF_SCALE );
                        c.intensity = child_intensity;
                        c.exit_obj = trans.exit_obj; c.enter_obj = trans.enter_obj;
                        c.depth = t.depth - 10; c.pixel = t.pixel;
                        n_ch++;
                    }
                    else
                    {
                        atomicOr( sc_in.flags, ACN_FLAG_CHILD_OVERFLOW );
                    }
                }
            }
            bsum = group_sum< LPT >( bsum ) * norm;
            lum.x += bg.x * bsum; lum.y += bg.y * bsum; lum.z += bg.z * bsum;
        }

        if( sub == 0 ) pixel_add( accum, sc.flags, t.pixel, v_mld( ldc( t.Tc ), lum ) );
        ACN_LAP( PH_SHADE );
    }
    chunk_close( cs + 0, hs_cap, kill_hs );
    chunk_close( cs + 1, hard_cap, kill_hp, p_counts + QS_DEAD_HP );
    chunk_close( cs + 2, child_cap, kill_ch, p_counts + QS_DEAD_C );
    wave_stat_add( p_counts + QS_HARD_SHADOW, n_hs );
    wave_stat_add( p_counts + QS_HARD_PATH, n_hp );
    wave_stat_add( p_counts + QS_CHILDREN, n_ch );
    ACN_PHASE_FLUSH( counters, 3 )
    wave_add_counters( counters, cnt );
}

/* the shadow rays k_shade could not decide inline: full occlusion test, one lane per ray, persistent waves */
template< bool COUNT, bool LDS, bool PRUNE >
__global__ __launch_bounds__( 256, ACN_WALK_WAVES )
void k_hard_shadow( ACN_SCENE_PARAMS, const HardShadow* __restrict__ recs, uint32_t cap, uint32_t fetch_batch, uint32_t* __restrict__ p_counts,
                    unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_CHUNK_FLAGS_LOAD
    uint32_t n = p_counts[ QC_HARD_SHADOW ];
    n = n < cap ? n : cap;
    n = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )n );
    ACN_LEAVE_IF_CHUNK_IS_LOST_BLOCK
    if( n == 0 ) return;
    ACN_SCENE_VIEW
    if( sc_in.lds_stack != ACN_NO_LDS_STACK ) sc.lds_stack = LDS ? sc.n_nodes * ( uint32_t )sizeof( GNode ) : 0u;   /* the CSG stacks follow the staged nodes */
    Cnt< COUNT > cnt;
    cnt.clear();
    if constexpr( LDS ) ACN_STAGE_NODES( sc )
    ACN_PHASE_INIT
    fetch_batch = balanced_batch( n, fetch_batch, 64u );
    FetchRange fr;
    range_init( fr, n > 0 );
#if ACN_POOLED
    RayPool pool = ray_pool_of( sc );
    for( ;; )
    {
        uint32_t first = 0;
        uint32_t got = range_take( fr, p_counts + QC_CUR_HS, fetch_batch, n, 64u, &first );
        if( !__syncthreads_or( got != 0 ? 1 : 0 ) ) break;   /* the four waves of the workgroup pool their rays: they leave together */
        HardShadow r;
        r.pos = mk( 0, 0, 0 ); r.d = mk( 0, 0, 1 ); r.limit = 0; r.contrib = mk( 0, 0, 0 ); r.pixel = ACN_INVALID; r.pad = 0;
        if( ( threadIdx.x & 63 ) < got ) r = recs[ first + ( threadIdx.x & 63 ) ];
        const bool live = r.pixel != ACN_INVALID;
        bool occ = false;
        ACN_LAP( PH_FETCH );
        /* a probe of a specular ray (probe_push) asks scene_s_trans_hit's question: the lights count as well.
         * One call site for both roots (the traversal with the CSG machine is in-line code). */
        #pragma unroll 1
        for( int k = 0; k < 2; k++ )
        {
            const int root = k ? sc.matter_root : sc.light_root;
            const bool want = live && !occ && ( k == 1 || ( r.pad & 1u ) );
            if( !__syncthreads_or( want ? 1 : 0 ) ) continue;
            bool o2;
            if constexpr( LDS ) o2 = root_occluded_pooled( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, ( LdsNodeP )acn_lds_raw ), pool, root, want, r.pos, r.d, r.limit, &cnt );
            else                o2 = root_occluded_pooled( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, sc.nodes ), pool, root, want, r.pos, r.d, r.limit, &cnt );
            if( want && o2 ) occ = true;
        }
        ACN_LAP( PH_ROOT_LEAF );
        if( live && !occ ) { cnt.cost( ACN_F_DIRECT_TAIL ); pixel_add( accum, sc.flags, r.pixel, r.contrib ); }
        ACN_LAP( PH_SHADE );
    }
#else
    for( ;; )
    {
        uint32_t first = 0;
        uint32_t got = range_take( fr, p_counts + QC_CUR_HS, fetch_batch, n, 64u, &first );
        if( got == 0 ) break;
        if( ( threadIdx.x & 63 ) < got )
        {
            HardShadow r = recs[ first + ( threadIdx.x & 63 ) ];
            if( r.pixel != ACN_INVALID )
            {
                bool occ = false;
                ACN_LAP( PH_FETCH );
                #pragma unroll 1
                for( int k = 0; k < 2; k++ )
                {
                    const int root = k ? sc.matter_root : sc.light_root;
                    const bool want = !occ && ( k == 1 || ( r.pad & 1u ) );
                    if( __ballot( want ) == 0ull ) continue;
                    if( want )
                    {
                        if constexpr( LDS ) occ = root_occluded( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, ( LdsNodeP )acn_lds_raw ), root, r.pos, r.d, r.limit, &cnt );
                        else                occ = root_occluded( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, sc.nodes ), root, r.pos, r.d, r.limit, &cnt );
                    }
                }
                ACN_LAP( PH_ROOT_LEAF );
                if( !occ ) { cnt.cost( ACN_F_DIRECT_TAIL ); pixel_add( accum, sc.flags, r.pixel, r.contrib ); }
                ACN_LAP( PH_SHADE );
            }
        }
    }
#endif
    ACN_LAP( PH_TAIL );
    ACN_PHASE_FLUSH( counters, 1 )
    wave_add_counters( counters, cnt );
}

/* the path rays k_shade could not finish inline: full transition hit; hits join the next level's HitRec queue */
template< bool COUNT, bool LDS, bool PRUNE >
__global__ __launch_bounds__( 256, ACN_HPATH_WAVES )
void k_hard_path( ACN_SCENE_PARAMS, const HardPath* __restrict__ recs, uint32_t cap, uint32_t fetch_batch, HitRec* __restrict__ p_children, uint32_t child_cap,
                  uint32_t* __restrict__ p_counts, unsigned long long* __restrict__ accum, unsigned long long* __restrict__ counters )
{
    ACN_CHUNK_STATES
    ACN_CHUNK_FLAGS_LOAD
    uint32_t n = p_counts[ QC_HARD_PATH ];
    n = n < cap ? n : cap;
    n = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )n );
    ACN_LEAVE_IF_CHUNK_IS_LOST_BLOCK
    if( n == 0 ) return;
    ACN_SCENE_VIEW
    if( sc_in.lds_stack != ACN_NO_LDS_STACK ) sc.lds_stack = LDS ? sc.n_nodes * ( uint32_t )sizeof( GNode ) : 0u;   /* the CSG stacks follow the staged nodes */
    Cnt< COUNT > cnt;
    cnt.clear();
    if constexpr( LDS ) ACN_STAGE_NODES( sc )
    ACN_PHASE_INIT
    const ChunkP cs = ACN_CHUNKS_OF_WAVE;
    chunks_init( cs );
    fetch_batch = balanced_batch( n, fetch_batch, 64u );
    uint32_t n_ch = 0;
    auto kill_ch = [ p_children ]( uint32_t k ) { p_children[ k ].pixel = ACN_INVALID; };
    FetchRange fr;
    range_init( fr, n > 0 );
    RayPool pool = ray_pool_of( sc );
    for( ;; )
    {
        uint32_t first = 0;
        uint32_t got = range_take( fr, p_counts + QC_CUR_HP, fetch_batch, n, 64u, &first );
#if ACN_POOLED
        if( !__syncthreads_or( got != 0 ? 1 : 0 ) ) break;   /* the four waves of the workgroup pool their rays: they leave together */
#else
        if( got == 0 ) break;
#endif
        bool hit = false;
        HardPath r;
        r.pos = mk( 0, 0, 0 ); r.d = mk( 0, 0, 1 ); r.T = mk( 0, 0, 0 ); r.intensity = 0; r.depth = 0; r.pixel = ACN_INVALID;
        Trans trans;
        trans.exit_nor = mk( 0, 0, 0 ); trans.exit_obj = -1; trans.enter_obj = -1;
        double a = F3_INF;
        if( ( threadIdx.x & 63 ) < got ) r = recs[ first + ( threadIdx.x & 63 ) ];
#if ACN_POOLED
        if constexpr( LDS ) a = root_trans_hit_pooled( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, ( LdsNodeP )acn_lds_raw ), pool, sc.matter_root, r.pixel != ACN_INVALID, r.pos, r.d, &trans, &cnt );
        else                a = root_trans_hit_pooled( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, sc.nodes ), pool, sc.matter_root, r.pixel != ACN_INVALID, r.pos, r.d, &trans, &cnt );
#endif
        if( r.pixel != ACN_INVALID )
        {
            ACN_LAP( PH_FETCH );
#if !ACN_POOLED
            if constexpr( LDS ) a = root_trans_hit( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, ( LdsNodeP )acn_lds_raw ), sc.matter_root, r.pos, r.d, &trans, &cnt );
            else                a = root_trans_hit( scene_view< PRUNE, ACN_PARK_ORIGIN != 0 >( sc, sc.nodes ), sc.matter_root, r.pos, r.d, &trans, &cnt );
#endif
            ACN_LAP( PH_ROOT_LEAF );
            hit = a < sc.prm.max_path_length;
            if( !hit )
            {
                V3 c = v_mlf( ld3( sc.prm.background_color ), r.intensity );
                pixel_add( accum, sc.flags, r.pixel, v_mld( r.T, c ) );
            }
        }
        uint32_t csl = chunk_alloc( cs + 0, &p_counts[ QC_CHILDREN ], hit );
        if( hit )
        {
            if( csl < child_cap )
            {
                HitRec& c = p_children[ csl ];
                c.p = r.pos; c.d = r.d; c.offs = a; c.exit_nor = trans.exit_nor; c.T = r.T;
                c.intensity = r.intensity;
                c.exit_obj = trans.exit_obj; c.enter_obj = trans.enter_obj;
                c.depth = r.depth; c.pixel = r.pixel;
                n_ch++;
            }
            else
            {
                atomicOr( sc_in.flags, ACN_FLAG_CHILD_OVERFLOW );
            }
        }
    }
    ACN_LAP( PH_SHADE );
    ACN_PHASE_FLUSH( counters, 2 )
    chunk_close( cs + 0, child_cap, kill_ch, p_counts + QS_DEAD_C );
    wave_stat_add( p_counts + QS_CHILDREN, n_ch );
    wave_add_counters( counters, cnt );
}

#endif /* ACN_PIPELINE_H */
