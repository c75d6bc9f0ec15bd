/* acn_device.h -- gfx950 device code of the trace / radiance path (included only by actinon_hip.hip).
 *
 * What the reference does with recursion over heap objects and vtables (src/objects.c, src/compound.c,
 * src/scene.c:420-667) is done here with index-linked POD nodes, explicit per-lane stacks and state machines:
 *   - hit_machine   : obj_ray_hit over CSG trees (pair_inside / pair_outside / neg / scale wrappers) as a
 *                     coroutine-style state machine; all lanes of a wave that are "at a leaf" execute the same
 *                     plane / sphere / squaroid / SDF code together regardless of their depth in the tree.
 *   - side_machine  : obj_side as an iterative boolean-tree evaluation (short-circuited; children are pure).
 *   - compound walk : closest hit / any hit / transition hit over (nested) compounds with an explicit stack.
 *   - lum evaluator : scene_s_lum's branching recursion turned into a throughput-carrying work stack (pending
 *                     rays) plus a small stack of suspended path loops.
 * Floating-point expressions are written in the reference's order and compiled with -ffp-contract=off, so
 * geometry, seeds and control flow are bit-identical to the CPU oracle; only the radiance sums are
 * re-associated (throughput form), which moves colours by ~1e-16 relative.
 */
#ifndef ACN_DEVICE_H
#define ACN_DEVICE_H

#include <hip/hip_runtime.h>
#include "actinon_hip.h"
#include "acn_detmath.h"
#include "acn_costs.h"

#pragma clang fp contract(off)

#define DEV __device__ __forceinline__
#define DEVN __device__ __noinline__
/* inlining policy of the two CSG machines (measured on MI355X) */
#ifndef ACN_SIDE_INLINE
#define ACN_SIDE_INLINE 1
#endif
#ifndef ACN_HIT_INLINE
#define ACN_HIT_INLINE 1
#endif
#if ACN_SIDE_INLINE
#define DEV_SIDE __device__ __forceinline__
#else
#define DEV_SIDE __device__ __noinline__
#endif
#if ACN_HIT_INLINE
#define DEV_HIT __device__ __forceinline__
#else
#define DEV_HIT __device__ __noinline__
#endif

#define F3_INF ( __builtin_huge_val() )
#define F3_MAG 1E+30
#define F3_EPS 1E-6
#define ACN_PI 3.14159265358979323846

#ifndef ACN_CSG_MAX_DEPTH
#define ACN_CSG_MAX_DEPTH   24   /* nesting of pair / neg / scale wrappers */
#endif
#ifndef ACN_CMP_MAX_DEPTH
#define ACN_CMP_MAX_DEPTH   12   /* nesting of compounds */
#endif
#define ACN_TASK_STACK      64   /* pending rays per lane */
#define ACN_MAX_PATH_LEVELS 5    /* suspended path loops: trace_depth <= 10 * 5 + 10 */

struct V3 { double x, y, z; };
struct M3 { V3 x, y, z; };
struct Ray { V3 p, d; };

/* Device layout of a node: geometry only (192 B); shading properties live in GMat (96 B, read once per shading
 * point).  Built from the ABI's acn_node by acn_scene_upload. */
struct GNode
{
    int32_t  type;
    uint32_t flags;
    int32_t  child0, child1;
    double   prm[ 4 ];
    double   pos[ 3 ];
    double   env_pos[ 3 ];
    double   env_radius;
    double   rax[ 9 ];
    double   surface_roughness;
    int32_t  sdf_kind, cycles;
};

struct GMat
{
    double color[ 3 ];
    double radiance;
    double refractive_index;
    double fresnel_reflectivity;
    double chromatic_reflectivity;
    double diffuse_reflectivity;
    double sigma;
    double transparency[ 3 ];
    int32_t texture;     /* index into the texture table, -1 = none */
    int32_t pad_;
};

/* The scene arrays are read through the CONSTANT address space: they never change while a kernel runs, so a load
 * whose address is the same in every lane (root-compound loops, a wave's shading task) is issued as a scalar load
 * (s_load -> SGPRs, scalar cache) and costs neither vector-memory bandwidth nor VGPRs; loads with per-lane
 * addresses (inside CSG trees) become ordinary global loads. */
#define ACN_CONST __attribute__( ( address_space( 4 ) ) )
typedef const GNode   ACN_CONST* NodeP;
typedef const GMat    ACN_CONST* MatP;
typedef const int32_t ACN_CONST* ElemP;
typedef const double  ACN_CONST* CDblP;
typedef const acn_texture ACN_CONST* TexP;

/* node array staged in LDS (per workgroup) for the kernels whose node accesses are per-lane */
#define ACN_LDS __attribute__( ( address_space( 3 ) ) )
typedef const GNode ACN_LDS* LdsNodeP;
typedef double ACN_LDS* LdsF64P;
typedef uint32_t ACN_LDS* LdsU32P;
/* dynamic LDS of the machine kernels: [ staged node array (optional) ][ CSG stacks of the block's 256 lanes ] */
extern __shared__ __attribute__( ( aligned( 16 ) ) ) double acn_lds_raw[];
/* ACN_POOLED (acn_pipeline.h; built in round 4, measured slower, OFF): the machine kernels pool the rays of a workgroup's four
 * waves per root element */
#ifndef ACN_POOLED
#define ACN_POOLED 0
#endif
#ifndef ACN_LDS_DEPTH
#if ACN_POOLED
#define ACN_LDS_DEPTH 2                 /* (the third level's 10 KB per workgroup go to the ray pool) */
#else
#define ACN_LDS_DEPTH 3                 /* stack levels kept in LDS; deeper nesting continues in scratch */
#endif
#endif
#define ACN_LDS_LANES 256               /* block size of the kernels that provide the stack area */
/* per level and lane: a 8 B, parked normal 24 B, w 4 B, side 4 B; doubles first (alignment):
 * [ a : D x 256 ][ nx, ny, nz : 3 x D x 256 ][ w : D x 256 ][ side : D x 256 ] */
#define ACN_LDS_STACK_BYTES ( ACN_LDS_DEPTH * ACN_LDS_LANES * 40 )
#define ACN_NO_LDS_STACK 0xFFFFFFFFu
/* behind the stacks: the ray pool of the workgroup (pooled_machine_hit): six planes of 256 doubles (ray in / result out), one
 * plane of 256 words (owners), two sets of the four waves' counts */
#if ACN_POOLED
#define ACN_LDS_POOL_BYTES ( 6 * ACN_LDS_LANES * 8 + ACN_LDS_LANES * 4 + 64 )
#else
#define ACN_LDS_POOL_BYTES 0
#endif
/* behind those: the ray origin of the lock-step machine at hand, one per lane (OrgLds): three planes of 256 doubles */
#ifndef ACN_PARK_ORIGIN
#define ACN_PARK_ORIGIN 1
#endif
#if ACN_PARK_ORIGIN
#define ACN_LDS_ORG_BYTES ( 3 * ACN_LDS_LANES * 8 )
#else
#define ACN_LDS_ORG_BYTES 0
#endif

/* One entry of a simple compound's pre-order table (simple_compound_hit): everything a visit needs -- the element's
 * envelope, its type and the two links -- in ONE 48-byte record, i.e. one memory round trip per visited node instead
 * of three dependent ones (element index -> node header -> envelope). */
struct SCEntry
{
    double  env_pos[ 3 ], env_radius;
    int32_t node;        /* node index (leaves: the object that is hit) */
    int32_t skip;        /* entry behind this element's subtree */
    int32_t type;        /* acn_node_type */
    uint32_t flags;      /* ACN_NODE_HAS_ENVELOPE | ACN_SC_SPHERE ( skip = index into sc_spheres ) | ACN_SC_ROUGH */
};
#define ACN_SC_SPHERE 0x10000u
#define ACN_SC_ROUGH  0x20000u
#define ACN_SC_BOUNDING 0x40000u   /* the upload step has verified that the envelope contains every leaf below the entry (all of them spheres) */
#ifndef ACN_SC_CULL
#define ACN_SC_CULL 1
#endif
#ifndef ACN_SC_TWO_ORDERS
#define ACN_SC_TWO_ORDERS 1
#endif
#ifndef ACN_SC_EARLY_NEXT
#define ACN_SC_EARLY_NEXT 0
#endif
#ifndef ACN_SC_DEFER
#define ACN_SC_DEFER 0
#endif
#ifndef ACN_SC_SPHERE_TABLE
#define ACN_SC_SPHERE_TABLE 1
#endif

/* device-resident scene, parameterised by where the node array is read from */
template< class NP >
struct DevSceneT
{
    NP nodes;
    const GNode ACN_CONST* gnodes;   /* the node array in global memory, whatever `nodes` points to (wave-uniform reads: scalar loads) */
    MatP  mats;
    ElemP elems;
    TexP  textures;
    int32_t light_root, matter_root;
    uint32_t n_nodes, n_elems;
    acn_params prm;
    /* camera basis, computed on the device by k_camera_setup with the same expressions as the oracle */
    M3 camera_rotation;
    double unit_f;
    uint32_t* flags;     /* device word for ACN_FLAG_* error bits */
    uint32_t lds_stack;  /* byte offset of the CSG stack area in dynamic LDS, ACN_NO_LDS_STACK if the kernel has none */
    uint32_t prune_base; /* elems[ prune_base + node ]: offset of the node's prune program in elems[], or -1 */
    uint32_t class0_min; /* shading tasks with more samples than this run on 64 lanes (ACN_CLASS0_MIN, acn_pipeline.h: size_class) */
    const SCEntry* sc_table;   /* pre-order tables of the simple compounds */
    const double* sc_spheres;  /* ( pos, radius ) of the sphere leaves of those tables, see SCEntry.flags */
    CDblP env_tab;             /* per entry of elems[ 0 .. 2 n_elems ): envelope centre and radius of that element ( radius < 0: none ), see root_candidates */
    static constexpr bool prune = false;
    static constexpr bool park = false;   /* the lock-step machines keep the ray origin in LDS (OrgLds): only where the kernel owns the slot */
};
/* the same scene for the "extras" kernel variants: interval-prune programs and in-line simple compounds.  Launched
 * only when the upload step produced either, so that the kernels of plain scenes (wine_glass) do not carry their code */
template< class NP > struct DevScenePT : DevSceneT< NP > { static constexpr bool prune = true; };
template< class BASE > struct ParkedScene : BASE { static constexpr bool park = true; };
typedef DevSceneT< NodeP > DevScene;

/* the same scene with its node array read from another address space */
template< class NP2 >
__device__ __forceinline__ DevSceneT< NP2 > scene_rebind( const DevScene& sc, NP2 nodes )
{
    DevSceneT< NP2 > r;
    r.nodes = nodes; r.gnodes = sc.nodes; r.mats = sc.mats; r.elems = sc.elems; r.textures = sc.textures;
    r.light_root = sc.light_root; r.matter_root = sc.matter_root; r.n_nodes = sc.n_nodes; r.n_elems = sc.n_elems;
    r.prm = sc.prm; r.camera_rotation = sc.camera_rotation; r.unit_f = sc.unit_f; r.flags = sc.flags; r.lds_stack = sc.lds_stack; r.prune_base = sc.prune_base; r.class0_min = sc.class0_min; r.sc_table = sc.sc_table; r.sc_spheres = sc.sc_spheres; r.env_tab = sc.env_tab;
    return r;
}

template< bool PR, bool PK = false, class NP2 >
__device__ __forceinline__ auto scene_view( const DevScene& sc, NP2 nodes )
{
    if constexpr( PR )
    {
        if constexpr( PK ) { ParkedScene< DevScenePT< NP2 > > r; static_cast< DevSceneT< NP2 >& >( r ) = scene_rebind( sc, nodes ); return r; }
        else               { DevScenePT< NP2 > r; static_cast< DevSceneT< NP2 >& >( r ) = scene_rebind( sc, nodes ); return r; }
    }
    else
    {
        if constexpr( PK ) { ParkedScene< DevSceneT< NP2 > > r; static_cast< DevSceneT< NP2 >& >( r ) = scene_rebind( sc, nodes ); return r; }
        else return scene_rebind( sc, nodes );
    }
}

enum
{
    CNT_TRANS_RAY = 0, CNT_SHADOW_RAY, CNT_OBJ_HIT, CNT_LUM, CNT_CAP_SAMPLE, CNT_SIDE, CNT_SDF_EVAL, CNT_OVERFLOW, CNT_N
};

/* per-lane event counts of one launch (32 bit each), wave-reduced into 64-bit global counters at kernel end.
 * Cnt< false > compiles to nothing: the counting kernels are only used when ACN_OPT_COUNT_WORK is set. */
template< bool ON > struct Cnt;
template<> struct Cnt< true >
{
    static constexpr bool counting = true;
    unsigned c[ CNT_N ];
    unsigned long long flop;   /* sum of event costs, acn_costs.h */
    unsigned transc;           /* transcendental calls */
    __device__ __forceinline__ void clear() { for( int k = 0; k < CNT_N; k++ ) c[ k ] = 0; flop = 0; transc = 0; }
    __device__ __forceinline__ void inc( int k ) { c[ k ]++; }
    __device__ __forceinline__ void add( int k, unsigned v ) { c[ k ] += v; }
    __device__ __forceinline__ void cost( unsigned f, unsigned t = 0 ) { flop += f; transc += t; }
};
template<> struct Cnt< false >
{
    static constexpr bool counting = false;
    __device__ __forceinline__ void clear() {}
    __device__ __forceinline__ void inc( int ) {}
    __device__ __forceinline__ void add( int, unsigned ) {}
    __device__ __forceinline__ void cost( unsigned, unsigned = 0 ) {}
};
/* the leaf routines below take the lane's counters as their last argument; call sites are written without it and the
 * macros behind each definition append the `cnt` of the calling function (our own pruning tests, which are not events
 * of the reference's algorithm, call the underscore forms with ACN_NO_CNT) */
#define ACN_NO_CNT ( ( Cnt< false >* )nullptr )

/* Phase timers (diagnostic builds only: make ... EXTRA=-DACN_PHASE_TIMERS).  Every wave keeps a time stamp and ACN_PH_N
 * accumulators in LDS; ACN_LAP( k ) books the shader-clock time since the wave's previous mark on phase k, whichever
 * lanes are active.  phase_flush adds the wave's sums to counters[ CNT_N + 2 + 16 * kernel + k ] at kernel end. */
#define ACN_PH_N 16
#define ACN_PH_KERNELS 4
#define ACN_CNT_SLOTS ( CNT_N + 2 + ACN_PH_N * ACN_PH_KERNELS )
enum { PH_OTHER = 0, PH_LIGHT, PH_ROOT_LEAF, PH_PRUNE, PH_M_LEAF, PH_M_PAIR, PH_M_FRAME, PH_M_SIDE, PH_SHADE, PH_COMPOUND, PH_FETCH, PH_TAIL };
#ifdef ACN_PHASE_TIMERS
__shared__ unsigned long long acn_phase_lds[ 4 ][ ACN_PH_N + 1 ];
__device__ __forceinline__ void phase_lap( int k )
{
    unsigned long long now = __builtin_readcyclecounter();
    unsigned long long ex = __ballot( 1 );
    if( ( int )( threadIdx.x & 63 ) == __ffsll( ( long long )ex ) - 1 )
    {
        unsigned long long* w = acn_phase_lds[ threadIdx.x >> 6 ];
        w[ 1 + k ] += now - w[ 0 ];
        w[ 0 ] = now;
    }
}
__device__ __forceinline__ void phase_init()
{
    if( ( threadIdx.x & 63 ) == 0 )
    {
        unsigned long long* w = acn_phase_lds[ threadIdx.x >> 6 ];
        for( int k = 1; k <= ACN_PH_N; k++ ) w[ k ] = 0;
        w[ 0 ] = __builtin_readcyclecounter();
    }
}
__device__ __forceinline__ void phase_flush( unsigned long long* counters, int kernel )
{
    if( ( threadIdx.x & 63 ) == 0 )
    {
        unsigned long long* w = acn_phase_lds[ threadIdx.x >> 6 ];
        for( int k = 0; k < ACN_PH_N; k++ ) if( w[ 1 + k ] ) atomicAdd( &counters[ CNT_N + 2 + ACN_PH_N * kernel + k ], w[ 1 + k ] );
    }
}
/* diagnostic tallies in the same slots: [ 12 ] lanes that entered a lock-step machine, [ 13 ] machine entries (waves),
 * [ 14 ] lanes active at leaf evaluations inside it, [ 15 ] leaf evaluations */
__device__ __forceinline__ void phase_tally( int k, bool x )
{
    unsigned long long ex = __ballot( 1 ), m = __ballot( x );
    if( ( int )( threadIdx.x & 63 ) == __ffsll( ( long long )ex ) - 1 )
    {
        unsigned long long* w = acn_phase_lds[ threadIdx.x >> 6 ];
        w[ 1 + k ] += ( unsigned long long )__popcll( m );
        w[ 2 + k ] += 1;
    }
}
#define ACN_TALLY( k, x ) phase_tally( k, x )
#define ACN_LAP( k ) phase_lap( k )
#define ACN_PHASE_INIT phase_init();
#define ACN_PHASE_FLUSH( counters, kernel ) phase_flush( counters, kernel );
#else
#define ACN_TALLY( k, x )
#define ACN_LAP( k )
#define ACN_PHASE_INIT
#define ACN_PHASE_FLUSH( counters, kernel )
#endif

/* the two read-only scene arrays, passed BY VALUE into the non-inlined machines (global address space, so the
 * scene struct is never forced into scratch) */
template< class NP > struct SceneRefT { NP nodes; const GNode ACN_CONST* gnodes; ElemP elems; uint32_t* flags; uint32_t n_elems; uint32_t lds_stack; };
#define ACN_FLAG_TASK_OVERFLOW  1u
#define ACN_FLAG_CHILD_OVERFLOW 2u
#define ACN_FLAG_STACK_OVERFLOW 4u   /* CSG / compound / ray stack exhausted: the result would be wrong, the call fails */
#define ACN_FLAG_CLAMPED        8u   /* a single pixel contribution exceeded the fixed-point clamp (reported, not an error) */
template< class NP > __device__ __forceinline__ SceneRefT< NP > sref( const DevSceneT< NP >& sc ) { SceneRefT< NP > r; r.nodes = sc.nodes; r.gnodes = sc.gnodes; r.elems = sc.elems; r.flags = sc.flags; r.n_elems = sc.n_elems; r.lds_stack = sc.lds_stack; return r; }

/* ---- vectors.h ---- */
DEV V3 mk( double x, double y, double z ) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
template< class P > DEV V3 ld3( P p ) { return mk( p[ 0 ], p[ 1 ], p[ 2 ] ); }
DEV V3 ldc( const V3 ACN_CONST& v ) { return mk( v.x, v.y, v.z ); }
DEV V3 v_neg( V3 o ) { return mk( -o.x, -o.y, -o.z ); }
DEV double v_sqr( V3 o ) { return ( o.x * o.x ) + ( o.y * o.y ) + ( o.z * o.z ); }
DEV V3 v_add( V3 o, V3 s ) { return mk( o.x + s.x, o.y + s.y, o.z + s.z ); }
DEV V3 v_sub( V3 o, V3 s ) { return mk( o.x - s.x, o.y - s.y, o.z - s.z ); }
DEV V3 v_mlf( V3 o, double f ) { return mk( o.x * f, o.y * f, o.z * f ); }
DEV V3 v_mlx( V3 o, V3 f ) { return mk( o.y * f.z - o.z * f.y, o.z * f.x - o.x * f.z, o.x * f.y - o.y * f.x ); }
DEV V3 v_mld( V3 o, V3 f ) { return mk( o.x * f.x, o.y * f.y, o.z * f.z ); }
DEV double v_mlv( V3 o, V3 m ) { return ( o.x * m.x ) + ( o.y * m.y ) + ( o.z * m.z ); }
DEV double v_sub_mlv( V3 o, V3 s, V3 m ) { return ( ( o.x - s.x ) * m.x ) + ( ( o.y - s.y ) * m.y ) + ( ( o.z - s.z ) * m.z ); }
DEV double f_sqr( double a ) { return a * a; }
DEV double v_diff_sqr( V3 o, V3 v ) { return f_sqr( o.x - v.x ) + f_sqr( o.y - v.y ) + f_sqr( o.z - v.z ); }
DEV double f_max( double a, double b ) { return a > b ? a : b; }
DEV double f_min( double a, double b ) { return a < b ? a : b; }
DEV double f_abs( double a ) { return a < 0 ? -a : a; }

DEV V3 v_of_length( V3 o, double a )   /* vectors.h:148-154 */
{
    double r_sqr = v_sqr( o );
    if( acn_fabs( r_sqr - 1.0 ) < 1E-8 ) return o;
    double f = r_sqr > 0 ? ( a / acn_sqrt( r_sqr ) ) : 0;
    return mk( o.x * f, o.y * f, o.z * f );
}

DEV V3 v_von( V3 o, V3 v )   /* vectors.h:157-162 */
{
    V3 o_n = v_of_length( o, 1.0 );
    v = v_sub( v, v_mlf( o_n, v_mlv( o_n, v ) ) );
    return v_of_length( v, 1.0 );
}

DEV V3 v_con( V3 o )   /* vectors.h:165-175 */
{
    double xx = o.x * o.x;
    double yy = o.y * o.y;
    double zz = o.z * o.z;
    V3 v;
    v.x = ( ( xx <= yy ) && ( xx <= zz ) ) ? 1 : 0;
    v.y = ( ( yy <= xx ) && ( yy <= zz ) ) ? 1 : 0;
    v.z = ( ( zz <= xx ) && ( zz <= yy ) ) ? 1 : 0;
    return v_von( o, v );
}

DEV V3 m_mlv( const M3& o, V3 v )   /* vectors.h:256-265 */
{
    return mk( o.x.x * v.x + o.x.y * v.y + o.x.z * v.z,
               o.y.x * v.x + o.y.y * v.y + o.y.z * v.z,
               o.z.x * v.x + o.z.y * v.y + o.z.z * v.z );
}

DEV V3 m_tmlv( const M3& o, V3 v )   /* vectors.h:268-276 */
{
    return mk( o.x.x * v.x + o.y.x * v.y + o.z.x * v.z,
               o.x.y * v.x + o.y.y * v.y + o.z.y * v.z,
               o.x.z * v.x + o.y.z * v.y + o.z.z * v.z );
}

DEV M3 m_transposed( const M3& o )
{
    M3 r;
    r.x = mk( o.x.x, o.y.x, o.z.x );
    r.y = mk( o.x.y, o.y.y, o.z.y );
    r.z = mk( o.x.z, o.y.z, o.z.z );
    return r;
}

DEV M3 m_con_z( V3 v )   /* vectors.h:315-322 */
{
    M3 m;
    m.z = v_of_length( v, 1.0 );
    m.x = v_con( v );
    m.y = v_mlx( m.z, m.x );
    return m;
}

DEV V3 ray_pos( V3 p, V3 d, double offs ) { return v_add( p, v_mlf( d, offs ) ); }   /* vectors.h:343-346 */

DEV V3 v_orthogonal_projection( V3 o, V3 nor )   /* vectors.h:223-232 */
{
    double f = v_mlv( o, nor );
    return mk( o.x - nor.x * f, o.y - nor.y * f, o.z - nor.z * f );
}

DEV V3 v_reflection( V3 dir, V3 nor )   /* vectors.h:238-241 */
{
    return v_of_length( v_sub( dir, v_mlf( nor, 2.0 * v_mlv( dir, nor ) ) ), 1.0 );
}

/* ---- LCG / sampling: vectors.h:45-48, 177-206 ---- */
DEV uint64_t lcg00( uint64_t v ) { return v * ACN_LCG00_A + ACN_LCG00_C; }
DEV uint64_t lcg01( uint64_t v ) { return v * ACN_LCG01_A + ACN_LCG01_C; }
DEV uint64_t lcg02( uint64_t v ) { return v * ACN_LCG02_A + ACN_LCG02_C; }
DEV double f3_rnd0( uint64_t* rv ) { return ( double )( *rv = lcg00( *rv ) ) * ( 2.0 / 0xFFFFFFFFFFFFFFFFull ) - 1.0; }
DEV double f3_rnd1( uint64_t* rv ) { return ( double )( *rv = lcg00( *rv ) ) * ( 1.0 / 0xFFFFFFFFFFFFFFFFull ); }

/* advance an lcg00 state by k steps in O(log k) (Brown's arbitrary-stride formula) */
DEV uint64_t lcg00_jump( uint64_t v, uint64_t k )
{
    uint64_t a = ACN_LCG00_A, c = ACN_LCG00_C;
    uint64_t acc_a = 1, acc_c = 0;
    while( k )
    {
        if( k & 1 ) { acc_a *= a; acc_c = acc_c * a + c; }
        c = ( a + 1 ) * c;
        a *= a;
        k >>= 1;
    }
    return acc_a * v + acc_c;
}

DEV uint64_t seed_from_f3( double v )
{
    int64_t seed_s3 = ( int64_t )( acn_frexp_mant( v ) * ( double )0x7FFFFFFFFFFFFFFF );
    return ( uint64_t )seed_s3 * 27362149ull;
}

DEV uint64_t v_random_seed( V3 o, uint64_t rv )
{
    return seed_from_f3( o.x ) * lcg00( rv ) + seed_from_f3( o.y ) * lcg01( rv ) + seed_from_f3( o.z ) * lcg02( rv );
}

DEV V3 v_random_sphere_cap( uint64_t* rv, double h )
{
    V3 v;
    double phi = 2.0 * ACN_PI * f3_rnd1( rv );
    v.z = 1.0 - f3_rnd1( rv ) * h;
    double scale = acn_sqrt( 1.0 - v.z * v.z );
    double s, c;
    acn_sincos( phi, &s, &c );
    v.x = s * scale;
    v.y = c * scale;
    return v;
}

DEV V3 v_random_sphere_belt( uint64_t* rv, double h )   /* vectors.h:209-218 */
{
    V3 v;
    double phi = 2.0 * ACN_PI * f3_rnd1( rv );
    v.z = f3_rnd0( rv ) * h;
    double scale = acn_sqrt( 1.0 - v.z * v.z );
    double s, c;
    acn_sincos( phi, &s, &c );
    v.x = s * scale;
    v.y = c * scale;
    return v;
}

/* ---- gmath.h:38-97 ---- */
template< class CT > DEV double plane_ray_hit_( V3 pos, V3 nor, V3 rp, V3 rd, bool want_nor, V3* p_nor, CT* cnt )
{
    cnt->cost( ACN_F_PLANE_HIT );
    double div = v_mlv( nor, rd );
    if( div == 0 ) return F3_INF;
    double offs = v_sub_mlv( pos, rp, nor ) / div;
    if( want_nor ) *p_nor = nor;
    return ( offs > 0 ) ? offs - F3_EPS : F3_INF;
}
#define plane_ray_hit( ... ) plane_ray_hit_( __VA_ARGS__, cnt )

template< class CT > DEV double sphere_ray_hit_( V3 pos, double r, V3 rp, V3 rd, bool want_nor, V3* p_nor, CT* cnt )
{
    V3 p = v_sub( rp, pos );
    double s = v_mlv( p, rd );
    double q = v_sqr( p ) - ( r * r );
    double s2 = s * s;
    if( s2 < q ) { cnt->cost( ACN_F_SPHERE_MISS ); return F3_INF; }
    double offs = F3_INF;
    if( s < 0 && q > 0 )
    {
        offs = -s - acn_sqrt( s2 - q ) - F3_EPS;
    }
    else if( s < 0 || q < 0 )
    {
        offs = -s + acn_sqrt( s2 - q ) - F3_EPS;
    }
    if( offs < F3_INF && want_nor ) *p_nor = v_of_length( v_sub( ray_pos( rp, rd, offs ), pos ), 1.0 );
    cnt->cost( offs < F3_INF ? ( want_nor ? ACN_F_SPHERE_HIT_NOR : ACN_F_SPHERE_HIT ) : ACN_F_SPHERE_MISS );
    return offs;
}
#define sphere_ray_hit( ... ) sphere_ray_hit_( __VA_ARGS__, cnt )

template< class CT > DEV int sphere_observer_side_( V3 pos, double r, V3 observer, CT* cnt )
{
    cnt->cost( ACN_F_SIDE_SPHERE );
    V3 diff = v_sub( observer, pos );
    return ( v_sqr( diff ) > r * r ) ? 1 : -1;
}
#define sphere_observer_side( ... ) sphere_observer_side_( __VA_ARGS__, cnt )

/* ---- gmath.c:68-113 ---- */
DEV double fresnel_reflection( V3 dir_i, V3 exit_nor, double trix, V3* dir )
{
    double c = v_mlv( dir_i, exit_nor );
    double f = c < 0 ? trix : 1.0 / trix;
    double cos_ai = acn_fabs( c );
    cos_ai = cos_ai > 1.0 ? 1.0 : cos_ai;
    double sin_ai = acn_sqrt( 1.0 - cos_ai * cos_ai );
    double sin_at = sin_ai * f;
    double reflectance = 1.0;
    if( sin_at < 1 )
    {
        double cos_at = acn_sqrt( 1.0 - sin_at * sin_at );
        double rs = f_sqr( ( f * cos_ai - cos_at ) / ( f * cos_ai + cos_at ) );
        double rp = f_sqr( ( f * cos_at - cos_ai ) / ( f * cos_at + cos_ai ) );
        reflectance = ( rs + rp ) * 0.5;
    }
    *dir = v_reflection( dir_i, exit_nor );
    return reflectance;
}

DEV V3 fresnel_refraction( V3 dir_i, V3 exit_nor, double trix )
{
    double c = v_mlv( dir_i, exit_nor );
    double f = c < 0 ? trix : 1.0 / trix;
    double a = f;
    double q = f * f * ( 1.0 - c * c );
    if( q < 1.0 )
    {
        double b = -f * c + ( c > 0 ? acn_sqrt( 1.0 - q ) : -acn_sqrt( 1.0 - q ) );
        return v_add( v_mlf( dir_i, a ), v_mlf( exit_nor, b ) );
    }
    return dir_i;
}

/* ---- node access ---- */
/* A node visit whose index is the same in every lane reads the node through the scalar cache.  Reading field by field,
 * behind the branches that need them (type -> envelope -> position -> axes), makes every visit a chain of three or four
 * DEPENDENT scalar loads of ~150-200 cycles each, which is what the lock-step machines wait for most of their time
 * (profiles/r02/pmc_sq_final.txt: 0.6 of k_walk's wave cycles in s_waitcnt, 6e8 scalar loads per frame).  NodeView copies
 * the whole 192-byte record into SGPRs at once -- three s_load_dwordx16 issued back to back, one wait -- and hands out
 * a pointer to the copy; node arrays in other address spaces (LDS: per-lane indices) are passed through. */
template< class NP > struct NodeView { NP p; DEV NodeView( NP q ) : p( q ) {} DEV NP ptr() const { return p; } };
#ifdef ACN_PRELOAD_NODES
template<> struct NodeView< const GNode ACN_CONST* >
{
    GNode v;
    DEV NodeView( const GNode ACN_CONST* q )
    {
        v.type = q->type; v.flags = q->flags; v.child0 = q->child0; v.child1 = q->child1;
        for( int k = 0; k < 4; k++ ) v.prm[ k ] = q->prm[ k ];
        for( int k = 0; k < 3; k++ ) { v.pos[ k ] = q->pos[ k ]; v.env_pos[ k ] = q->env_pos[ k ]; }
        v.env_radius = q->env_radius;
        for( int k = 0; k < 9; k++ ) v.rax[ k ] = q->rax[ k ];
        v.surface_roughness = q->surface_roughness; v.sdf_kind = q->sdf_kind; v.cycles = q->cycles;
    }
    DEV const GNode* ptr() const { return &v; }
};
#endif
#define ACN_NODE( name, expr ) const auto name##_view_ = NodeView< decltype( expr ) >( expr ); const auto name = name##_view_.ptr();
#define ACN_NODE_UNIFORM( name, expr ) ACN_NODE( name, expr )

template< class NP > DEV bool node_has_env( NP n ) { return ( n->flags & ACN_NODE_HAS_ENVELOPE ) != 0; }
template< class NP > DEV M3 node_rax( NP n )
{
    M3 m;
    m.x = ld3( n->rax ); m.y = ld3( n->rax + 3 ); m.z = ld3( n->rax + 6 );
    return m;
}
/* envelope_s_ray_hits (objects.c:90-93) = sphere_ray_hit( ... ) < f3_inf.  Only the predicate is needed: by
 * gmath.h:64-83 the offset is finite exactly when s*s >= q and ( s < 0 or q < 0 ) -- the square root of the
 * non-negative discriminant is finite for finite inputs -- so the sqrt is not evaluated. */
template< class NP, class CT > DEV bool env_ray_hits_( NP n, V3 rp, V3 rd, CT* cnt );
/* the same on an envelope given by value */
template< class CT > DEV bool env_ray_hits_raw( V3 env_pos, double r, V3 rp, V3 rd, CT* cnt )
{
    V3 p = v_sub( rp, env_pos );
    double s = v_mlv( p, rd );
    double q = v_sqr( p ) - ( r * r );
    double s2 = s * s;
    bool hit = !( s2 < q ) && ( ( s < 0 && q > 0 ) || ( s < 0 || q < 0 ) );
    cnt->cost( hit ? ACN_F_ENV_HIT : ACN_F_ENV_MISS );
    return hit;
}
template< class NP, class CT > DEV bool env_ray_hits_( NP n, V3 rp, V3 rd, CT* cnt )
{
    V3 p = v_sub( rp, ld3( n->env_pos ) );
    double r = n->env_radius;
    double s = v_mlv( p, rd );
    double q = v_sqr( p ) - ( r * r );
    double s2 = s * s;
    bool hit = !( s2 < q ) && ( ( s < 0 && q > 0 ) || ( s < 0 || q < 0 ) );
    cnt->cost( hit ? ACN_F_ENV_HIT : ACN_F_ENV_MISS );
    return hit;
}
#define env_ray_hits( ... ) env_ray_hits_( __VA_ARGS__, cnt )
template< class NP, class CT > DEV int env_side_( NP n, V3 pos, CT* cnt ) { return sphere_observer_side_( ld3( n->env_pos ), n->env_radius, pos, cnt ); }
#define env_side( ... ) env_side_( __VA_ARGS__, cnt )

/* ---- distance.c:39-42, 83-92 ---- */
template< class NP, class CT > DEV double sdf_eval_( NP n, V3 pos, CT* cnt )
{
    if( n->sdf_kind == ACN_SDF_TORUS )
    {
        cnt->cost( ACN_F_SDF_TORUS );
        double x = pos.x;
        double y = pos.y;
        double f = acn_sqrt( x * x + y * y );
        double f_inv = ( f > 0 ) ? ( 1.0 / f ) : 1.0;
        x *= f_inv;
        y *= f_inv;
        return acn_sqrt( f_sqr( x - pos.x ) + f_sqr( y - pos.y ) + f_sqr( pos.z ) ) - n->prm[ 1 ];
    }
    cnt->cost( ACN_F_SDF_SPHERE );
    return acn_sqrt( f_sqr( pos.x ) + f_sqr( pos.y ) + f_sqr( pos.z ) ) - 1.0;
}
#define sdf_eval( ... ) sdf_eval_( __VA_ARGS__, cnt )

/* ---- leaves ---- */
template< class NP, class CT > DEV double squaroid_ray_hit_( NP o, V3 rp, V3 rd, bool want_nor, V3* p_nor, CT* cnt )   /* objects.c:778-821 */
{
    cnt->cost( ACN_F_SQUAROID_MISS );
    M3 rax = node_rax( o );
    double oa = o->prm[ 0 ], ob = o->prm[ 1 ], oc = o->prm[ 2 ], orr = o->prm[ 3 ];
    V3 p = m_mlv( rax, v_sub( rp, ld3( o->pos ) ) );
    V3 d = m_mlv( rax, rd );
    double f  = oa * d.x * d.x + ob * d.y * d.y + oc * d.z * d.z;
    double fs = oa * d.x * p.x + ob * d.y * p.y + oc * d.z * p.z;
    double fq = oa * p.x * p.x + ob * p.y * p.y + oc * p.z * p.z + orr;
    double a = F3_INF;
    if( f != 0 )
    {
        double f_inv = 1.0 / f;
        double s = fs * f_inv;
        double q = fq * f_inv;
        double rr = s * s - q;
        if( rr < 0 ) return F3_INF;
        rr = acn_sqrt( rr );
        a = -s - rr;
        if( a < 0 ) a = -s + rr;
        if( a < 0 ) a = F3_INF;
    }
    else
    {
        a = ( fq != 0 ) ? -fs / ( 2 * fq ) : F3_INF;
    }
    if( a == F3_INF ) return F3_INF;
    cnt->cost( ( want_nor ? ACN_F_SQUAROID_HIT_NOR : ACN_F_SQUAROID_HIT ) - ACN_F_SQUAROID_MISS );
    if( want_nor )
    {
        double x = p.x + a * d.x;
        double y = p.y + a * d.y;
        double z = p.z + a * d.z;
        V3 n1 = mk( x * oa, y * ob, z * oc );
        *p_nor = v_of_length( m_tmlv( rax, n1 ), 1.0 );
    }
    return a - F3_EPS;
}
#define squaroid_ray_hit( ... ) squaroid_ray_hit_( __VA_ARGS__, cnt )

template< class NP, class CT > DEV int squaroid_side_( NP o, V3 pos, CT* cnt )   /* objects.c:823-827 */
{
    cnt->cost( ACN_F_SIDE_SQUAROID );
    M3 rax = node_rax( o );
    V3 p = m_mlv( rax, v_sub( pos, ld3( o->pos ) ) );
    return ( o->prm[ 0 ] * p.x * p.x + o->prm[ 1 ] * p.y * p.y + o->prm[ 2 ] * p.z * p.z + o->prm[ 3 ] ) > 0 ? 1 : -1;
}
#define squaroid_side( ... ) squaroid_side_( __VA_ARGS__, cnt )

template< class NP, class CT >
DEVN double distance_ray_hit( NP o, V3 rp, V3 rd, bool want_nor, V3* p_nor, CT* cnt )   /* objects.c:903-959 */
{
    M3 rax = node_rax( o );
    double inv_scale = o->prm[ 0 ];
    V3 p = rp;
    double offs0 = 0;
    if( node_has_env( o ) )
    {
        if( env_side( o, rp ) == 1 )
        {
            offs0 = sphere_ray_hit( ld3( o->env_pos ), o->env_radius, rp, rd, false, nullptr );
            if( offs0 >= F3_INF ) return F3_INF;
            p = ray_pos( rp, rd, offs0 );
        }
    }
    p = v_mlf( m_mlv( rax, v_sub( p, ld3( o->pos ) ) ), inv_scale );
    V3 d = m_mlv( rax, rd );

    double offs1 = 0;
    double dist = sdf_eval( o, p );
    unsigned evals = 1;
    int cycles = o->cycles;
    if( dist > 0 )
    {
        for( int i = 0; i < cycles; i++ )
        {
            offs1 += dist + F3_EPS;
            dist = sdf_eval( o, ray_pos( p, d, offs1 ) );
            evals++;
            if( dist < 0 || dist > F3_MAG ) break;
        }
    }
    else
    {
        for( int i = 0; i < cycles; i++ )
        {
            offs1 -= dist - F3_EPS;
            dist = sdf_eval( o, ray_pos( p, d, offs1 ) );
            evals++;
            if( dist > 0 || dist < -F3_MAG ) break;
        }
    }
    cnt->add( CNT_SDF_EVAL, evals );
    cnt->cost( ACN_F_SDF_RAY + ACN_F_SDF_STEP * ( evals - 1 ) );
    if( f_abs( dist ) <= F3_EPS )
    {
        if( want_nor )
        {
            cnt->cost( ACN_F_SDF_NORMAL );
            V3 q = ray_pos( p, d, offs1 );
            double d0 = sdf_eval( o, q );
            V3 n;
            n.x = ( sdf_eval( o, mk( q.x + F3_EPS, q.y, q.z ) ) - d0 ) / F3_EPS;
            n.y = ( sdf_eval( o, mk( q.x, q.y + F3_EPS, q.z ) ) - d0 ) / F3_EPS;
            n.z = ( sdf_eval( o, mk( q.x, q.y, q.z + F3_EPS ) ) - d0 ) / F3_EPS;
            *p_nor = v_of_length( m_tmlv( rax, n ), 1.0 );
        }
        return offs0 + ( offs1 / inv_scale ) - F3_EPS;
    }
    return F3_INF;
}

template< class NP, class CT > DEV int distance_side_( NP o, V3 pos, CT* cnt )   /* objects.c:961-966 */
{
    cnt->cost( ACN_F_SIDE_SDF );
    if( node_has_env( o ) && env_side( o, pos ) == 1 ) return 1;
    M3 rax = node_rax( o );
    V3 p = v_mlf( m_mlv( rax, v_sub( pos, ld3( o->pos ) ) ), o->prm[ 0 ] );
    return sdf_eval( o, p ) > 0 ? 1 : -1;
}
#define distance_side( ... ) distance_side_( __VA_ARGS__, cnt )

/* objects.c:267-282.  A REAL call (ACN_ROUGH_INLINE restores the in-line form): the perturbation sits behind every one of the
 * ~130 places where a hit returns a normal (leaves, operands, pairs, composites), three logarithms and a seed each -- in line
 * that was 35 - 40 % of k_walk's 1.4 MB of code, in scenes of which most (the wine glass, the diamond, many_spheres) have no rough
 * surface at all: the hot code was spread over twice the instruction-cache lines it needs.  A rough surface pays a call. */
#ifdef ACN_ROUGH_INLINE
#define DEV_ROUGH DEV
#else
#define DEV_ROUGH DEVN
#endif
DEV_ROUGH V3 roughness_apply( double surface_roughness, V3 n, V3 hit_pos )
{
    uint64_t rv = v_random_seed( hit_pos, 1246 );
    double f;
    f = f3_rnd0( &rv ) * 0.99;
    n.x += surface_roughness * acn_log( ( 1.0 - f ) / ( 1.0 + f ) );
    f = f3_rnd0( &rv ) * 0.99;
    n.y += surface_roughness * acn_log( ( 1.0 - f ) / ( 1.0 + f ) );
    f = f3_rnd0( &rv ) * 0.99;
    n.z += surface_roughness * acn_log( ( 1.0 - f ) / ( 1.0 + f ) );
    return v_of_length( n, 1.0 );
}
template< class NP, class CT > DEV V3 roughness_normal_( NP hdr, V3 n, V3 hit_pos, CT* cnt )
{
    cnt->cost( ACN_F_ROUGHNESS, ACN_T_ROUGHNESS );
    return roughness_apply( hdr->surface_roughness, n, hit_pos );
}
#define roughness_normal( ... ) roughness_normal_( __VA_ARGS__, cnt )

/* ------------------------------------------------------------------------------------------------------------------ */
/* Leaf pairs.  Most composites at the bottom of a CSG tree combine two simple operands -- a plane, sphere or squaroid,
 * possibly complemented (slabs, lenses, capped cylinders; 5 of the wine glass's 9 pairs, the whole liquid).  The upload
 * step marks such pairs (ACN_GFLAG_LEAF_PAIR) and both machines evaluate them in line: the same sequence of child
 * evaluations, side tests and walk steps as the general frames (objects.c:1052-1094 / 1209-1251, 1096-1099 / 1253-1256),
 * without frame, stack or nested loops. */
#define ACN_GFLAG_LEAF_PAIR 0x100u      /* device-only bits of GNode.flags: a level-1 pair ... */
#define ACN_GFLAG_PAIR2     0x400u      /* ... a level-2 pair: at least one operand is a level-1 pair (machines only) */
#define ACN_GFLAG_PRUNE_LEVELS_SHIFT 12  /* bits 12 - 14: see surely_outside */

template< class NP, class CT > DEV int simple_leaf_side_( NP g, V3 pos, CT* cnt )
{
    int type = g->type;
    if( type == ACN_PLANE )  { cnt->cost( ACN_F_SIDE_PLANE ); return v_sub_mlv( pos, ld3( g->pos ), ld3( g->rax + 6 ) ) > 0 ? 1 : -1; }
    if( type == ACN_SPHERE ) return sphere_observer_side( ld3( g->pos ), g->prm[ 0 ], pos );
    return squaroid_side( g, pos );
}
#define simple_leaf_side( ... ) simple_leaf_side_( __VA_ARGS__, cnt )

/* Where a ray's ORIGIN lives while a CSG object is evaluated.  The machines of k_walk run at 128 VGPRs, and the value the
 * allocator gives up first is the origin of the step's ray: stored to scratch once and re-loaded at 97 places -- every leaf and
 * envelope test at a frame the ray enters unchanged (ISA of round 4, scripts/isa_scratch.py) -- ~60 % of the kernel's scratch loads
 * and most of its 22 GB per frame.  It is one value per lane for the whole evaluation of a frame, so the lock-step machine parks
 * it in LDS (three planes of 256 doubles behind the stacks and the pool) and the operand code reads it where it uses it: a
 * ds_read, ~64 cycles, no VMEM slot.  A pair's alternating walk continues from a DERIVED origin that lives for one operand
 * evaluation: a plain V3.  The operand / pair code takes either through org_get(). */
struct OrgLds
{
    volatile double ACN_LDS* p;
    DEV V3 get() const { return mk( p[ 0 ], p[ ACN_LDS_LANES ], p[ 2 * ACN_LDS_LANES ] ); }
    DEV void set( V3 v ) const { p[ 0 ] = v.x; p[ ACN_LDS_LANES ] = v.y; p[ 2 * ACN_LDS_LANES ] = v.z; }
};
DEV V3 org_get( const V3& v ) { return v; }
DEV V3 org_get( const OrgLds& o ) { return o.get(); }

/* Pairs of level L: both operands are a simple leaf, NEG( simple leaf ) or -- for L = 2 -- a level-1 pair.  The
 * functions below are the reference's obj_side / obj_ray_hit for such operands and pairs, recursion unrolled by L.
 * (Level-1 operands as real function calls instead of in-line expansion: 69.4 vs 57.4 ms on c2 -- calls spill.) */
template< int L, class SR, class NP, class CT > DEV int pair_side( SR sc, NP n, V3 pos, CT* cnt );
template< int L, bool SPLIT = false, class SR, class NP, class O, class CT > DEV double pair_hit( SR sc, NP n, O rp, V3 rd, bool want_nor, V3* nor, CT* cnt );

/* An operand that is NEG( simple leaf ) goes through the SAME copy of the leaf code as a plain one (the leaf's node is selected,
 * the sign applied afterwards): every in-line copy of an operand is one plane / sphere / squaroid routine, not two. */
template< int L, class SR, class CT > DEV int operand_side( SR sc, int c, V3 pos, CT* cnt )
{
    const auto cn = &sc.nodes[ c ];
    cnt->inc( CNT_SIDE );
    if( node_has_env( cn ) && env_side( cn, pos ) == 1 ) return 1;
    if constexpr( L > 1 ) { if( cn->flags & ACN_GFLAG_LEAF_PAIR ) return pair_side< L - 1 >( sc, cn, pos, cnt ); }
    const bool neg = cn->type == ACN_NEG;
    const auto g = neg ? &sc.nodes[ cn->child0 ] : cn;
    if( neg )
    {
        cnt->inc( CNT_SIDE );
        if( node_has_env( g ) && env_side( g, pos ) == 1 ) return -1;
    }
    const int r = simple_leaf_side( g, pos );
    return neg ? -r : r;
}

template< class NP, class CT > DEV double simple_leaf_hit_( NP g, V3 rp, V3 rd, bool want_nor, V3* nor, CT* cnt )
{
    int type = g->type;
    double a;
    if( type == ACN_PLANE )       a = plane_ray_hit( ld3( g->pos ), ld3( g->rax + 6 ), rp, rd, want_nor, nor );
    else if( type == ACN_SPHERE ) a = sphere_ray_hit( ld3( g->pos ), g->prm[ 0 ], rp, rd, want_nor, nor );
    else                          a = squaroid_ray_hit( g, rp, rd, want_nor, nor );
    if( want_nor && a < F3_INF && g->surface_roughness > 0 ) *nor = roughness_normal( g, *nor, ray_pos( rp, rd, a ) );
    return a;
}
#define simple_leaf_hit( ... ) simple_leaf_hit_( __VA_ARGS__, cnt )

template< int L, bool SPLIT = false, class SR, class O, class CT > DEV double operand_hit( SR sc, int c, O rp, V3 rd, bool want_nor, V3* nor, CT* cnt )
{
    const auto cn = &sc.nodes[ c ];
    cnt->inc( CNT_OBJ_HIT );
    if( node_has_env( cn ) && !env_ray_hits( cn, org_get( rp ), rd ) ) return F3_INF;
    if constexpr( L > 1 )
    {
        if( cn->flags & ACN_GFLAG_LEAF_PAIR )
        {
            double a = pair_hit< L - 1, SPLIT >( sc, cn, rp, rd, want_nor, nor, cnt );
            if( want_nor && a < F3_INF && cn->surface_roughness > 0 ) *nor = roughness_normal( cn, *nor, ray_pos( org_get( rp ), rd, a ) );
            return a;
        }
    }
    const bool neg = cn->type == ACN_NEG;
    const auto g = neg ? &sc.nodes[ cn->child0 ] : cn;
    if( neg )
    {
        cnt->inc( CNT_OBJ_HIT );
        if( node_has_env( g ) && !env_ray_hits( g, org_get( rp ), rd ) ) return F3_INF;
    }
    const double a = simple_leaf_hit( g, org_get( rp ), rd, want_nor, nor );
    if( neg && a < F3_INF && want_nor )
    {
        *nor = v_neg( *nor );                                                                  /* objects.c:1329-1339 */
        if( cn->surface_roughness > 0 ) *nor = roughness_normal( cn, *nor, ray_pos( org_get( rp ), rd, a ) );
    }
    return a;
}

template< int L, class SR, class NP, class CT > DEV int pair_side( SR sc, NP n, V3 pos, CT* cnt )
{
    int want = ( n->type == ACN_PAIR_INSIDE ) ? -1 : 1;
    if( operand_side< L >( sc, n->child0, pos, cnt ) != want ) return -want;
    return operand_side< L >( sc, n->child1, pos, cnt ) == want ? want : -want;
}

/* the pair's hit without its own envelope test and roughness (the caller does both, as for any node): objects.c:1052-1094 /
 * 1209-1251.  The operands' code is expanded in line (calls spill: level-1 operands as real calls cost 69.4 vs 57.4 ms on c2),
 * and how often decides the size of the machine kernels, which do not fit the instruction cache:
 * SPLIT: the pair node is the same in every lane (a root element tested by k_shade, every pair the lock-step machines evaluate):
 * the operand at hand is chosen by a SCALAR index -- its node stays wave-uniform, i.e. scalar loads from the scalar cache instead of
 * a dozen per-lane vector loads per step -- and each of the three places where the reference evaluates "one operand, then the
 * other" (both hits, the two candidate tests, a round of the alternating walk) is a two-trip scalar loop over ONE copy of the
 * operand code, entered by the lanes whose turn it is: two copies of operand_hit and two of operand_side per pair where
 * round 3 had four of each, a quarter of the code two levels deep.  A lane's own sequence of evaluations, and with it every
 * bit of its result, is the reference's. */
template< int L, bool SPLIT, class SR, class NP, class O, class CT > DEV double pair_hit( SR sc, NP n, O rp, V3 rd, bool want_nor, V3* nor, CT* cnt )
{
    const int want = ( n->type == ACN_PAIR_INSIDE ) ? -1 : 1;
    const int c0 = n->child0, c1 = n->child1;
    if constexpr( SPLIT )
    {
        V3 n1 = mk( 0, 0, 0 ), n2 = mk( 0, 0, 0 );
        double a1 = F3_INF, a2 = F3_INF;
        #pragma unroll 1
        for( int k = 0; k < 2; k++ )
        {
            V3 nn = mk( 0, 0, 0 );
            const double aa = operand_hit< L, SPLIT >( sc, k ? c1 : c0, rp, rd, want_nor, &nn, cnt );
            if( k ) { a2 = aa; n2 = nn; } else { a1 = aa; n1 = nn; }
        }
        /* the two candidates: a1 if it is the nearer one and lies on the wanted side of operand 1, else a2 against operand 0 */
        bool open = true;          /* the lane has no result yet */
        double res = F3_INF;
        #pragma unroll 1
        for( int k = 0; k < 2; k++ )
        {
            if( k && open && a2 >= F3_INF ) open = false;   /* res = f3_inf */
            const bool m = k ? open : a1 < a2;
            if( m )
            {
                cnt->cost( ACN_F_PAIR_STEP );
                const double ac = k ? a2 : a1;
                if( operand_side< L >( sc, k ? c0 : c1, ray_pos( org_get( rp ), rd, ac ), cnt ) == want ) { open = false; res = ac; if( want_nor ) *nor = k ? n2 : n1; }
            }
            else if( !k ) cnt->cost( ACN_F_PAIR_STEP );
        }
        if( !open ) return res;
        /* the alternating walk: operand 0 from the far candidate on, then operand 1, ... until a hit lies on the wanted side of
         * the other operand.  A round of the wave serves the lanes whose turn is operand 0, then those whose turn is operand 1 */
        double offs = a2;
        bool swapped = false;
        for( ;; )
        {
            const V3 walk_p = ray_pos( org_get( rp ), rd, offs );
            double a = F3_INF;
            int sd = 0;
            #pragma unroll 1
            for( int k = 0; k < 2; k++ )
            {
                if( swapped == ( k != 0 ) )
                {
                    a = operand_hit< L, SPLIT >( sc, k ? c1 : c0, walk_p, rd, want_nor, &n1, cnt );
                    if( a < F3_INF ) sd = operand_side< L >( sc, k ? c0 : c1, ray_pos( walk_p, rd, a ), cnt );
                }
            }
            cnt->cost( ACN_F_PAIR_STEP );
            if( a >= F3_INF ) return F3_INF;
            if( sd == want ) { if( want_nor ) *nor = n1; return offs + a; }
            offs += a + 2 * F3_EPS;
            if( !( offs < F3_INF ) ) return F3_INF;
            swapped = !swapped;
        }
    }
    else
    {
        V3 n1 = mk( 0, 0, 0 ), n2 = mk( 0, 0, 0 );
        double a1 = operand_hit< L, SPLIT >( sc, c0, rp, rd, want_nor, &n1, cnt );
        double a2 = operand_hit< L, SPLIT >( sc, c1, rp, rd, want_nor, &n2, cnt );
        cnt->cost( ACN_F_PAIR_STEP );
        if( a1 < a2 && operand_side< L >( sc, c1, ray_pos( org_get( rp ), rd, a1 ), cnt ) == want ) { *nor = n1; return a1; }
        if( a2 >= F3_INF ) return F3_INF;
        cnt->cost( ACN_F_PAIR_STEP );
        if( operand_side< L >( sc, c0, ray_pos( org_get( rp ), rd, a2 ), cnt ) == want ) { *nor = n2; return a2; }
        double offs = a2;
        bool swapped = false;
        for( ;; )
        {
            V3 walk_p = ray_pos( org_get( rp ), rd, offs );
            double a = operand_hit< L, SPLIT >( sc, swapped ? c1 : c0, walk_p, rd, want_nor, &n1, cnt );
            cnt->cost( ACN_F_PAIR_STEP );
            if( a >= F3_INF ) return F3_INF;
            int sd = operand_side< L >( sc, swapped ? c0 : c1, ray_pos( walk_p, rd, a ), cnt );
            if( sd == want ) { *nor = n1; return offs + a; }
            offs += a + 2 * F3_EPS;
            if( !( offs < F3_INF ) ) return F3_INF;
            swapped = !swapped;
        }
    }
}

template< class SR, class NP, class CT > DEV int leaf_pair_side( SR sc, NP n, V3 pos, CT* cnt ) { return pair_side< 1 >( sc, n, pos, cnt ); }
template< class SR, class NP, class CT > DEV double leaf_pair_hit( SR sc, NP n, V3 rp, V3 rd, bool want_nor, V3* nor, CT* cnt ) { return pair_hit< 1 >( sc, n, rp, rd, want_nor, nor, cnt ); }
/* the same for a pair node that is the same in every lane */
template< class SR, class NP, class CT > DEV double leaf_pair_hit_uniform( SR sc, NP n, V3 rp, V3 rd, bool want_nor, V3* nor, CT* cnt ) { return pair_hit< 1, true >( sc, n, rp, rd, want_nor, nor, cnt ); }

/* ------------------------------------------------------------------------------------------------------------------ */
/* CSG machines.  obj_side and obj_ray_hit recurse through pair / neg / scale nodes in the reference; here they are
 * per-lane state machines.  The innermost composite's frame lives in registers; enclosing frames sit on a per-lane
 * stack.  Stack traffic is scratch traffic, and at full occupancy scratch does not fit L2 -- round-1 profiles showed
 * the machine kernels moving ~390 GB per 1080p frame through HBM at 3.7-4.7 TB/s, i.e. bound by their own stacks.
 * Frames are therefore as small as the algorithm allows:
 *   side frame  4 B   node << 2 | pc.  The position is the same for every frame unless a scale wrapper intervenes;
 *                     a scale wrapper parks the outer position on a small auxiliary stack.
 *   hit frame  12 B   node / pc / swapped / inherit packed into one word + ONE double: a1 while the second child is
 *                     evaluated (pc 2), the walk offset afterwards (pc 3), d_factor for a scale wrapper
 *             +24 B   n1, only when normals are wanted (transition hits; never for occlusion tests).
 *   The origin a frame received is not stored: a child evaluated at its parent's own origin inherits it (the common
 *   case); only children entered from the alternating walk or below a scale wrapper park the parent's origin (and the
 *   wrapper its direction) on the auxiliary stack. */
#define ACN_PACK_SIDE( node, pc ) ( ( ( uint32_t )( node ) << 2 ) | ( uint32_t )( pc ) )

template< class SR, class CT >
DEV_SIDE int obj_side_dev( SR sc, int root, V3 pos, CT* cnt )
{
    uint32_t st[ ACN_CSG_MAX_DEPTH ];
    V3 aux[ ACN_CSG_MAX_DEPTH ];
    const bool lds = sc.lds_stack != ACN_NO_LDS_STACK;
    LdsU32P ls = ( LdsU32P )( ( char ACN_LDS* )acn_lds_raw + sc.lds_stack ) + ACN_LDS_DEPTH * ACN_LDS_LANES * 9 + threadIdx.x;
    uint32_t cur = 0;
    int depth = 0, na = 0;
    int node = root;
    int r = 1;
    ACN_LAP( PH_M_FRAME );
    for( ;; )
    {
        /* EVAL( node, pos ) */
        auto n = &sc.nodes[ node ];
        cnt->inc( CNT_SIDE );
        bool have = true;
        int type = n->type;
        if( node_has_env( n ) && env_side( n, pos ) == 1 )
        {
            r = 1;
        }
        else if( type <= ACN_DISTANCE )
        {
            switch( type )
            {
                case ACN_PLANE:    cnt->cost( ACN_F_SIDE_PLANE ); r = v_sub_mlv( pos, ld3( n->pos ), ld3( n->rax + 6 ) ) > 0 ? 1 : -1; break;   /* gmath.h:52-55 */
                case ACN_SPHERE:   r = sphere_observer_side( ld3( n->pos ), n->prm[ 0 ], pos ); break;
                case ACN_SQUAROID: r = squaroid_side( n, pos ); break;
                default:           r = distance_side( n, pos ); cnt->inc( CNT_SDF_EVAL ); break;
            }
        }
        else if( n->flags & ACN_GFLAG_LEAF_PAIR )
        {
            r = leaf_pair_side( sc, n, pos, cnt );
        }
        else if( n->flags & ACN_GFLAG_PAIR2 )
        {
            r = pair_side< 2 >( sc, n, pos, cnt );
        }
        else if( depth >= ACN_CSG_MAX_DEPTH )
        {
            atomicOr( sc.flags, ACN_FLAG_STACK_OVERFLOW );
            r = 1;
        }
        else
        {
            if( depth > 0 )
            {
                if( lds && depth <= ACN_LDS_DEPTH ) ls[ ( depth - 1 ) * ACN_LDS_LANES ] = cur; else st[ depth - 1 ] = cur;
            }
            depth++;
            cur = ACN_PACK_SIDE( node, 1 );
            if( type == ACN_SCALE )   /* objects.c:1439-1443 */
            {
                cnt->cost( ACN_F_SIDE_SCALE );
                aux[ na++ ] = pos;    /* na <= depth <= ACN_CSG_MAX_DEPTH */
                M3 rax = node_rax( n );
                V3 p = m_mlv( rax, v_sub( pos, ld3( n->pos ) ) );
                pos = v_mld( p, mk( n->prm[ 0 ], n->prm[ 1 ], n->prm[ 2 ] ) );
            }
            node = n->child0;
            have = false;
        }
        /* RETURN( r ) into the enclosing composites */
        while( have )
        {
            if( depth == 0 ) { ACN_LAP( PH_M_SIDE ); return r; }
            int cnode = ( int )( cur >> 2 );
            auto fn = &sc.nodes[ cnode ];
            int ftype = fn->type;
            bool done = true;
            if( ftype == ACN_NEG ) r = -r;                                   /* objects.c:1341-1344 */
            else if( ftype == ACN_SCALE ) pos = aux[ --na ];
            else
            {
                int want = ( ftype == ACN_PAIR_INSIDE ) ? -1 : 1;           /* objects.c:1096-1099, 1253-1256 */
                if( ( cur & 3u ) == 1u )
                {
                    if( r != want ) r = -want;
                    else { cur = ACN_PACK_SIDE( cnode, 2 ); node = fn->child1; done = false; have = false; }
                }
                else
                {
                    r = ( r == want ) ? want : -want;
                }
            }
            if( done )
            {
                depth--;
                if( depth > 0 ) cur = ( lds && depth <= ACN_LDS_DEPTH ) ? ls[ ( depth - 1 ) * ACN_LDS_LANES ] : st[ depth - 1 ];
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* hit machine: obj_ray_hit (objects.c:261-284) with pair / neg / scale recursion unrolled into frames */
#define ACN_HW_PC( w )       ( ( w ) & 3u )
#define ACN_HW_SWAPPED( w )  ( ( ( w ) >> 2 ) & 1u )
#define ACN_HW_INHERIT( w )  ( ( ( w ) >> 3 ) & 1u )
#define ACN_HW_NODE( w )     ( ( int )( ( w ) >> 4 ) )
#define ACN_HW_PACK( node, pc, swapped, inherit ) ( ( ( uint32_t )( node ) << 4 ) | ( ( uint32_t )( inherit ) << 3 ) | ( ( uint32_t )( swapped ) << 2 ) | ( uint32_t )( pc ) )


template< class SR, class CT >
DEV_HIT double obj_ray_hit_dev( SR sc, int root, V3 rp, V3 rd, bool want_nor, V3* out_nor, CT* cnt )
{
    uint32_t st_w[ ACN_CSG_MAX_DEPTH ];
    double   st_a[ ACN_CSG_MAX_DEPTH ];
    V3       st_n[ ACN_CSG_MAX_DEPTH ];     /* touched only when want_nor */
    V3       aux[ 2 * ACN_CSG_MAX_DEPTH ];  /* parked origins / directions */
    const bool lds = sc.lds_stack != ACN_NO_LDS_STACK;
    LdsF64P la = ( LdsF64P )( ( char ACN_LDS* )acn_lds_raw + sc.lds_stack ) + threadIdx.x;
    LdsF64P ln = la + ACN_LDS_DEPTH * ACN_LDS_LANES;   /* x, y, z planes of the parked normals */
    LdsU32P lw = ( LdsU32P )( ( char ACN_LDS* )acn_lds_raw + sc.lds_stack ) + ACN_LDS_DEPTH * ACN_LDS_LANES * 8 + threadIdx.x;
    uint32_t cur_w = 0;
    double cur_a = 0;                       /* pair: a1 (pc 2), walk offset (pc 3) | scale: d_factor */
    V3 cur_n1 = mk( 0, 0, 0 );
    V3 cur_rp = rp;                         /* origin of the ray the current frame received */
    bool rp_derived = false;                /* rp differs from cur_rp (set by the walk and by scale wrappers) */
    int depth = 0, na = 0;
    int node = root;
    double ret_a = F3_INF;
    V3 ret_n = mk( 0, 0, 0 );
    for( ;; )
    {
        /* ---- EVAL( node, rp, rd ) ---- */
        ACN_LAP( PH_M_FRAME );
        auto n = &sc.nodes[ node ];
        cnt->inc( CNT_OBJ_HIT );
        bool have = true;
        int type = n->type;
        if( node_has_env( n ) && !env_ray_hits( n, rp, rd ) )
        {
            ret_a = F3_INF;
        }
        else if( type <= ACN_DISTANCE )
        {
            switch( type )
            {
                case ACN_PLANE:    ret_a = plane_ray_hit( ld3( n->pos ), ld3( n->rax + 6 ), rp, rd, want_nor, &ret_n ); break;
                case ACN_SPHERE:   ret_a = sphere_ray_hit( ld3( n->pos ), n->prm[ 0 ], rp, rd, want_nor, &ret_n ); break;
                case ACN_SQUAROID: ret_a = squaroid_ray_hit( n, rp, rd, want_nor, &ret_n ); break;
                default:           { V3 dn = mk( 0, 0, 0 ); ret_a = distance_ray_hit( n, rp, rd, want_nor, &dn, cnt ); if( want_nor && ret_a < F3_INF ) ret_n = dn; } break;   /* (a real call: only dn's address escapes, ret_n stays in registers) */
            }
            if( want_nor && ret_a < F3_INF && n->surface_roughness > 0 ) ret_n = roughness_normal( n, ret_n, ray_pos( rp, rd, ret_a ) );
            ACN_LAP( PH_M_LEAF );
        }
        else if( n->flags & ( ACN_GFLAG_LEAF_PAIR | ACN_GFLAG_PAIR2 ) )
        {
            ret_a = ( n->flags & ACN_GFLAG_PAIR2 ) ? pair_hit< 2 >( sc, n, rp, rd, want_nor, &ret_n, cnt )
                                                   : pair_hit< 1 >( sc, n, rp, rd, want_nor, &ret_n, cnt );
            if( want_nor && ret_a < F3_INF && n->surface_roughness > 0 ) ret_n = roughness_normal( n, ret_n, ray_pos( rp, rd, ret_a ) );
            ACN_LAP( PH_M_PAIR );
        }
        else if( depth >= ACN_CSG_MAX_DEPTH )
        {
            atomicOr( sc.flags, ACN_FLAG_STACK_OVERFLOW );
            ret_a = F3_INF;
        }
        else
        {
            if( depth > 0 )
            {
                if( lds && depth <= ACN_LDS_DEPTH )
                {
                    const int o = ( depth - 1 ) * ACN_LDS_LANES;
                    lw[ o ] = cur_w; la[ o ] = cur_a;
                    if( want_nor ) { ln[ o ] = cur_n1.x; ln[ o + ACN_LDS_DEPTH * ACN_LDS_LANES ] = cur_n1.y; ln[ o + 2 * ACN_LDS_DEPTH * ACN_LDS_LANES ] = cur_n1.z; }
                }
                else
                {
                    st_w[ depth - 1 ] = cur_w; st_a[ depth - 1 ] = cur_a;
                    if( want_nor ) st_n[ depth - 1 ] = cur_n1;
                }
            }
            depth++;
            if( rp_derived ) aux[ na++ ] = cur_rp;      /* na <= 2 * depth */
            cur_w = ACN_HW_PACK( node, 1, 0, rp_derived ? 0 : 1 );
            cur_rp = rp;
            rp_derived = false;
            if( type == ACN_SCALE )   /* objects.c:1418-1428 */
            {
                cnt->cost( ACN_F_SCALE_WRAP );
                M3 rax = node_rax( n );
                V3 inv_scale = mk( n->prm[ 0 ], n->prm[ 1 ], n->prm[ 2 ] );
                V3 p2 = v_mld( m_mlv( rax, v_sub( rp, ld3( n->pos ) ) ), inv_scale );
                V3 d2 = v_mld( m_mlv( rax, rd ), inv_scale );
                double d_length = acn_sqrt( v_sqr( d2 ) );
                double d_factor = ( d_length > 0 ) ? ( 1.0 / d_length ) : 0;
                d2 = v_mlf( d2, d_factor );
                aux[ na++ ] = rd; cur_a = d_factor;
                rp = p2; rd = d2;
                rp_derived = true;
            }
            node = n->child0;
            have = false;
        }

        /* ---- RETURN( ret_a, ret_n ) into the enclosing composites ---- */
        while( have )
        {
            if( depth == 0 )
            {
                /* like the reference, the caller's normal is only written on a hit (obj_ray_exit relies on it) */
                if( want_nor && ret_a < F3_INF ) *out_nor = ret_n;
                ACN_LAP( PH_M_FRAME );
                return ret_a;
            }
            int cnode = ACN_HW_NODE( cur_w );
            auto fn = &sc.nodes[ cnode ];
            int ftype = fn->type;
            bool done = true;
            if( ftype == ACN_NEG )   /* objects.c:1329-1339 */
            {
                if( ret_a < F3_INF ) ret_n = v_neg( ret_n );
            }
            else if( ftype == ACN_SCALE )   /* objects.c:1430-1437 */
            {
                double a1 = ret_a + F3_EPS;
                rd = aux[ --na ];
                if( a1 < F3_INF )
                {
                    if( want_nor )
                    {
                        V3 n1 = v_mld( ret_n, mk( fn->prm[ 0 ], fn->prm[ 1 ], fn->prm[ 2 ] ) );
                        ret_n = v_of_length( m_tmlv( node_rax( fn ), n1 ), 1.0 );
                    }
                    ret_a = a1 * cur_a - F3_EPS;
                }
                else
                {
                    ret_a = F3_INF;
                }
            }
            else   /* pair: objects.c:1052-1094 / 1209-1251 */
            {
                int want = ( ftype == ACN_PAIR_INSIDE ) ? -1 : 1;
                uint32_t pc = ACN_HW_PC( cur_w );
                if( pc == 1 )
                {
                    cur_a = ret_a; cur_n1 = ret_n;
                    cur_w = ACN_HW_PACK( cnode, 2, 0, ACN_HW_INHERIT( cur_w ) );
                    node = fn->child1; rp = cur_rp; rp_derived = false;
                    done = false; have = false;
                }
                else if( pc == 2 )
                {
                    cnt->cost( 2 * ACN_F_PAIR_STEP );
                    double a1 = cur_a, a2 = ret_a;
                    if( a1 < a2 && obj_side_dev( sc, fn->child1, ray_pos( cur_rp, rd, a1 ), cnt ) == want )
                    {
                        ret_a = a1; ret_n = cur_n1;
                    }
                    else if( a2 >= F3_INF )
                    {
                        ret_a = F3_INF;
                    }
                    else if( obj_side_dev( sc, fn->child0, ray_pos( cur_rp, rd, a2 ), cnt ) == want )
                    {
                        /* ret_a = a2, ret_n = n2 already */
                    }
                    else
                    {
                        cur_a = a2;   /* the walk offset */
                        cur_w = ACN_HW_PACK( cnode, 3, 0, ACN_HW_INHERIT( cur_w ) );
                        node = fn->child0; rp = ray_pos( cur_rp, rd, cur_a ); rp_derived = true;
                        done = false; have = false;
                    }
                }
                else
                {
                    cnt->cost( ACN_F_PAIR_STEP );
                    double a = ret_a;
                    if( a >= F3_INF )
                    {
                        ret_a = F3_INF;
                    }
                    else
                    {
                        V3 walk_p = ray_pos( cur_rp, rd, cur_a );
                        uint32_t swapped = ACN_HW_SWAPPED( cur_w );
                        int obj2 = swapped ? fn->child0 : fn->child1;
                        if( obj_side_dev( sc, obj2, ray_pos( walk_p, rd, a ), cnt ) == want )
                        {
                            ret_a = cur_a + a;   /* ret_n = n1 of the last child call */
                        }
                        else
                        {
                            cur_a += a + 2 * F3_EPS;
                            if( !( cur_a < F3_INF ) )
                            {
                                ret_a = F3_INF;
                            }
                            else
                            {
                                swapped ^= 1u;
                                cur_w = ACN_HW_PACK( cnode, 3, swapped, ACN_HW_INHERIT( cur_w ) );
                                node = swapped ? fn->child1 : fn->child0;
                                rp = ray_pos( cur_rp, rd, cur_a ); rp_derived = true;
                                done = false; have = false;
                            }
                        }
                    }
                }
            }
            if( done )
            {
                /* POST of the composite itself (objects.c:266-282), then hand its result to its parent */
                rp = cur_rp;
                if( want_nor && ret_a < F3_INF && fn->surface_roughness > 0 ) ret_n = roughness_normal( fn, ret_n, ray_pos( rp, rd, ret_a ) );
                if( !ACN_HW_INHERIT( cur_w ) ) cur_rp = aux[ --na ];
                rp_derived = false;
                depth--;
                if( depth > 0 )
                {
                    if( lds && depth <= ACN_LDS_DEPTH )
                    {
                        const int o = ( depth - 1 ) * ACN_LDS_LANES;
                        cur_w = lw[ o ]; cur_a = la[ o ];
                        if( want_nor ) cur_n1 = mk( ln[ o ], ln[ o + ACN_LDS_DEPTH * ACN_LDS_LANES ], ln[ o + 2 * ACN_LDS_DEPTH * ACN_LDS_LANES ] );
                    }
                    else
                    {
                        cur_w = st_w[ depth - 1 ]; cur_a = st_a[ depth - 1 ];
                        if( want_nor ) cur_n1 = st_n[ depth - 1 ];
                    }
                }
            }
        }
    }
}

#include "acn_unimachine.h"

/* ------------------------------------------------------------------------------------------------------------------ */
/* compounds: compound.c:215-299 */

/* Generic (per-lane indices) closest hit inside compound `cmp`, recursing into nested compounds with an explicit
 * stack. limit: stop as soon as a hit <= limit is found (any-hit for occlusion: "compound_s_ray_hit( matter ) > a"
 * is false iff some element hits at <= a). */
template< class SR, class CT >
DEVN double compound_ray_hit_dev( SR sc, int cmp, V3 rp, V3 rd, bool want_nor, V3* p_nor, int* hit_obj,
                                  double limit, CT* cnt )
{
    int st_i[ ACN_CMP_MAX_DEPTH ], st_end[ ACN_CMP_MAX_DEPTH ];
    int sp = 0;
    double min_a = F3_INF;
    const int order = limit >= 0 ? ( int )sc.n_elems : 0;   /* any-hit queries walk the cost-ordered copy of elems */
    {
        auto o = &sc.nodes[ cmp ];
        if( node_has_env( o ) && !env_ray_hits( o, rp, rd ) ) return F3_INF;
        st_i[ 0 ] = o->child0 + order; st_end[ 0 ] = o->child0 + order + o->child1; sp = 1;
    }
    while( sp > 0 )
    {
        if( st_i[ sp - 1 ] >= st_end[ sp - 1 ] ) { sp--; continue; }
        int element = sc.elems[ st_i[ sp - 1 ]++ ];
        auto e = &sc.nodes[ element ];
        if( e->type == ACN_COMPOUND )
        {
            if( node_has_env( e ) && !env_ray_hits( e, rp, rd ) ) continue;
            if( sp >= ACN_CMP_MAX_DEPTH ) { atomicOr( sc.flags, ACN_FLAG_STACK_OVERFLOW ); continue; }
            st_i[ sp ] = e->child0 + order; st_end[ sp ] = e->child0 + order + e->child1; sp++;
            continue;
        }
        V3 nor;
        double a;
        if( e->type >= ACN_PLANE && e->type <= ACN_SQUAROID )   /* a simple leaf (every element of many_spheres): no need for the machine */
        {
            cnt->inc( CNT_OBJ_HIT );
            a = ( node_has_env( e ) && !env_ray_hits( e, rp, rd ) ) ? F3_INF : simple_leaf_hit( e, rp, rd, want_nor, &nor );
        }
        else a = obj_ray_hit_dev( sc, element, rp, rd, want_nor, &nor, cnt );
        if( a < min_a )
        {
            min_a = a;
            if( want_nor ) *p_nor = nor;
            if( hit_obj ) *hit_obj = element;
            if( a <= limit ) return a;
        }
    }
    return min_a;
}

/* true only if every point of the sphere ( env_pos, r ) on the ray lies beyond ray parameter `bound` by a margin that covers
 * the F3_EPS the hit routines subtract and all rounding: the origin is outside the sphere, the sphere lies ahead, and its entry
 * point -s - sqrt( s*s - q ) exceeds bound + margin.  Only called for envelopes the ray hits ( s*s >= q ).  Conservative: a `false`
 * costs a visit, a wrong `true` would cost a hit. */
DEV bool env_behind( V3 env_pos, double r, V3 rp, V3 rd, double bound )
{
    V3 p = v_sub( rp, env_pos );
    double s = v_mlv( p, rd );
    double q = v_sqr( p ) - ( r * r );
    double s2 = s * s;
    double u = ( -s - bound ) - ( 1E-5 + 1E-9 * ( fabs( s ) + fabs( bound ) ) );
    return q > 0 && u > 0 && u * u > ( s2 - q ) + 1E-9 * ( s2 + fabs( q ) );
}

/* Simple compounds: a root element that is a compound whose whole subtree consists of nested compounds and simple
 * leaves (many_spheres: 32 768 spheres under five levels of enveloped compounds).  The upload step lays its subtree
 * out in pre-order as SCEntry records with a skip link per entry (elems[ offset ], elems[ offset + 1 ]: first entry and
 * entry count); compound_s_ray_hit (compound.c:215-243) then is a stackless loop: an enveloped compound that the
 * ray misses is skipped by its link, everything else advances by one.  Same element order as the recursion, so ties
 * between equal distances resolve identically.  No stack, no machine: k_shade runs it in line. */
#define ACN_GFLAG_SIMPLE_COMPOUND 0x200u   /* device-only bit of GNode.flags */

template< bool NOR, class SC, class CT >
DEV double simple_compound_hit( const SC& sc, int cmp, V3 rp, V3 rd, V3* p_nor, int* hit_obj, double limit, CT* cnt )
{
    int off = sc.elems[ sc.prune_base + ( uint32_t )cmp ];
    int i = sc.elems[ off ];
    const int count = sc.elems[ off + 1 ];
    double min_a = F3_INF;
    if( count <= 0 ) return min_a;
    /* The result is the minimum over the leaves, the FIRST leaf in the reference's order winning a tie -- it does not depend on the
     * order of the walk as long as that rule is kept.  The upload step lays big subtrees out a second time with the children of
     * every compound in reverse order; a ray that runs against the order of the first table (order_dir: the direction along which
     * later children lie, summed over the subtree's compounds) walks the second, meets its near leaves sooner and culls more.  In
     * the reversed table the leaves come in exactly the reverse order, so "first wins" becomes "last visited wins": `<=`. */
    bool rev = false;
    if( ACN_SC_CULL && ACN_SC_TWO_ORDERS && !CT::counting )
    {
        const int first_rev = sc.elems[ off + 2 ];
        if( first_rev >= 0 )
        {
            const double* g = sc.sc_spheres + 4 * ( size_t )sc.elems[ off + 3 ];
            rev = v_mlv( rd, mk( g[ 0 ], g[ 1 ], g[ 2 ] ) ) < 0;
            if( rev ) i = first_rev;
        }
    }
    const int end = i + count;
    /* The walk does not depend on what the leaves return (an envelope test is a predicate of the ray alone; only the occlusion
     * form leaves early), and k_shade<64> on many_spheres is short of VALU issue slots, not of memory (PMC, profiles/r04/NOTES.md
     * section 6: 72 % of the issue slots busy, waves waiting 20 % of their time): one lane in fifty stands on a leaf whose envelope
     * its ray hits, so three wave iterations in four ran the whole sphere routine for one or two lanes.  DEFER: a lane that finds
     * such a leaf parks it (one slot) and walks on; only when a lane of the wave finds a SECOND one do all lanes evaluate what they
     * have parked, behind a wave-uniform branch.  Leaves are evaluated in walk order per lane, so min_a, the strict `<` tie rule
     * and the first leaf within `limit` are the ones of the plain loop; the occlusion form may walk a few entries further before
     * it learns that it could have left.  A hit sphere's normal is computed once, behind the loop, from the same expression.
     * The counting kernels keep the plain loop: their event counts are compared with the oracle's. */
    constexpr bool DEFER = ACN_SC_DEFER && !CT::counting;
    /* CULL: an entry whose envelope provably contains everything below it (ACN_SC_BOUNDING) and lies wholly behind the best hit so
     * far cannot change the result -- every hit in it is farther, and a farther hit never replaces a nearer one -- so it is treated
     * like a missed envelope.  The reference visits it (compound.c:215-243 has no such test); the counting kernels therefore do,
     * too.  many_spheres, every 16th pixel: see profiles/r04/NOTES.md section 6. */
    constexpr bool CULL = ACN_SC_CULL && !CT::counting;
    uint32_t pend_flags = 0; int pend_node = -1, pend_sph = 0;   /* the parked leaf ( flags != 0: one is parked ) */
    int best_sph = -1; uint32_t best_flags = 0;                   /* DEFER && NOR: the sphere min_a belongs to */
    auto leaf = [ & ]( int node, int sph, uint32_t flags ) -> bool   /* true: the occlusion form is done */
    {
        V3 nor = mk( 0, 0, 0 );
        double a;
        bool table = false;
#if ACN_SC_SPHERE_TABLE
        if( flags & ACN_SC_SPHERE )   /* the sphere itself from the compact table beside the entries (32 B, L2-resident) instead of its 192-byte node */
        {
            const double* g = sc.sc_spheres + 4 * ( size_t )sph;
            a = sphere_ray_hit( mk( g[ 0 ], g[ 1 ], g[ 2 ] ), g[ 3 ], rp, rd, NOR && !DEFER, &nor );
            if( NOR && !DEFER && a < F3_INF && ( flags & ACN_SC_ROUGH ) ) nor = roughness_normal( &sc.nodes[ node ], nor, ray_pos( rp, rd, a ) );
            table = true;
        }
        else
#endif
        a = simple_leaf_hit( &sc.nodes[ node ], rp, rd, NOR, &nor );
        if( rev ? ( a <= min_a && a < F3_INF ) : ( a < min_a ) )
        {
            min_a = a;
            if( NOR ) *p_nor = nor;
            if( NOR && DEFER ) { best_sph = table ? sph : -1; best_flags = flags; }
            *hit_obj = node;
            if( a <= limit ) return true;
        }
        return false;
    };
    SCEntry e = sc.sc_table[ i ];
    while( i < end )
    {
#ifdef ACN_SC_PREFETCH
        const SCEntry ahead = sc.sc_table[ i + 1 < end ? i + 1 : i ];
#endif
        bool miss = ( e.flags & ACN_NODE_HAS_ENVELOPE ) && !env_ray_hits_raw( ld3( e.env_pos ), e.env_radius, rp, rd, cnt );
        if( CULL && !miss && ( e.flags & ACN_SC_BOUNDING ) )
        {
            const double bound = NOR ? min_a : limit;   /* the occlusion form leaves at the first hit within `limit`: nothing beyond it matters */
            if( bound < F3_INF ) miss = env_behind( ld3( e.env_pos ), e.env_radius, rp, rd, bound );
        }
        const bool is_leaf = e.type != ACN_COMPOUND;
        const int next = ( !is_leaf && miss ) ? e.skip : i + 1;
#if ACN_SC_EARLY_NEXT
        /* the next entry requested BEFORE the leaf is evaluated.  Measured (profiles/r04/ab_c3_table_s20.txt): -4 % alone, +5 % on top
         * of the sphere table: off */
        SCEntry e_next = e;
        if( next < end ) e_next = sc.sc_table[ next ];
#endif
        if( is_leaf ) cnt->inc( CNT_OBJ_HIT );
        const bool want = is_leaf && !miss;
        if( DEFER )
        {
            if( __ballot( want && pend_flags != 0 ) != 0 )
            {
                if( pend_flags != 0 )
                {
                    if( leaf( pend_node, pend_sph, pend_flags ) ) return min_a;
                    pend_flags = 0;
                }
            }
            if( want ) { pend_node = e.node; pend_sph = e.skip; pend_flags = e.flags | 0x80000000u; }
        }
        else if( want )
        {
            if( leaf( e.node, e.skip, e.flags ) ) return min_a;
        }
#if ACN_SC_EARLY_NEXT
        e = e_next;
#elif defined( ACN_SC_PREFETCH )
        if( next == i + 1 ) e = ahead;
        else if( next < end ) e = sc.sc_table[ next ];
#else
        if( next < end ) e = sc.sc_table[ next ];
#endif
        i = next;
    }
    if( DEFER )
    {
        if( pend_flags != 0 && leaf( pend_node, pend_sph, pend_flags ) ) return min_a;
#if ACN_SC_SPHERE_TABLE
        if( NOR && best_sph >= 0 )   /* sphere_ray_hit's normal (gmath.h:64-83), once, for the hit that won */
        {
            const double* g = sc.sc_spheres + 4 * ( size_t )best_sph;
            V3 nor = v_of_length( v_sub( ray_pos( rp, rd, min_a ), mk( g[ 0 ], g[ 1 ], g[ 2 ] ) ), 1.0 );
            if( best_flags & ACN_SC_ROUGH ) nor = roughness_normal( &sc.nodes[ *hit_obj ], nor, ray_pos( rp, rd, min_a ) );
            *p_nor = nor;
        }
#endif
    }
    return min_a;
}

/* Conservative pruning before a ray is handed to the CSG machine: surely_outside( n ) == true guarantees that the
 * whole ray lies outside n's bounding envelopes, in which case the reference returns f3_inf for n AND obj_side( n )
 * is +1 at every point of the ray:
 *   - n has an envelope and the ray misses it (objects.c:264, :368);
 *   - pair_outside: both children surely outside -> a1 = a2 = f3_inf -> f3_inf (objects.c:1224), side 1+1 == 2;
 *   - pair_inside: one child surely outside -> it returns f3_inf and classifies every point of the ray as outside, so
 *     neither the direct candidates nor the alternating walk can accept a hit (objects.c:1057-1092), side != -2.
 * Complements and scale wrappers are never pruned.  Expanded D levels deep. */
template< int D, class NP >
DEV bool surely_outside_n( NP nodes, int node, V3 rp, V3 rd )
{
    auto n = &nodes[ node ];
    /* levels of this node worth reading (upload step, actinon_hip.hip): 0 nothing to test here or below, 1 only the node's own
     * envelope, ...: the same tests as a blind descent, without the operand reads that lead to no test */
    const uint32_t levels = ( n->flags >> ACN_GFLAG_PRUNE_LEVELS_SHIFT ) & 7u;
    if( levels == 0 ) return false;
    if( node_has_env( n ) && !env_ray_hits_( n, rp, rd, ACN_NO_CNT ) ) return true;
    if constexpr( D > 0 )
    {
        if( levels < 2 ) return false;
        int type = n->type;
        if( type == ACN_PAIR_OUTSIDE ) return surely_outside_n< D - 1 >( nodes, n->child0, rp, rd ) && surely_outside_n< D - 1 >( nodes, n->child1, rp, rd );
        if( type == ACN_PAIR_INSIDE )  return surely_outside_n< D - 1 >( nodes, n->child0, rp, rd ) || surely_outside_n< D - 1 >( nodes, n->child1, rp, rd );
    }
    return false;
}
#ifndef ACN_PRUNE_CALL
#define ACN_PRUNE_CALL 0
#endif
/* ACN_PRUNE_CALL: ONE copy of the descent per kernel behind a real call instead of one expansion per call site */
template< int D, class NP >
__device__ __attribute__( ( noinline ) ) bool surely_outside_call( NP nodes, int node, V3 rp, V3 rd ) { return surely_outside_n< D >( nodes, node, rp, rd ); }
template< int D, class SC >
DEV bool surely_outside( const SC& sc, int node, V3 rp, V3 rd )
{
#if ACN_PRUNE_CALL
    return surely_outside_call< D >( sc.nodes, node, rp, rd );
#else
    return surely_outside_n< D >( sc.nodes, node, rp, rd );
#endif
}
#ifndef ACN_PRUNE_DEPTH
#define ACN_PRUNE_DEPTH 3
#endif

/* ------------------------------------------------------------------------------------------------------------------ */
/* Interval pruning of big CSG objects.  For a root element with many nodes the upload step compiles a small postfix
 * program (actinon_hip.hip: build_prune_programs) that computes a conservative parameter interval [ lo, hi ] of the
 * ray: every t at which rp + t * rd could be classified "inside the object" by obj_side, or be reported as a hit,
 * lies in it.  An empty interval means the reference returns f3_inf for the object and +1 for obj_side at every point
 * it would look at (the argument of surely_outside, made sharper), so the object is skipped with results unchanged:
 *   plane                     half line (linear inequality along the ray)
 *   sphere, envelope          chord of the ball
 *   squaroid                  roots of the quadric restricted to the ray when that restriction is convex (A > 0)
 *   NEG( plane )              the complementary half line; any other complement: no statement (whole ray)
 *   pair_inside               intersection of the children's intervals       pair_outside: hull of their union
 *   scale, distance, subtrees beyond the program's budget: no statement beyond the node's envelope
 * Conservative by construction: surfaces are widened by 1e-9 relative, every end point by ACN_IV_W = 1e-4 ray units
 * (hits are reported f3_eps = 1e-6 before the surface and the alternating walk steps 2 * f3_eps at a time,
 * objects.c:1080-1090) plus 1e-8 relative for root cancellation.  Directions are unit vectors at the call sites (the
 * program never descends through a scale wrapper).  Program and node index are wave-uniform: every load is scalar and
 * every branch of the interpreter loop is a scalar branch; the interval stack is six register pairs that shift. */
struct Iv { double lo, hi; };
#define ACN_IV_W 1.0e-4
DEV Iv iv_all() { Iv r; r.lo = -ACN_IV_W; r.hi = F3_INF; return r; }
DEV Iv iv_none() { Iv r; r.lo = 1.0; r.hi = 0.0; return r; }
DEV bool iv_empty( Iv a ) { return !( a.lo <= a.hi ); }
DEV Iv iv_and( Iv a, Iv b ) { Iv r; r.lo = f_max( a.lo, b.lo ); r.hi = f_min( a.hi, b.hi ); return r; }
DEV Iv iv_or( Iv a, Iv b )
{
    if( iv_empty( a ) ) return b;
    if( iv_empty( b ) ) return a;
    Iv r; r.lo = f_min( a.lo, b.lo ); r.hi = f_max( a.hi, b.hi );
    return r;
}
DEV double iv_widen_lo( double t ) { return t - ( ACN_IV_W + 1.0e-8 * f_abs( t ) ); }
DEV double iv_widen_hi( double t ) { return t + ( ACN_IV_W + 1.0e-8 * f_abs( t ) ); }

/* { t : ( rp + t rd - pos ) * nor <= 0 }, or >= 0 for the complement */
DEV Iv iv_halfspace( V3 pos, V3 nor, V3 rp, V3 rd, bool complement )
{
    double s0 = v_sub_mlv( rp, pos, nor );
    double s1 = v_mlv( nor, rd );
    if( complement ) { s0 = -s0; s1 = -s1; }
    s0 -= 1.0e-9 * ( f_abs( s0 ) + 1.0 );
    Iv r = iv_all();
    if( s1 > 0 )      r.hi = iv_widen_hi( -s0 / s1 );
    else if( s1 < 0 ) r.lo = f_max( r.lo, iv_widen_lo( -s0 / s1 ) );
    else if( s0 > 0 ) r = iv_none();
    return r;
}

DEV Iv iv_ball( V3 c, double radius, V3 rp, V3 rd )
{
    V3 p = v_sub( rp, c );
    double s = v_mlv( p, rd );
    double rr = radius * radius, pp = v_sqr( p );
    double q = pp - ( rr + 1.0e-9 * ( rr + pp ) + 1.0e-12 );
    double disc = s * s - q;
    if( !( disc >= 0 ) ) return iv_none();
    double sq = acn_sqrt( disc );
    Iv r; r.lo = iv_widen_lo( -s - sq ); r.hi = iv_widen_hi( -s + sq );
    return iv_and( r, iv_all() );
}

template< class NP > DEV Iv iv_squaroid( NP o, V3 rp, V3 rd )
{
    M3 rax = node_rax( o );
    V3 p = m_mlv( rax, v_sub( rp, ld3( o->pos ) ) );
    V3 d = m_mlv( rax, rd );
    double a = o->prm[ 0 ], b = o->prm[ 1 ], c = o->prm[ 2 ], r = o->prm[ 3 ];
    double A = a * d.x * d.x + b * d.y * d.y + c * d.z * d.z;
    double B = a * p.x * d.x + b * p.y * d.y + c * p.z * d.z;
    double C = a * p.x * p.x + b * p.y * p.y + c * p.z * p.z + r;
    double cmag = f_abs( a ) * p.x * p.x + f_abs( b ) * p.y * p.y + f_abs( c ) * p.z * p.z + f_abs( r );
    double amag = f_abs( a ) * d.x * d.x + f_abs( b ) * d.y * d.y + f_abs( c ) * d.z * d.z;
    C -= 1.0e-9 * cmag + 1.0e-12;
    if( !( A > 1.0e-6 * amag ) ) return iv_all();   /* not convex along this ray, or ill conditioned: no statement */
    double disc = B * B - A * C;
    if( !( disc >= 0 ) ) return iv_none();
    double sq = acn_sqrt( disc );
    Iv iv; iv.lo = iv_widen_lo( ( -B - sq ) / A ); iv.hi = iv_widen_hi( ( -B + sq ) / A );
    return iv_and( iv, iv_all() );
}

/* Two intervals per sub-object X (envelopes make them differ: obj_ray_hit tests an envelope once per RAY,
 * objects.c:264, so hits may lie outside the envelope ball, while obj_side clips per POINT, objects.c:368):
 *   H( X )  where a surface point of X can be reported as a hit (from any origin on the line),
 *   S( X )  where a point can be classified "inside X".
 *   leaf            H = S = primitive interval; plane: H is the crossing point alone (plane_ray_hit's own expression)
 *   NEG( X )        H = H( X ),  S = complement (a half line for a plane, else the whole ray)
 *   pair_inside     H = hull( H(A) n S(B), H(B) n S(A) )    S = S(A) n S(B)      (objects.c:1057-1092: a reported
 *                                                            hit is a surface point of one child inside the other)
 *   pair_outside    H = hull( H(A) u H(B) )                 S = hull( S(A) u S(B) )
 *   envelope E      ray misses E: H = S = empty; else S = S n chord( E ), H unchanged
 * The object is skipped iff H( root ) has no point in [ -W, limit + W ]. */
enum { ACN_PO_END = 0, ACN_PO_PLANE, ACN_PO_SPHERE, ACN_PO_QUAD, ACN_PO_ALL, ACN_PO_NEG, ACN_PO_AND, ACN_PO_OR, ACN_PO_ENV };
#define ACN_PO( op, node ) ( ( uint32_t )( op ) | ( ( uint32_t )( node ) << 4 ) )
#ifdef ACN_PRUNE_CHECK
#define ACN_FAST_PRUNE( ... ) false
#else
#define ACN_FAST_PRUNE( ... ) prune_run( __VA_ARGS__ )
#endif
#define ACN_PRUNE_STACK 5

/* true: `node` (a wave-uniform index) cannot report a hit at any t <= limit.  false also when it has no program. */
template< class NP >
DEV bool prune_exec( NP nodes, ElemP elems, int pc, V3 rp, V3 rd, double limit );

/* true: `node` (a wave-uniform index) cannot report a hit at any t <= limit.  false also when it has no program.
 * Compiled in only for the DevScenePT kernel variants: the interval stack costs ~40 VGPRs, which the spill-free
 * k_shade of scenes without programs (wine_glass) must not pay for. */
template< class SC >
DEV bool prune_run( const SC& sc, int node, V3 rp, V3 rd, double limit )
{
    if constexpr( SC::prune )
    {
        int pc = sc.elems[ sc.prune_base + ( uint32_t )node ];
        if( pc < 0 ) return false;
        return prune_exec( sc.nodes, sc.elems, pc, rp, rd, limit );
    }
    else return false;
}

template< class NP >
DEV bool prune_exec( NP nodes, ElemP elems, int pc, V3 rp, V3 rd, double limit )
{
    Iv h0 = iv_all(), h1 = iv_all(), h2 = iv_all(), h3 = iv_all(), h4 = iv_all();
    Iv s0 = iv_all(), s1 = iv_all(), s2 = iv_all(), s3 = iv_all(), s4 = iv_all();
    for( ;; pc++ )
    {
        uint32_t w = ( uint32_t )elems[ pc ];
        uint32_t op = w & 15u;
        if( op == ACN_PO_END ) break;
        auto n = &nodes[ w >> 4 ];
        if( op <= ACN_PO_ALL )
        {
            Iv xs = iv_all(), xh = iv_all();
            if( op == ACN_PO_PLANE )
            {
                V3 pos = ld3( n->pos ), nor = ld3( n->rax + 6 );
                xs = iv_halfspace( pos, nor, rp, rd, false );
                double div = v_mlv( nor, rd );   /* plane_ray_hit (objects.c:530-541) */
                if( div == 0 ) xh = iv_none();
                else { double offs = v_sub_mlv( pos, rp, nor ) / div; xh.lo = iv_widen_lo( offs ); xh.hi = iv_widen_hi( offs ); xh = iv_and( xh, iv_all() ); }
            }
            else if( op == ACN_PO_SPHERE ) xs = xh = iv_ball( ld3( n->pos ), n->prm[ 0 ], rp, rd );
            else if( op == ACN_PO_QUAD )   xs = xh = iv_squaroid( n, rp, rd );
            h4 = h3; h3 = h2; h2 = h1; h1 = h0; h0 = xh;
            s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = xs;
        }
        else if( op == ACN_PO_NEG )
        {
            /* complement: a plane's is the other half line (w carries the plane's node), anything else: no statement */
            s0 = ( w >> 4 ) ? iv_halfspace( ld3( n->pos ), ld3( n->rax + 6 ), rp, rd, true ) : iv_all();
        }
        else if( op == ACN_PO_ENV )
        {
            if( !env_ray_hits_( n, rp, rd, ACN_NO_CNT ) ) { h0 = iv_none(); s0 = iv_none(); }   /* the machine's own test (objects.c:264) */
            else s0 = iv_and( s0, iv_ball( ld3( n->env_pos ), n->env_radius, rp, rd ) );
        }
        else
        {
            if( op == ACN_PO_AND ) { h0 = iv_or( iv_and( h1, s0 ), iv_and( h0, s1 ) ); s0 = iv_and( s1, s0 ); }
            else                   { h0 = iv_or( h1, h0 ); s0 = iv_or( s1, s0 ); }
            h1 = h2; h2 = h3; h3 = h4;
            s1 = s2; s2 = s3; s3 = s4;
        }
    }
    h0.hi = f_min( h0.hi, iv_widen_hi( limit ) );
    return iv_empty( h0 );
}

/* Hit test of ROOT element `e` (an index that is the same in every active lane of the wave, so the node is read
 * through the scalar cache into SGPRs and the type dispatch is a scalar branch): obj_ray_hit (objects.c:261-284)
 * for objects -- plane / sphere / squaroid inline, CSG and SDF objects through the hit machine -- and
 * compound_s_ray_hit for nested compounds. */
template< bool NOR, class SC, class CT >
DEV double element_hit( const SC& sc, int e, V3 rp, V3 rd, V3* nor, int* hit_obj, double limit, CT* cnt )
{
    ACN_NODE_UNIFORM( n, &sc.nodes[ e ] )
    int type = n->type;
    if( type == ACN_COMPOUND )
    {
        if constexpr( SC::prune )   /* the "extras" kernel variants (DevScenePT) */
        {
            if( n->flags & ACN_GFLAG_SIMPLE_COMPOUND )
            {
                if( node_has_env( n ) && !env_ray_hits( n, rp, rd ) ) return F3_INF;
                return simple_compound_hit< NOR >( sc, e, rp, rd, nor, hit_obj, limit, cnt );
            }
        }
        ACN_LAP( PH_ROOT_LEAF );
        /* a real call: it gets the addresses of two locals, so that the caller's normal and hit object -- which every other
         * branch of this function writes too -- are not forced into scratch memory by an escaping pointer */
        V3 cn = mk( 0, 0, 0 );
        int co = *hit_obj;
        double ac = compound_ray_hit_dev( sref( sc ), e, rp, rd, NOR, &cn, &co, limit, cnt );
        if( ac < F3_INF ) { if constexpr( NOR ) *nor = cn; *hit_obj = co; }
        ACN_LAP( PH_COMPOUND );
        return ac;
    }
    *hit_obj = e;
    bool env = node_has_env( n );
    if( env && !env_ray_hits( n, rp, rd ) ) { cnt->inc( CNT_OBJ_HIT ); return F3_INF; }
    if( type > ACN_SQUAROID )
    {
#ifdef ACN_PRUNE_CHECK
        if( type != ACN_DISTANCE && surely_outside< ACN_PRUNE_DEPTH >( sc, e, rp, rd ) ) { cnt->inc( CNT_OBJ_HIT ); return F3_INF; }
        {
            bool pr = type != ACN_DISTANCE && prune_run( sc, e, rp, rd, F3_INF );
            V3 nn = mk( 0, 0, 0 );
            double aa = obj_ray_hit_dev( sref( sc ), e, rp, rd, NOR, NOR ? nor : &nn, cnt );
            if( pr && aa < F3_INF ) printf( "PRUNE VIOLATION e=%d a=%.17g rp=%.17g %.17g %.17g rd=%.17g %.17g %.17g\n", e, aa, rp.x, rp.y, rp.z, rd.x, rd.y, rd.z );
            return aa;
        }
#endif
        ACN_LAP( PH_ROOT_LEAF );
        if( type != ACN_DISTANCE && ( surely_outside< ACN_PRUNE_DEPTH >( sc, e, rp, rd ) || prune_run( sc, e, rp, rd, F3_INF ) ) ) { cnt->inc( CNT_OBJ_HIT ); ACN_LAP( PH_PRUNE ); return F3_INF; }
        ACN_LAP( PH_PRUNE );
#if ACN_UNI_MACHINE
        return obj_ray_hit_uni< NOR, SC::park >( sref( sc ), e, rp, rd, nor, cnt );   /* the machine redoes the envelope test */
#else
        return obj_ray_hit_dev( sref( sc ), e, rp, rd, NOR, nor, cnt );
#endif
    }
    cnt->inc( CNT_OBJ_HIT );
    double a;
    if( type == ACN_PLANE )       a = plane_ray_hit( ld3( n->pos ), ld3( n->rax + 6 ), rp, rd, NOR, nor );
    else if( type == ACN_SPHERE ) a = sphere_ray_hit( ld3( n->pos ), n->prm[ 0 ], rp, rd, NOR, nor );
    else                          a = squaroid_ray_hit( n, rp, rd, NOR, nor );
    if( NOR && a < F3_INF && n->surface_roughness > 0 ) *nor = roughness_normal( n, *nor, ray_pos( rp, rd, a ) );
    return a;
}

/* obj_ray_hit of a light that is not a plain sphere / plane (k_shade's LEAF_LIGHTS = false variants): a real function
 * call, so that those rarely used kernels do not each carry an in-line copy of the CSG machines */
template< class SC, class CT >
DEVN double light_hit_call( SC sc, int e, V3 rp, V3 rd, CT* cnt )
{
    int ho;
    return element_hit< false >( sc, e, rp, rd, ( V3* )nullptr, &ho, -F3_INF, cnt );
}

/* Broad phase of a root loop (ACN_ROOT_CANDIDATES; built, measured, off).  A root compound of a lamp scene has 63 elements, and the loops below visit every one of them for
 * every ray: element index -> node header -> envelope, three dependent scalar loads (~200 cycles each) before the envelope test
 * that rejects the ray for nearly all of them (objects.c:264: a ray that misses an element's envelope gets f3_inf).  The
 * envelopes of the slice elems[ first .. first + count ) lie side by side in env_tab (32 bytes per element, the same order), so
 * this loop is independent loads the compiler overlaps, and its result -- bit i: the ray enters element i's envelope, or the
 * element has none -- lets the loops skip an element NO lane of the wave needs without touching its node, and lets the other
 * lanes sit an element out.  Results do not change: a skipped evaluation is one that returns f3_inf.  Elements from the 65th on
 * are always candidates. */
template< class SC, class CT >
DEV uint64_t root_candidates( const SC& sc, int first, int count, V3 rp, V3 rd, CT* cnt )
{
    uint64_t m = 0;
#ifndef ACN_ROOT_CANDIDATES   /* OFF: measured neutral on every workload (hanging_lamp 600x800 305 vs 308 ms, paraffin_lamp 335 - 346 both
                                 ways, hanging_lamp 2160p every 256th pixel 4 639 vs 4 614 ms: profiles/r04/ab_root_candidates_s15.txt) -- the
                                 root loop's chain of scalar loads is not what the lamp scenes wait for */
    return ~0ull;
#endif
    const int n = count < 64 ? count : 64;
    #pragma unroll 4
    for( int i = 0; i < n; i++ )
    {
        const CDblP e = sc.env_tab + 4 * ( size_t )( first + i );
        const double r = e[ 3 ];
        bool cand = true;
        if( r >= 0 )
        {
            cand = env_ray_hits_raw( mk( e[ 0 ], e[ 1 ], e[ 2 ] ), r, rp, rd, ACN_NO_CNT );
            if( !cand ) { cnt->inc( CNT_OBJ_HIT ); cnt->cost( ACN_F_ENV_MISS ); }   /* what element_hit books for such a ray */
        }
        if( cand ) m |= 1ull << i;
    }
    return m;
}

/* compound_s_ray_hit on a root compound, any-hit form for occlusion tests: true iff some element hits at <= limit */
template< class SC, class CT >
DEV bool root_occluded( const SC& sc, int cmp, V3 rp, V3 rd, double limit, CT* cnt )
{
    auto o = &sc.nodes[ cmp ];
    if( node_has_env( o ) && !env_ray_hits( o, rp, rd ) ) return false;
    int first = o->child0 + ( int )sc.n_elems, count = o->child1;   /* the cost-ordered copy: cheap elements first */
    const uint64_t cand = root_candidates( sc, first, count, rp, rd, cnt );
    bool occ = false;
    for( int i = 0; i < count; i++ )
    {
        const bool mine = !occ && ( i >= 64 || ( ( cand >> i ) & 1ull ) );
        if( __ballot( mine ) == 0ull ) continue;
        int element = __builtin_amdgcn_readfirstlane( sc.elems[ first + i ] );
        if( mine )
        {
            int hit_obj;
            double a = element_hit< false >( sc, element, rp, rd, nullptr, &hit_obj, limit, cnt );
            if( a <= limit ) occ = true;
        }
        if( __ballot( !occ ) == 0ull ) break;
    }
    return occ;
}

struct Trans { V3 exit_nor; int exit_obj; int enter_obj; };

/* compound_s_ray_trans_hit on a root compound (compound.c:246-299) */
template< class SC, class CT >
DEV double root_trans_hit( const SC& sc, int cmp, V3 rp, V3 rd, Trans* trans, CT* cnt )
{
    auto o = &sc.nodes[ cmp ];
    cnt->inc( CNT_TRANS_RAY );
    if( node_has_env( o ) && !env_ray_hits( o, rp, rd ) ) return F3_INF;
    double min_a = F3_INF;
    int first = o->child0, count = o->child1;
    const uint64_t cand = root_candidates( sc, first, count, rp, rd, cnt );
    for( int i = 0; i < count; i++ )
    {
        const bool mine = i >= 64 || ( ( cand >> i ) & 1ull );
        if( __ballot( mine ) == 0ull ) continue;
        int element = __builtin_amdgcn_readfirstlane( sc.elems[ first + i ] );
        int hit_obj = -1;
        V3 nor = mk( 0, 0, 0 );
        double a = F3_INF;
        if( mine ) a = element_hit< true >( sc, element, rp, rd, &nor, &hit_obj, -F3_INF, cnt );
        if( a < F3_INF )
        {
            cnt->cost( ACN_F_TRANS_RESOLVE );
            if( a < min_a - F3_EPS )
            {
                min_a = a;
                if( v_mlv( nor, rd ) > 0 )
                {
                    trans->exit_nor = nor; trans->exit_obj = hit_obj; trans->enter_obj = -1;
                }
                else
                {
                    trans->exit_nor = v_neg( nor ); trans->exit_obj = -1; trans->enter_obj = hit_obj;
                }
            }
            else if( f_abs( a - min_a ) < F3_EPS )
            {
                min_a = a < min_a ? a : min_a;
                if( v_mlv( nor, rd ) > 0 ) trans->exit_obj = hit_obj;
                else                       trans->enter_obj = hit_obj;
            }
        }
    }
    return min_a;
}

/* ---- fast-path forms for k_shade: leaf root elements are tested inline; a ray that gets inside the envelope of a
 * root element that needs the machine (CSG, SDF, nested compound) is reported as `hard` and handed to the hard-ray
 * kernels, which redo the query with the full traversal.  The results are identical: an occlusion test is an OR over
 * the elements, and a transition hit is only computed here when every machine element was missed at its envelope
 * (obj_ray_hit then returns f3_inf for it, objects.c:264). ---- */
DEV bool is_fast_type( int type ) { return type >= ACN_PLANE && type <= ACN_SQUAROID; }

template< bool NOR, class NP, class CT >
DEV double leaf_element_hit( NP n, int type, V3 rp, V3 rd, V3* nor, CT* cnt )
{
    cnt->inc( CNT_OBJ_HIT );
    if( node_has_env( n ) && !env_ray_hits( n, rp, rd ) ) return F3_INF;
    double a;
    if( type == ACN_PLANE )       a = plane_ray_hit( ld3( n->pos ), ld3( n->rax + 6 ), rp, rd, NOR, nor );
    else if( type == ACN_SPHERE ) a = sphere_ray_hit( ld3( n->pos ), n->prm[ 0 ], rp, rd, NOR, nor );
    else                          a = squaroid_ray_hit( n, rp, rd, NOR, nor );
    if( NOR && a < F3_INF && n->surface_roughness > 0 ) *nor = roughness_normal( n, *nor, ray_pos( rp, rd, a ) );
    return a;
}

/* a ROOT element that is a leaf pair: obj_ray_hit (objects.c:261-284) in line, no machine, no deferral */
template< bool NOR, class SC, class NP, class CT >
DEV double leaf_pair_element_hit( const SC& sc, NP n, V3 rp, V3 rd, V3* nor, CT* cnt )
{
    cnt->inc( CNT_OBJ_HIT );
    if( node_has_env( n ) && !env_ray_hits( n, rp, rd ) ) return F3_INF;
    V3 nn = mk( 0, 0, 0 );
    double a = leaf_pair_hit_uniform( sref( sc ), n, rp, rd, NOR, &nn, cnt );
    if( NOR && a < F3_INF )
    {
        if( n->surface_roughness > 0 ) nn = roughness_normal( n, nn, ray_pos( rp, rd, a ) );
        *nor = nn;
    }
    return a;
}

/* Cone culling of the root elements for the direct-light loop of one shading point (k_shade).  All of a loop's shadow
 * rays start at `pos` and lie inside the cone the light is sampled in: out_d = src_con * sphere_cap( cyl_hgt ) has
 * out_d . axis = 1 - u * cyl_hgt >= cos_theta (scene.c:549-558, vectors.h:197-206).  Bit i of the result: element i of the
 * root compound cannot be hit by ANY ray of that cone -- obj_ray_hit would return f3_inf for every sample -- so the sample
 * loop skips it and every result stays what it was:
 *   - an element with an envelope (objects.c:264: the ray must hit the envelope sphere first), or a sphere leaf: the ball
 *     lies outside the cone when the angle between axis and centre exceeds theta + asin( R / L );
 *   - a plane (gmath.h:38-50): hit iff ( plane.pos - pos ) . nor and nor . rd have the same sign; over the cone nor . rd
 *     stays on one side of zero when the angle between nor and axis differs from 90 degrees by more than theta.
 * Conservative by 1e-9 (directions are unit vectors to ~1e-16); anything uncertain is kept.  Per shading point and
 * light ~40 flop per element instead of ~25 per element AND SAMPLE: on the wine glass a floor point outside the glass's
 * shadow tests nothing per sample (k_shade spent 23 % of its time in the occlusion test, profiles/r03). */
template< class SC >
DEV uint64_t root_cone_cull( const SC& sc, int cmp, V3 pos, V3 axis, double cos_theta )
{
    auto o = &sc.nodes[ cmp ];
    uint64_t skip = 0;
#ifdef ACN_NO_CONE_CULL
    return 0;
#endif
    if( !( cos_theta > 1.0e-6 ) ) return 0;                 /* half space or more (plane lights, points inside a light) */
    const double sin_theta = acn_sqrt( f_max( 0.0, 1.0 - cos_theta * cos_theta ) );
    const int first = o->child0, count = o->child1 < 64 ? o->child1 : 64;
    for( int i = 0; i < count; i++ )
    {
        int element = __builtin_amdgcn_readfirstlane( sc.elems[ first + i ] );
        auto n = &sc.nodes[ element ];
        const int type = n->type;
        bool out = false;
        if( node_has_env( n ) || type == ACN_SPHERE )
        {
            const bool env = node_has_env( n );
            const V3 c = env ? ld3( n->env_pos ) : ld3( n->pos );
            const double R = env ? n->env_radius : n->prm[ 0 ];
            const V3 v = v_sub( c, pos );
            const double L2 = v_sqr( v ), R2 = R * R;
            if( L2 > R2 * ( 1.0 + 1.0e-9 ) + 1.0e-30 )
            {
                const double L = acn_sqrt( L2 );
                const double cos_phi = v_mlv( v, axis ) / L;
                const double sin_alpha = R / L;
                const double cos_alpha = acn_sqrt( f_max( 0.0, 1.0 - sin_alpha * sin_alpha ) );
                const double cos_sum = cos_theta * cos_alpha - sin_theta * sin_alpha;     /* cos( theta + alpha ), theta + alpha < pi */
                out = cos_phi < cos_sum - 1.0e-9;
            }
        }
        else if( type == ACN_PLANE )
        {
            const V3 nor = ld3( n->rax + 6 );
            const double s0 = v_sub_mlv( ld3( n->pos ), pos, nor );
            const double c = v_mlv( nor, axis );                                           /* cos of the angle between nor and axis */
            const double s = acn_sqrt( f_max( 0.0, 1.0 - c * c ) );
            const double d_min = c * cos_theta - s * sin_theta, d_max = c * cos_theta + s * sin_theta;   /* range of nor . rd over the cone */
            out = ( s0 < 0 && d_min > 1.0e-9 ) || ( s0 > 0 && d_max < -1.0e-9 );
        }
        if( out ) skip |= 1ull << i;
    }
    return skip;
}

/* 0: not occluded, 1: occluded, 2: undecided (hard).  skip: root_cone_cull's bits */
template< class SC, class CT >
DEV int root_occluded_fast( const SC& sc, int cmp, V3 rp, V3 rd, double limit, uint64_t skip, CT* cnt )
{
    auto o = &sc.nodes[ cmp ];
    if( node_has_env( o ) && !env_ray_hits( o, rp, rd ) ) return 0;
    int first = o->child0, count = o->child1;
    bool hard = false;
    for( int i = 0; i < count; i++ )
    {
        if( i < 64 && ( ( skip >> i ) & 1ull ) ) continue;
        int element = __builtin_amdgcn_readfirstlane( sc.elems[ first + i ] );
        ACN_NODE( n, &sc.nodes[ element ] )
        int type = n->type;
        if( is_fast_type( type ) )
        {
            double a = leaf_element_hit< false >( n, type, rp, rd, nullptr, cnt );
            if( a <= limit ) return 1;
        }
        else if( n->flags & ACN_GFLAG_LEAF_PAIR )
        {
            double a = leaf_pair_element_hit< false >( sc, n, rp, rd, nullptr, cnt );
            if( a <= limit ) return 1;
        }
        else if( SC::prune && ( n->flags & ACN_GFLAG_SIMPLE_COMPOUND ) )
        {
            if( node_has_env( n ) && !env_ray_hits( n, rp, rd ) ) continue;
            int ho;
            double a = simple_compound_hit< false >( sc, element, rp, rd, nullptr, &ho, limit, cnt );
            if( a <= limit ) return 1;
        }
        else if( type == ACN_COMPOUND || type == ACN_DISTANCE ? ( !node_has_env( n ) || env_ray_hits_( n, rp, rd, ACN_NO_CNT ) )
                                                              : !( surely_outside< ACN_PRUNE_DEPTH >( sc, element, rp, rd ) || ACN_FAST_PRUNE( sc, element, rp, rd, limit ) ) )
        {
            hard = true;
        }
    }
    return hard ? 2 : 0;
}

/* compound_s_ray_trans_hit on a root compound; *hard is set when the query must be redone by the full traversal */
template< class SC, class CT >
DEV double root_trans_hit_fast( const SC& sc, int cmp, V3 rp, V3 rd, Trans* trans, bool* hard, CT* cnt )
{
    auto o = &sc.nodes[ cmp ];
    *hard = false;
    if( node_has_env( o ) && !env_ray_hits( o, rp, rd ) ) { cnt->inc( CNT_TRANS_RAY ); return F3_INF; }
    double min_a = F3_INF;
    int first = o->child0, count = o->child1;
    bool h = false;
    for( int i = 0; i < count; i++ )
    {
        int element = __builtin_amdgcn_readfirstlane( sc.elems[ first + i ] );
        ACN_NODE( n, &sc.nodes[ element ] )
        int type = n->type;
        if( !is_fast_type( type ) && !( n->flags & ( ACN_GFLAG_LEAF_PAIR | ( SC::prune ? ACN_GFLAG_SIMPLE_COMPOUND : 0u ) ) ) )
        {
            if( type == ACN_COMPOUND || type == ACN_DISTANCE ? ( !node_has_env( n ) || env_ray_hits_( n, rp, rd, ACN_NO_CNT ) )
                                                             : !( surely_outside< ACN_PRUNE_DEPTH >( sc, element, rp, rd ) || ACN_FAST_PRUNE( sc, element, rp, rd, F3_INF ) ) ) h = true;
            continue;
        }
        V3 nor = mk( 0, 0, 0 );
        double a;
        if( is_fast_type( type ) ) a = leaf_element_hit< true >( n, type, rp, rd, &nor, cnt );
        else if( n->flags & ACN_GFLAG_LEAF_PAIR ) a = leaf_pair_element_hit< true >( sc, n, rp, rd, &nor, cnt );
        else
        {
            /* the hit object of a compound is the leaf that was hit (compound.c:225-243) */
            a = F3_INF;
            if constexpr( SC::prune )
            {
                if( !node_has_env( n ) || env_ray_hits( n, rp, rd ) ) a = simple_compound_hit< true >( sc, element, rp, rd, &nor, &element, -F3_INF, cnt );
            }
        }
        if( a < F3_INF )
        {
            cnt->cost( ACN_F_TRANS_RESOLVE );
            if( a < min_a - F3_EPS )
            {
                min_a = a;
                if( v_mlv( nor, rd ) > 0 )
                {
                    trans->exit_nor = nor; trans->exit_obj = element; trans->enter_obj = -1;
                }
                else
                {
                    trans->exit_nor = v_neg( nor ); trans->exit_obj = -1; trans->enter_obj = element;
                }
            }
            else if( f_abs( a - min_a ) < F3_EPS )
            {
                min_a = a < min_a ? a : min_a;
                if( v_mlv( nor, rd ) > 0 ) trans->exit_obj = element;
                else                       trans->enter_obj = element;
            }
        }
    }
    *hard = h;
    if( !h ) cnt->inc( CNT_TRANS_RAY );
    return min_a;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* Pooled machines (ACN_POOLED=1; built, parity-green, measured SLOWER, off -- see the end of this comment).  A lock-step machine
 * serves the lanes of ONE wave that meet the same root element, and the rays of a wave
 * rarely agree on one: 18 of 64 lanes per entry on the wine glass, 4.4 on hanging_lamp (63 root elements), 7 in its k_hard_path
 * (profiles/r04/phase_ticks_*.txt) -- the machine kernels of the lamp scenes ran at 7 - 10 % of their lanes.  The four waves of a
 * workgroup now POOL the rays that need the same element: every lane with `need` writes its ray into the workgroup's pool in LDS
 * (slots handed out by ballot + the waves' counts), the pooled rays are evaluated in batches of 64 by as few waves as it takes
 * (in turn, so that the work spreads over the SIMDs), and every lane collects its result from its slot: one machine entry with
 * up to 64 lanes instead of four with a quarter each.  A ray's own arithmetic is untouched -- which wave evaluates it is not
 * part of any result.  ALL lanes of the workgroup must call (three barriers); the callers' loops are workgroup-uniform.
 * Measured (profiles/r04/ab_pooled_s17.txt, same box, parity suite green with it): 1080p 49.9 -> 60.4 ms, hanging_lamp 600x800
 * 307 -> 430, paraffin_lamp 340 -> 393, diamond every 16th pixel 2 245 -> 2 659.  Lanes per machine entry were the wrong target:
 * the four waves ran their quarter-full machines SIDE BY SIDE on four SIMDs, and these kernels wait for latency, not for issue
 * slots -- pooling puts the same evaluations one after another on one SIMD while three waves stand at a barrier. */
struct RayPool
{
    LdsF64P v;          /* plane k of entry i: v[ k * 256 + i ]; in: origin 0 - 2, direction 3 - 5; out: a 0, normal 1 - 3 */
    LdsU32P counts;     /* [ 2 ][ 4 ]: the waves' counts of the call at hand, double-buffered by the parity of `turn` */
    uint32_t turn;      /* calls so far: parity of the counts, and the rotation of batches over the waves */
};
template< class SC > DEV RayPool ray_pool_of( const SC& sc )
{
    RayPool pl;
    pl.v = ( LdsF64P )( ( char ACN_LDS* )acn_lds_raw + sc.lds_stack + ACN_LDS_STACK_BYTES );
    pl.counts = ( LdsU32P )( pl.v + 6 * ACN_LDS_LANES ) + ACN_LDS_LANES;
    pl.turn = 0;
    return pl;
}

/* obj_ray_hit of root element e (a CSG / SDF object) for every lane of the WORKGROUP with `need`; f3_inf for the others */
template< bool NOR, class SC, class CT >
DEV double pooled_machine_hit( const SC& sc, RayPool& pl, int e, bool need, V3 rp, V3 rd, V3* nor, CT* cnt )
{
    const uint32_t wave = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )( threadIdx.x >> 6 ) );
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long m = __ballot( need );
    const uint32_t par = ( pl.turn & 1u ) * 4u;
    const uint32_t turn = pl.turn++;
    if( lane == 0 ) pl.counts[ par + wave ] = ( uint32_t )__popcll( m );
    __syncthreads();
    uint32_t total = 0, off = 0;
    for( uint32_t w = 0; w < 4u; w++ )
    {
        const uint32_t c = ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )pl.counts[ par + w ] );
        if( w < wave ) off += c;
        total += c;
    }
    if( total == 0 ) return F3_INF;      /* the same in every wave of the workgroup */
    const uint32_t slot = off + ( uint32_t )__builtin_amdgcn_mbcnt_hi( ( uint32_t )( m >> 32 ), __builtin_amdgcn_mbcnt_lo( ( uint32_t )m, 0u ) );
    if( need )
    {
        pl.v[ slot ] = rp.x; pl.v[ ACN_LDS_LANES + slot ] = rp.y; pl.v[ 2 * ACN_LDS_LANES + slot ] = rp.z;
        pl.v[ 3 * ACN_LDS_LANES + slot ] = rd.x; pl.v[ 4 * ACN_LDS_LANES + slot ] = rd.y; pl.v[ 5 * ACN_LDS_LANES + slot ] = rd.z;
    }
    __syncthreads();
    for( uint32_t b = 0; b * 64u < total; b++ )
    {
        if( ( ( b + turn ) & 3u ) != wave ) continue;      /* batch b is this wave's turn */
        const uint32_t idx = b * 64u + lane;
        if( idx < total )
        {
            const V3 p = mk( pl.v[ idx ], pl.v[ ACN_LDS_LANES + idx ], pl.v[ 2 * ACN_LDS_LANES + idx ] );
            const V3 d = mk( pl.v[ 3 * ACN_LDS_LANES + idx ], pl.v[ 4 * ACN_LDS_LANES + idx ], pl.v[ 5 * ACN_LDS_LANES + idx ] );
            V3 n = mk( 0, 0, 0 );
            const double a = obj_ray_hit_uni< NOR >( sref( sc ), e, p, d, &n, cnt );
            pl.v[ idx ] = a;
            if( NOR ) { pl.v[ ACN_LDS_LANES + idx ] = n.x; pl.v[ 2 * ACN_LDS_LANES + idx ] = n.y; pl.v[ 3 * ACN_LDS_LANES + idx ] = n.z; }
        }
    }
    __syncthreads();
    double a = F3_INF;
    if( need )
    {
        a = pl.v[ slot ];
        if( NOR && a < F3_INF ) *nor = mk( pl.v[ ACN_LDS_LANES + slot ], pl.v[ 2 * ACN_LDS_LANES + slot ], pl.v[ 3 * ACN_LDS_LANES + slot ] );
    }
    return a;
}

/* element_hit for the pooled root loops: `live` lanes want the element tested; machine elements go through the pool.  Every lane
 * of the workgroup calls with the same e. */
template< bool NOR, class SC, class CT >
DEV double element_hit_pooled( const SC& sc, RayPool& pl, int e, bool live, V3 rp, V3 rd, V3* nor, int* hit_obj, double limit, CT* cnt )
{
    ACN_NODE_UNIFORM( n, &sc.nodes[ e ] )
    const int type = n->type;
    if( type == ACN_COMPOUND || type <= ACN_SQUAROID )
    {
        double a = F3_INF;
        if( live ) a = element_hit< NOR >( sc, e, rp, rd, nor, hit_obj, limit, cnt );   /* no machine in these branches */
        return a;
    }
    bool need = live;
    if( need )
    {
        *hit_obj = e;
        if( node_has_env( n ) && !env_ray_hits( n, rp, rd ) ) { cnt->inc( CNT_OBJ_HIT ); need = false; }
        else if( type != ACN_DISTANCE && ( surely_outside< ACN_PRUNE_DEPTH >( sc, e, rp, rd ) || prune_run( sc, e, rp, rd, limit >= 0 ? limit : F3_INF ) ) ) { cnt->inc( CNT_OBJ_HIT ); need = false; }
    }
    return pooled_machine_hit< NOR >( sc, pl, e, need, rp, rd, nor, cnt );
}

/* root_occluded for the lanes with `want`, pooled */
template< class SC, class CT >
DEV bool root_occluded_pooled( const SC& sc, RayPool& pl, int cmp, bool want, V3 rp, V3 rd, double limit, CT* cnt )
{
    auto o = &sc.nodes[ cmp ];
    if( want && node_has_env( o ) && !env_ray_hits( o, rp, rd ) ) want = false;
    int first = o->child0 + ( int )sc.n_elems, count = o->child1;   /* the cost-ordered copy: cheap elements first */
    bool occ = false;
    for( int i = 0; i < count; i++ )
    {
        int element = __builtin_amdgcn_readfirstlane( sc.elems[ first + i ] );
        int hit_obj;
        double a = element_hit_pooled< false >( sc, pl, element, want && !occ, rp, rd, ( V3* )nullptr, &hit_obj, limit, cnt );
        if( a <= limit ) occ = true;
    }
    return occ;
}

/* root_trans_hit for the lanes with `live`, pooled */
template< class SC, class CT >
DEV double root_trans_hit_pooled( const SC& sc, RayPool& pl, int cmp, bool live, V3 rp, V3 rd, Trans* trans, CT* cnt )
{
    auto o = &sc.nodes[ cmp ];
    if( live ) cnt->inc( CNT_TRANS_RAY );
    if( live && node_has_env( o ) && !env_ray_hits( o, rp, rd ) ) live = false;
    double min_a = F3_INF;
    int first = o->child0, count = o->child1;
    for( int i = 0; i < count; i++ )
    {
        int element = __builtin_amdgcn_readfirstlane( sc.elems[ first + i ] );
        int hit_obj = -1;
        V3 nor = mk( 0, 0, 0 );
        double a = element_hit_pooled< true >( sc, pl, element, live, rp, rd, &nor, &hit_obj, -F3_INF, cnt );
        if( a < F3_INF )
        {
            cnt->cost( ACN_F_TRANS_RESOLVE );
            if( a < min_a - F3_EPS )
            {
                min_a = a;
                if( v_mlv( nor, rd ) > 0 )
                {
                    trans->exit_nor = nor; trans->exit_obj = hit_obj; trans->enter_obj = -1;
                }
                else
                {
                    trans->exit_nor = v_neg( nor ); trans->exit_obj = -1; trans->enter_obj = hit_obj;
                }
            }
            else if( f_abs( a - min_a ) < F3_EPS )
            {
                min_a = a < min_a ? a : min_a;
                if( v_mlv( nor, rd ) > 0 ) trans->exit_obj = hit_obj;
                else                       trans->enter_obj = hit_obj;
            }
        }
    }
    return min_a;
}

/* scene_s_trans_hit (scene.c:362-382) for the lanes with `live`, pooled: every lane of the workgroup calls */
template< class SC, class CT >
DEV double scene_trans_hit_pooled( const SC& sc, RayPool& pl, bool live, V3 rp, V3 rd, Trans* trans, CT* cnt )
{
    double min_a = F3_INF;
    double a;
    Trans trans_l;
    trans_l.exit_nor = mk( 0, 0, 0 ); trans_l.exit_obj = -1; trans_l.enter_obj = -1;
    #pragma unroll 1
    for( int k = 0; k < 2; k++ )
    {
        if( ( a = root_trans_hit_pooled( sc, pl, k ? sc.matter_root : sc.light_root, live, rp, rd, &trans_l, cnt ) ) < min_a )
        {
            min_a = a;
            *trans = trans_l;
        }
    }
    return min_a;
}

template< class SC, class CT >
DEV double scene_trans_hit_dev( const SC& sc, V3 rp, V3 rd, Trans* trans, CT* cnt )   /* scene.c:362-382 */
{
    double min_a = F3_INF;
    double a;
    Trans trans_l;
    trans_l.exit_nor = mk( 0, 0, 0 ); trans_l.exit_obj = -1; trans_l.enter_obj = -1;
    ACN_LAP( PH_FETCH );
    /* lights, then matter, through ONE in-line copy of the root traversal (and of the CSG machines in it): the root is a scalar */
#ifdef ACN_ROOT_TWO_COPIES
    #pragma unroll
#else
    #pragma unroll 1
#endif
    for( int k = 0; k < 2; k++ )
    {
        if( ( a = root_trans_hit( sc, k ? sc.matter_root : sc.light_root, rp, rd, &trans_l, cnt ) ) < min_a )
        {
            min_a = a;
            *trans = trans_l;
        }
        if( k == 0 ) ACN_LAP( PH_LIGHT );
    }
    ACN_LAP( PH_ROOT_LEAF );
    return min_a;
}

/* ---- fov of a light: objects.c:254-259 -> :520-527, :619-637, :1035-1045 ---- */
DEV void sphere_fov( V3 center, double radius, V3 pos, V3* dir, double* cos_rs )
{
    V3 diff = v_sub( center, pos );
    *dir = v_of_length( diff, 1.0 );
    double diff_sqr = v_sqr( diff );
    double radius_sqr = f_sqr( radius );
    *cos_rs = ( diff_sqr > radius_sqr ) ? acn_sqrt( 1.0 - ( radius_sqr / diff_sqr ) ) : -1;
}

template< class NP > DEV void obj_fov_dev( NP o, V3 pos, V3* dir, double* cos_rs )
{
    if( o->type == ACN_PLANE )
    {
        *dir = v_neg( ld3( o->rax + 6 ) );
        *cos_rs = v_mlv( v_sub( ld3( o->pos ), pos ), *dir ) > 0 ? 0 : 1;
    }
    else if( o->type == ACN_SPHERE )
    {
        sphere_fov( ld3( o->pos ), o->prm[ 0 ], pos, dir, cos_rs );
    }
    else if( node_has_env( o ) )
    {
        sphere_fov( ld3( o->env_pos ), o->env_radius, pos, dir, cos_rs );
    }
    else
    {
        *dir = v_of_length( v_sub( ld3( o->pos ), pos ), 1.0 );
        *cos_rs = 0;
    }
}

/* scene.c:394-416 */
DEV double oren_nayar_weight( double weight, double theta_i, double on_a, double on_b, V3 out_d, V3 nor, V3 ray_prj )
{
    double theta_r = acn_acos( weight );
    double cos_phi = -v_mlv( v_of_length( v_orthogonal_projection( out_d, nor ), 1.0 ), ray_prj );
    double ta = f_max( theta_i, theta_r ), tb = f_min( theta_i, theta_r );
    double s1, c1, s2, c2;
    acn_sincos( ta, &s1, &c1 );
    acn_sincos( tb, &s2, &c2 );
    return weight * ( on_a + ( on_b * f_max( cos_phi, 0 ) * s1 * ( s2 / c2 ) ) );
}

/* the same with sin / cos of theta_i supplied by the caller (one value per shading point, many samples): of the two
 * angles only theta_r changes per sample, so one of the two sincos evaluations is loop invariant.  Bit-identical: the
 * values used are the same function results as above. */
DEV double oren_nayar_weight_pre( double weight, double theta_i, double sin_i, double cos_i, double on_a, double on_b, V3 out_d, V3 nor, V3 ray_prj )
{
    double theta_r = acn_acos( weight );
    double cos_phi = -v_mlv( v_of_length( v_orthogonal_projection( out_d, nor ), 1.0 ), ray_prj );
    double sr, cr;
    acn_sincos( theta_r, &sr, &cr );
    bool i_is_max = !( theta_i < theta_r );          /* f_max( theta_i, theta_r ) == theta_i (a > b ? a : b picks b on a tie) */
    double s1 = i_is_max ? sin_i : sr;
    double s2 = i_is_max ? sr : sin_i;
    double c2 = i_is_max ? cr : cos_i;
    return weight * ( on_a + ( on_b * f_max( cos_phi, 0 ) * s1 * ( s2 / c2 ) ) );
}

/* The same weight for the DIRECT-LIGHT loop (scene.c:556-576), without transcendentals.  There the weight reaches cl_sum and
 * nothing else -- no intensity, sample count, seed or branch -- so it need not carry the rounding of acos / sin / tan, only
 * their value.  With w = weight = out_d . nor (unit vectors), theta_r = acos( w ):
 *     cos( theta_r ) = w,   sin( theta_r ) = sr = sqrt( 1 - w^2 ) = | out_d - nor * w |   (the length v_of_length divides by),
 *     theta_i < theta_r  <=>  cos_i > w,
 *     cos_phi = m / sr   with   m = -( out_d - nor * w ) . ray_prj,
 * hence  max( cos_phi, 0 ) * sin( max ) * tan( min )  =  theta_i >= theta_r:  ( m+ / sr ) * sin_i * ( sr / w ) = m+ * sin_i / w
 *                                                        theta_i <  theta_r:  ( m+ / sr ) * sr * tan_i       = m+ * tan_i
 * and the weight is  w * on_a + on_b * m+ * ( cos_i <= w ? sin_i : w * tan_i ):  eleven multiply-adds and a select instead of
 * acos + sincos + sqrt + two divisions (16 % of k_shade's time, profiles/r03/NOTES.md section 2).  Agrees with
 * oren_nayar_weight to ~1e-15 relative (5e-9 relative where w < 1e-4, the range in which v_of_length returns its argument
 * unscaled, vectors.h:151: a sample whose weight is below 1e-4 to begin with); the path loop keeps the exact form because its
 * weight becomes the child's intensity (scene.c:596-617).  tan_i is used only where cos_i > w > 0. */
DEV double oren_nayar_weight_direct( double w, double sin_i, double cos_i, double tan_i, double on_a, double on_b, V3 out_d, V3 nor, V3 ray_prj )
{
    double m = -v_mlv( v_orthogonal_projection( out_d, nor ), ray_prj );
    m = f_max( m, 0 );
    return w * on_a + on_b * m * ( cos_i <= w ? sin_i : w * tan_i );
}

/* obj_color (objects.c:411-422): texture field if present (textures.c:99-102, 142-148), else prp.color.
 * obj_projection: plane objects.c:514-518, sphere :602-617, distance :893-896. */
DEV V3 obj_color_dev( const DevScene& sc, int node, V3 pos )
{
    MatP m = &sc.mats[ node ];
    int tex = m->texture;
    if( tex < 0 ) return ld3( m->color );
    TexP t = &sc.textures[ tex ];
    if( t->kind == ACN_TXM_PLAIN ) return ld3( t->color1 );
    NodeP o = &sc.nodes[ node ];
    double px = 0, py = 0;
    if( o->type == ACN_PLANE )
    {
        V3 p = v_sub( pos, ld3( o->pos ) );
        px = v_mlv( p, ld3( o->rax ) );
        py = v_mlv( p, ld3( o->rax + 3 ) );
    }
    else if( o->type == ACN_SPHERE )
    {
        V3 r = v_of_length( v_sub( pos, ld3( o->pos ) ), 1.0 );
        double x = v_mlv( r, ld3( o->rax ) );
        double y = v_mlv( r, v_mlx( ld3( o->rax + 6 ), ld3( o->rax ) ) );
        double z = v_mlv( r, ld3( o->rax + 6 ) );
        px = acn_atan2( x, y );
        z = z >  1.0 ?  1.0 : z;
        z = z < -1.0 ? -1.0 : z;
        py = acn_asin( z );
    }
    long long x = acn_llrint( px * t->scale );
    long long y = acn_llrint( py * t->scale );
    return ( ( x ^ y ) & 1 ) ? ld3( t->color1 ) : ld3( t->color2 );
}

/* vectors.h:372-384 */
DEV V3 cl_sat( V3 o, double gamma )
{
    double x = acn_pow( o.x, gamma );
    double y = acn_pow( o.y, gamma );
    double z = acn_pow( o.z, gamma );
    x = x > 0.0 ? x < 1.0 ? x : 1.0 : 0.0;
    y = y > 0.0 ? y < 1.0 ? y : 1.0 : 0.0;
    z = z > 0.0 ? z < 1.0 ? z : 1.0 : 0.0;
    return mk( x, y, z );
}

/* camera ray for a sample position: scene.c:980-990 */
template< class SC > DEV void camera_ray( const SC& sc, double monitor_x, double monitor_y, V3* rp, V3* rd )
{
    uint64_t width = sc.prm.image_width, height = sc.prm.image_height;
    double z = sc.unit_f * ( ( height >> 1 ) - monitor_y );
    double x = sc.unit_f * ( monitor_x - ( width >> 1 ) );
    V3 d = v_of_length( mk( x, sc.prm.camera_focal_length, z ), 1.0 );
    *rp = ld3( sc.prm.camera_position );
    *rd = m_mlv( sc.camera_rotation, d );
}

#endif /* ACN_DEVICE_H */
