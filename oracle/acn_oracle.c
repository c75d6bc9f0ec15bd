/* acn_oracle.c -- TEST INFRASTRUCTURE ONLY (see acn_oracle.h for scope and parity status).
 *
 * Plain recursive restatement of the reference's hot path.  Every function names the reference lines it
 * follows (paths relative to /root/reference).  Expressions keep the reference's evaluation order; build
 * with -ffp-contract=off.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "acn_oracle.h"

#ifdef ACN_ORACLE_LIBM
#define M_SIN( x )     sin( x )
#define M_COS( x )     cos( x )
#define M_TAN( x )     tan( x )
#define M_ACOS( x )    acos( x )
#define M_LOG( x )     log( x )
#define M_POW( x, y )  pow( x, y )
#define M_SQRT( x )    sqrt( x )
#define M_ATAN2( y, x ) atan2( y, x )
#define M_ASIN( x )    asin( x )
#define M_LLRINT( x )  llrint( x )
static double m_frexp_mant( double v ) { int e = 0; return frexp( v, &e ); }
int acn_oracle_math_mode( void ) { return 1; }
#else
#include "../actinon_amd/csrc/acn_detmath.h"
#define M_SIN( x )     acn_sin( x )
#define M_COS( x )     acn_cos( x )
#define M_TAN( x )     acn_tan( x )
#define M_ACOS( x )    acn_acos( x )
#define M_LOG( x )     acn_log( x )
#define M_POW( x, y )  acn_pow( x, y )
#define M_SQRT( x )    acn_sqrt( x )
#define M_ATAN2( y, x ) acn_atan2( y, x )
#define M_ASIN( x )    acn_asin( x )
#define M_LLRINT( x )  acn_llrint( x )
static double m_frexp_mant( double v ) { return acn_frexp_mant( v ); }
int acn_oracle_math_mode( void ) { return 0; }
#endif

/* src/vectors.h:30-33 */
#define F3_INF INFINITY
#define F3_MAG 1E+30
#define F3_EPS 1E-6
#define ORC_PI 3.14159265358979323846

typedef struct { double x, y, z; } v3;
typedef struct { v3 x, y, z; } m3;
typedef struct { v3 p, d; } ray_t;
typedef struct { ray_t ray; double cos_rs; } cone_t;
typedef struct { v3 exit_nor; int exit_obj; int enter_obj; } trans_t; /* compound.h:31-36; -1 = NULL */

typedef struct
{
    const acn_flat_scene* sc;
    uint64_t* cnt; /* nullable */
    /* ACN_SHARD_SAMPLES (include/actinon_hip.h): this evaluation contributes rank `rank` of `world`'s share -- of the
     * OUTERMOST sample loops (level == 0: not inside a path loop) the iterations [ n rank / world, n ( rank + 1 ) / world ),
     * of the terms under no such loop everything if rank == 0, nothing otherwise.  world <= 1: the whole thing. */
    uint32_t rank, world;
    int level;
} ctx_t;
static int shard_skips_terms( const ctx_t* c ) { return c->world > 1 && c->level == 0 && c->rank != 0; }
static int shard_skips_sample( const ctx_t* c, uint64_t j, uint64_t n )
{
    if( c->world <= 1 || c->level != 0 ) return 0;
    return !( j >= n * c->rank / c->world && j < n * ( c->rank + 1 ) / c->world );
}

#define COUNT( c, k ) do { if( ( c )->cnt ) ( c )->cnt[ k ]++; } while( 0 )
/* F_alg / T_alg of SURVEY.md App. B: event costs from actinon_amd/csrc/acn_costs.h, tallied at the same places of the
 * algorithm as by the instrumented device kernels */
#include "../actinon_amd/csrc/acn_costs.h"
#define COST( c, f, t ) do { if( ( c )->cnt ) { ( c )->cnt[ ORC_N_FLOP ] += ( f ); ( c )->cnt[ ORC_N_TRANSC ] += ( t ); } } while( 0 )

static double f3_sqr( double a ) { return a * a; }
static double f3_max( double a, double b ) { return a > b ? a : b; }
static double f3_min( double a, double b ) { return a < b ? a : b; }
static double f3_abs( double a ) { return a < 0 ? -a : a; }

/* ---- src/vectors.h:106-241 --------------------------------------------------------------------------------------- */
static v3 V( double x, double y, double z ) { v3 v = { x, y, z }; return v; }
static v3 v3_ld( const double* p ) { return V( p[ 0 ], p[ 1 ], p[ 2 ] ); }
static v3 v3_neg( v3 o ) { return V( -o.x, -o.y, -o.z ); }
static double v3_sqr( v3 o ) { return ( o.x * o.x ) + ( o.y * o.y ) + ( o.z * o.z ); }
static v3 v3_add( v3 o, v3 s ) { return V( o.x + s.x, o.y + s.y, o.z + s.z ); }
static v3 v3_sub( v3 o, v3 s ) { return V( o.x - s.x, o.y - s.y, o.z - s.z ); }
static v3 v3_mlf( v3 o, double f ) { return V( o.x * f, o.y * f, o.z * f ); }
static v3 v3_mlx( v3 o, v3 f ) { return V( o.y * f.z - o.z * f.y, o.z * f.x - o.x * f.z, o.x * f.y - o.y * f.x ); }
static v3 v3_mld( v3 o, v3 f ) { return V( o.x * f.x, o.y * f.y, o.z * f.z ); }
static double v3_mlv( v3 o, v3 m ) { return ( o.x * m.x ) + ( o.y * m.y ) + ( o.z * m.z ); }
static double v3_sub_mlv( v3 o, v3 s, v3 m ) { return ( ( o.x - s.x ) * m.x ) + ( ( o.y - s.y ) * m.y ) + ( ( o.z - s.z ) * m.z ); }
static double v3_diff_sqr( v3 o, v3 v ) { return f3_sqr( o.x - v.x ) + f3_sqr( o.y - v.y ) + f3_sqr( o.z - v.z ); }

/* vectors.h:148-154 */
static v3 v3_of_length( v3 o, double a )
{
    double r_sqr = v3_sqr( o );
    if( fabs( r_sqr - 1.0 ) < 1E-8 ) return o;
    double f = r_sqr > 0 ? ( a / M_SQRT( r_sqr ) ) : 0;
    return V( o.x * f, o.y * f, o.z * f );
}

/* vectors.h:157-162 */
static v3 v3_von( v3 o, v3 v )
{
    v3 o_n = v3_of_length( o, 1.0 );
    v = v3_sub( v, v3_mlf( o_n, v3_mlv( o_n, v ) ) );
    return v3_of_length( v, 1.0 );
}

/* vectors.h:165-175 */
static v3 v3_con( v3 o )
{
    double xx = o.x * o.x;
    double yy = o.y * o.y;
    double zz = o.z * o.z;
    v3 v;
    v.x = ( ( xx <= yy ) && ( xx <= zz ) ) ? 1 : 0;
    v.y = ( ( yy <= xx ) && ( yy <= zz ) ) ? 1 : 0;
    v.z = ( ( zz <= xx ) && ( zz <= yy ) ) ? 1 : 0;
    return v3_von( o, v );
}

/* beth bcore_lcg00/01/02_u3 -- constants declared in include/actinon_hip.h (parity unpinned) */
static uint64_t lcg00( uint64_t v ) { return v * ACN_LCG00_A + ACN_LCG00_C; }
static uint64_t lcg01( uint64_t v ) { return v * ACN_LCG01_A + ACN_LCG01_C; }
static uint64_t lcg02( uint64_t v ) { return v * ACN_LCG02_A + ACN_LCG02_C; }

/* vectors.h:45,48 */
static double f3_rnd0( uint64_t* rv ) { return ( double )( *rv = lcg00( *rv ) ) * ( 2.0 / 0xFFFFFFFFFFFFFFFFull ) - 1.0; }
static double f3_rnd1( uint64_t* rv ) { return ( double )( *rv = lcg00( *rv ) ) * ( 1.0 / 0xFFFFFFFFFFFFFFFFull ); }

/* vectors.h:177-182; the reference multiplies in s3_t and returns u3_t: two's-complement wrap */
static uint64_t seed_from_f3( double v )
{
    int64_t seed_s3 = ( int64_t )( m_frexp_mant( v ) * ( double )0x7FFFFFFFFFFFFFFF );
    return ( uint64_t )seed_s3 * 27362149ull;
}

/* vectors.h:185-190 */
static uint64_t v3_random_seed( v3 o, uint64_t rv )
{
    return seed_from_f3( o.x ) * lcg00( rv ) +
           seed_from_f3( o.y ) * lcg01( rv ) +
           seed_from_f3( o.z ) * lcg02( rv );
}

/* vectors.h:197-206 */
static v3 v3_random_sphere_cap( uint64_t* rv, double h )
{
    v3 v;
    double phi = 2.0 * ORC_PI * f3_rnd1( rv );
    v.z = 1.0 - f3_rnd1( rv ) * h;
    double scale = M_SQRT( 1.0 - v.z * v.z );
    v.x = M_SIN( phi ) * scale;
    v.y = M_COS( phi ) * scale;
    return v;
}

/* vectors.h:209-218 */
static v3 v3_random_sphere_belt( uint64_t* rv, double h )
{
    v3 v;
    double phi = 2.0 * ORC_PI * f3_rnd1( rv );
    v.z = f3_rnd0( rv ) * h;
    double scale = M_SQRT( 1.0 - v.z * v.z );
    v.x = M_SIN( phi ) * scale;
    v.y = M_COS( phi ) * scale;
    return v;
}

/* vectors.h:223-232 */
static v3 v3_orthogonal_projection( v3 o, v3 nor )
{
    double f = v3_mlv( o, nor );
    return V( o.x - nor.x * f, o.y - nor.y * f, o.z - nor.z * f );
}

/* vectors.h:238-241 */
static v3 v3_reflection( v3 dir, v3 nor )
{
    return v3_of_length( v3_sub( dir, v3_mlf( nor, 2.0 * v3_mlv( dir, nor ) ) ), 1.0 );
}

/* vectors.h:256-276 */
static v3 m3_mlv( const m3* o, v3 v )
{
    return V( o->x.x * v.x + o->x.y * v.y + o->x.z * v.z,
              o->y.x * v.x + o->y.y * v.y + o->y.z * v.z,
              o->z.x * v.x + o->z.y * v.y + o->z.z * v.z );
}

static v3 m3_tmlv( const m3* o, v3 v )
{
    return V( o->x.x * v.x + o->y.x * v.y + o->z.x * v.z,
              o->x.y * v.x + o->y.y * v.y + o->z.y * v.z,
              o->x.z * v.x + o->y.z * v.y + o->z.z * v.z );
}

/* vectors.h:309-322 */
static m3 m3_transposed( m3 o )
{
    m3 r = { { o.x.x, o.y.x, o.z.x }, { o.x.y, o.y.y, o.z.y }, { o.x.z, o.y.z, o.z.z } };
    return r;
}

static m3 m3_con_z( v3 v )
{
    m3 m;
    m.z = v3_of_length( v, 1.0 );
    m.x = v3_con( v );
    m.y = v3_mlx( m.z, m.x );
    return m;
}

static v3 ray_pos( const ray_t* o, double offs ) { return v3_add( o->p, v3_mlf( o->d, offs ) ); } /* vectors.h:343-346 */

static m3 node_rax( const acn_node* n )
{
    m3 m = { { n->rax[ 0 ], n->rax[ 1 ], n->rax[ 2 ] }, { n->rax[ 3 ], n->rax[ 4 ], n->rax[ 5 ] }, { n->rax[ 6 ], n->rax[ 7 ], n->rax[ 8 ] } };
    return m;
}

/* ---- src/gmath.h:38-97 ------------------------------------------------------------------------------------------- */
static double plane_ray_hit( v3 pos, v3 nor, const ray_t* ray, v3* p_nor )
{
    double div = v3_mlv( nor, ray->d );
    if( div == 0 ) return F3_INF;
    double offs = v3_sub_mlv( pos, ray->p, nor ) / div;
    if( p_nor ) *p_nor = nor;
    return ( offs > 0 ) ? offs - F3_EPS : F3_INF;
}

static int plane_observer_side( v3 pos, v3 nor, v3 observer )
{
    return v3_sub_mlv( observer, pos, nor ) > 0 ? 1 : -1;
}

static double sphere_ray_hit( v3 pos, double r, const ray_t* ray, v3* p_nor )
{
    v3 p = v3_sub( ray->p, pos );
    double s = v3_mlv( p, ray->d );
    double q = v3_sqr( p ) - ( r * r );

    double s2 = s * s;
    if( s2 < q ) return F3_INF;

    double offs = F3_INF;
    if( s < 0 && q > 0 )
    {
        offs = -s - M_SQRT( s2 - q ) - F3_EPS;
    }
    else if( s < 0 || q < 0 )
    {
        offs = -s + M_SQRT( s2 - q ) - F3_EPS;
    }

    if( offs < F3_INF && p_nor ) *p_nor = v3_of_length( v3_sub( ray_pos( ray, offs ), pos ), 1.0 );
    return offs;
}

static int sphere_observer_side( v3 pos, double r, v3 observer )
{
    v3 diff = v3_sub( observer, pos );
    return ( v3_sqr( diff ) > r * r ) ? 1 : -1;
}

/* ---- src/gmath.c:68-113 ------------------------------------------------------------------------------------------ */
static double fresnel_reflection( v3 dir_i, v3 exit_nor, double trix, v3* dir )
{
    double c = v3_mlv( dir_i, exit_nor );
    double f = c < 0 ? trix : 1.0 / trix;

    double cos_ai = fabs( c );
    cos_ai = cos_ai > 1.0 ? 1.0 : cos_ai;
    double sin_ai = M_SQRT( 1.0 - cos_ai * cos_ai );
    double sin_at = sin_ai * f;

    double reflectance = 1.0;

    if( sin_at < 1 )
    {
        double cos_at = M_SQRT( 1.0 - sin_at * sin_at );
        double rs = f3_sqr( ( f * cos_ai - cos_at ) / ( f * cos_ai + cos_at ) );
        double rp = f3_sqr( ( f * cos_at - cos_ai ) / ( f * cos_at + cos_ai ) );
        reflectance = ( rs + rp ) * 0.5;
    }

    if( dir ) *dir = v3_reflection( dir_i, exit_nor );
    return reflectance;
}

static void fresnel_refraction( v3 dir_i, v3 exit_nor, double trix, v3* dir )
{
    double c = v3_mlv( dir_i, exit_nor );
    double f = c < 0 ? trix : 1.0 / trix;
    double a = f;
    double q = f * f * ( 1.0 - c * c );
    if( q < 1.0 )
    {
        double b = -f * c + ( c > 0 ? M_SQRT( 1.0 - q ) : -M_SQRT( 1.0 - q ) );
        *dir = v3_add( v3_mlf( dir_i, a ), v3_mlf( exit_nor, b ) );
    }
    else
    {
        *dir = dir_i;
    }
}

/* ---- src/distance.c:39-42, 83-92 --------------------------------------------------------------------------------- */
static double sdf_eval( ctx_t* c, const acn_node* n, v3 pos )
{
    COUNT( c, ORC_N_SDF_EVAL );
    COST( c, n->sdf_kind == ACN_SDF_TORUS ? ACN_F_SDF_TORUS : ACN_F_SDF_SPHERE, 0 );
    if( n->sdf_kind == ACN_SDF_TORUS )
    {
        double x = pos.x;
        double y = pos.y;
        double f = M_SQRT( x * x + y * y );
        double f_inv = ( f > 0 ) ? ( 1.0 / f ) : 1.0;
        x *= f_inv;
        y *= f_inv;
        return M_SQRT( f3_sqr( x - pos.x ) + f3_sqr( y - pos.y ) + f3_sqr( pos.z ) ) - n->prm[ 1 ];
    }
    return M_SQRT( f3_sqr( pos.x ) + f3_sqr( pos.y ) + f3_sqr( pos.z ) ) - 1.0;
}

/* ---- envelopes: src/objects.c:90-103 ----------------------------------------------------------------------------- */
static int env_ray_hits( ctx_t* c, const acn_node* n, const ray_t* r )
{
    COUNT( c, ORC_N_ENV_TEST );
    int hit = sphere_ray_hit( v3_ld( n->env_pos ), n->env_radius, r, NULL ) < F3_INF;
    COST( c, hit ? ACN_F_ENV_HIT : ACN_F_ENV_MISS, 0 );
    return hit;
}

static int env_side( ctx_t* c, const acn_node* n, v3 pos )
{
    COUNT( c, ORC_N_ENV_TEST );
    COST( c, ACN_F_SIDE_SPHERE, 0 );
    return sphere_observer_side( v3_ld( n->env_pos ), n->env_radius, pos );
}

static int has_env( const acn_node* n ) { return ( n->flags & ACN_NODE_HAS_ENVELOPE ) != 0; }

static double obj_ray_hit( ctx_t* c, int node, const ray_t* ray, v3* p_nor );
static int    obj_side( ctx_t* c, int node, v3 pos );

/* ---- per-type ray_hit -------------------------------------------------------------------------------------------- */

/* objects.c:778-821 */
static double squaroid_ray_hit( const acn_node* o, const ray_t* r, v3* p_nor )
{
    m3 rax = node_rax( o );
    double oa = o->prm[ 0 ], ob = o->prm[ 1 ], oc = o->prm[ 2 ], orr = o->prm[ 3 ];
    v3 p = m3_mlv( &rax, v3_sub( r->p, v3_ld( o->pos ) ) );
    v3 d = m3_mlv( &rax, r->d );

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
        rr = M_SQRT( rr );
        a = -s - rr;
        if( a < 0 ) a = -s + rr;
        if( a < 0 ) a = F3_INF;
    }
    else
    {
        a = ( fq != 0 ) ? -fs / ( 2 * fq ) : F3_INF;
    }

    if( a == F3_INF ) return F3_INF;

    if( p_nor )
    {
        double x = p.x + a * d.x;
        double y = p.y + a * d.y;
        double z = p.z + a * d.z;
        v3 n1;
        n1.x = x * oa;
        n1.y = y * ob;
        n1.z = z * oc;
        *p_nor = v3_of_length( m3_tmlv( &rax, n1 ), 1.0 );
    }
    return a - F3_EPS;
}

/* objects.c:823-827 */
static int squaroid_side( const acn_node* o, v3 pos )
{
    m3 rax = node_rax( o );
    v3 p = m3_mlv( &rax, v3_sub( pos, v3_ld( o->pos ) ) );
    return ( o->prm[ 0 ] * p.x * p.x + o->prm[ 1 ] * p.y * p.y + o->prm[ 2 ] * p.z * p.z + o->prm[ 3 ] ) > 0 ? 1 : -1;
}

/* objects.c:903-959 */
static double distance_ray_hit( ctx_t* c, const acn_node* o, const ray_t* r, v3* p_nor )
{
    COUNT( c, ORC_N_SDF_RAY );
    m3 rax = node_rax( o );
    double inv_scale = o->prm[ 0 ];
    ray_t ray = *r;
    double offs0 = 0;
    if( has_env( o ) )
    {
        if( env_side( c, o, r->p ) == 1 )
        {
            offs0 = sphere_ray_hit( v3_ld( o->env_pos ), o->env_radius, &ray, NULL );
            COST( c, offs0 < F3_INF ? ACN_F_SPHERE_HIT : ACN_F_SPHERE_MISS, 0 );
            if( offs0 >= F3_INF ) return F3_INF;
            ray.p = ray_pos( &ray, offs0 );
        }
    }

    ray.p = v3_mlf( m3_mlv( &rax, v3_sub( ray.p, v3_ld( o->pos ) ) ), inv_scale );
    ray.d = m3_mlv( &rax, ray.d );

    double offs1 = 0;
    double dist = sdf_eval( c, o, ray.p );
    unsigned steps = 0;

    if( dist > 0 )
    {
        for( int64_t i = 0; i < o->cycles; i++ )
        {
            offs1 += dist + F3_EPS;
            dist = sdf_eval( c, o, ray_pos( &ray, offs1 ) );
            steps++;
            if( dist < 0 || dist > F3_MAG ) break;
        }
    }
    else
    {
        for( int64_t i = 0; i < o->cycles; i++ )
        {
            offs1 -= dist - F3_EPS;
            dist = sdf_eval( c, o, ray_pos( &ray, offs1 ) );
            steps++;
            if( dist > 0 || dist < -F3_MAG ) break;
        }
    }
    COST( c, ACN_F_SDF_RAY + ACN_F_SDF_STEP * steps, 0 );

    if( f3_abs( dist ) <= F3_EPS )
    {
        if( p_nor )
        {
            COST( c, ACN_F_SDF_NORMAL, 0 );
            v3 p = ray_pos( &ray, offs1 );
            double d0 = sdf_eval( c, o, p );
            v3 n;
            n.x = ( sdf_eval( c, o, V( p.x + F3_EPS, p.y, p.z ) ) - d0 ) / F3_EPS;
            n.y = ( sdf_eval( c, o, V( p.x, p.y + F3_EPS, p.z ) ) - d0 ) / F3_EPS;
            n.z = ( sdf_eval( c, o, V( p.x, p.y, p.z + F3_EPS ) ) - d0 ) / F3_EPS;
            *p_nor = v3_of_length( m3_tmlv( &rax, n ), 1.0 );
        }
        return offs0 + ( offs1 / inv_scale ) - F3_EPS;
    }
    return F3_INF;
}

/* objects.c:961-966 */
static int distance_side( ctx_t* c, const acn_node* o, v3 pos )
{
    COST( c, ACN_F_SIDE_SDF, 0 );
    if( has_env( o ) && env_side( c, o, pos ) == 1 ) return 1;
    m3 rax = node_rax( o );
    v3 p = v3_mlf( m3_mlv( &rax, v3_sub( pos, v3_ld( o->pos ) ) ), o->prm[ 0 ] );
    return sdf_eval( c, o, p ) > 0 ? 1 : -1;
}

/* objects.c:1052-1094 (want = -1) and :1209-1251 (want = +1) */
static double pair_ray_hit( ctx_t* c, const acn_node* o, const ray_t* r, v3* p_nor, int want )
{
    COUNT( c, ORC_N_PAIR_HIT );
    v3 n1 = { 0, 0, 0 }, n2 = { 0, 0, 0 };
    double a1 = obj_ray_hit( c, o->child0, r, &n1 );
    double a2 = obj_ray_hit( c, o->child1, r, &n2 );
    COST( c, ACN_F_PAIR_STEP, 0 );
    if( a1 < a2 && obj_side( c, o->child1, ray_pos( r, a1 ) ) == want )
    {
        if( p_nor ) *p_nor = n1;
        return a1;
    }

    if( a2 >= F3_INF ) return F3_INF;
    COST( c, ACN_F_PAIR_STEP, 0 );

    if( obj_side( c, o->child0, ray_pos( r, a2 ) ) == want )
    {
        if( p_nor ) *p_nor = n2;
        return a2;
    }

    double offs = a2;
    ray_t ray;
    ray.d = r->d;
    ray.p = ray_pos( r, offs );
    int obj1 = o->child0;
    int obj2 = o->child1;

    while( offs < F3_INF )
    {
        double a = obj_ray_hit( c, obj1, &ray, &n1 );
        COST( c, ACN_F_PAIR_STEP, 0 );
        if( a >= F3_INF ) return F3_INF;
        if( obj_side( c, obj2, ray_pos( &ray, a ) ) == want )
        {
            if( p_nor ) *p_nor = n1;
            return offs + a;
        }
        offs += a + 2 * F3_EPS;
        ray.p = ray_pos( r, offs );
        int tmp = obj2;
        obj2 = obj1;
        obj1 = tmp;
    }
    return F3_INF;
}

/* objects.c:1418-1437 */
static double scale_ray_hit( ctx_t* c, const acn_node* o, const ray_t* r, v3* p_nor )
{
    COST( c, ACN_F_SCALE_WRAP, 0 );
    m3 rax = node_rax( o );
    v3 inv_scale = V( o->prm[ 0 ], o->prm[ 1 ], o->prm[ 2 ] );
    ray_t ray;
    ray.p = v3_mld( m3_mlv( &rax, v3_sub( r->p, v3_ld( o->pos ) ) ), inv_scale );
    ray.d = v3_mld( m3_mlv( &rax, r->d ), inv_scale );

    double d_length = M_SQRT( v3_sqr( ray.d ) );
    double d_factor = ( d_length > 0 ) ? ( 1.0 / d_length ) : 0;
    ray.d = v3_mlf( ray.d, d_factor );

    v3 n1 = { 0, 0, 0 };
    double a1 = obj_ray_hit( c, o->child0, &ray, &n1 ) + F3_EPS;
    if( a1 < F3_INF )
    {
        n1 = v3_mld( n1, inv_scale );
        if( p_nor ) *p_nor = v3_of_length( m3_tmlv( &rax, n1 ), 1.0 );
        return a1 * d_factor - F3_EPS;
    }
    return F3_INF;
}

/* vtable dispatch fp_ray_hit (objects.c:212-225) */
static double type_ray_hit( ctx_t* c, const acn_node* o, const ray_t* ray, v3* p_nor )
{
    switch( o->type )
    {
        case ACN_PLANE:    COUNT( c, ORC_N_PLANE_HIT ); COST( c, ACN_F_PLANE_HIT, 0 ); return plane_ray_hit( v3_ld( o->pos ), v3_ld( o->rax + 6 ), ray, p_nor ); /* objects.c:529-532 */
        case ACN_SPHERE:   /* objects.c:649-652 */
        {
            COUNT( c, ORC_N_SPHERE_HIT );
            double a = sphere_ray_hit( v3_ld( o->pos ), o->prm[ 0 ], ray, p_nor );
            COST( c, a < F3_INF ? ( p_nor ? ACN_F_SPHERE_HIT_NOR : ACN_F_SPHERE_HIT ) : ACN_F_SPHERE_MISS, 0 );
            return a;
        }
        case ACN_SQUAROID:
        {
            COUNT( c, ORC_N_SQUAROID_HIT );
            double a = squaroid_ray_hit( o, ray, p_nor );
            COST( c, a < F3_INF ? ( p_nor ? ACN_F_SQUAROID_HIT_NOR : ACN_F_SQUAROID_HIT ) : ACN_F_SQUAROID_MISS, 0 );
            return a;
        }
        case ACN_DISTANCE: return distance_ray_hit( c, o, ray, p_nor );
        case ACN_PAIR_INSIDE:  return pair_ray_hit( c, o, ray, p_nor, -1 );
        case ACN_PAIR_OUTSIDE: return pair_ray_hit( c, o, ray, p_nor,  1 );
        case ACN_NEG: /* objects.c:1329-1339 */
        {
            v3 n1 = { 0, 0, 0 };
            double a1 = obj_ray_hit( c, o->child0, ray, &n1 );
            if( a1 < F3_INF )
            {
                if( p_nor ) *p_nor = v3_neg( n1 );
                return a1;
            }
            return F3_INF;
        }
        case ACN_SCALE: return scale_ray_hit( c, o, ray, p_nor );
        default: return F3_INF;
    }
}

/* objects.c:261-284 */
static double obj_ray_hit( ctx_t* c, int node, const ray_t* ray, v3* p_nor )
{
    const acn_node* hdr = &c->sc->nodes[ node ];
    COUNT( c, ORC_N_OBJ_HIT );
    COUNT( c, ORC_N_NODE_VISIT );
    if( has_env( hdr ) && !env_ray_hits( c, hdr, ray ) ) return F3_INF;
    double a = type_ray_hit( c, hdr, ray, p_nor );
    if( a < F3_INF && hdr->surface_roughness > 0 && p_nor )
    {
        COST( c, ACN_F_ROUGHNESS, ACN_T_ROUGHNESS );
        v3 n = *p_nor;
        uint64_t rv = v3_random_seed( ray_pos( ray, a ), 1246 );
        double f;

        f = f3_rnd0( &rv ) * 0.99;
        n.x += hdr->surface_roughness * M_LOG( ( 1.0 - f ) / ( 1.0 + f ) );

        f = f3_rnd0( &rv ) * 0.99;
        n.y += hdr->surface_roughness * M_LOG( ( 1.0 - f ) / ( 1.0 + f ) );

        f = f3_rnd0( &rv ) * 0.99;
        n.z += hdr->surface_roughness * M_LOG( ( 1.0 - f ) / ( 1.0 + f ) );

        *p_nor = v3_of_length( n, 1.0 );
    }
    return a;
}

/* objects.c:365-370 + per-type side */
static int obj_side( ctx_t* c, int node, v3 pos )
{
    const acn_node* o = &c->sc->nodes[ node ];
    COUNT( c, ORC_N_SIDE );
    if( has_env( o ) && env_side( c, o, pos ) == 1 ) return 1;
    switch( o->type )
    {
        case ACN_PLANE:    COST( c, ACN_F_SIDE_PLANE, 0 );    return plane_observer_side( v3_ld( o->pos ), v3_ld( o->rax + 6 ), pos );   /* objects.c:534-537 */
        case ACN_SPHERE:   COST( c, ACN_F_SIDE_SPHERE, 0 );   return sphere_observer_side( v3_ld( o->pos ), o->prm[ 0 ], pos );          /* objects.c:654-657 */
        case ACN_SQUAROID: COST( c, ACN_F_SIDE_SQUAROID, 0 ); return squaroid_side( o, pos );
        case ACN_DISTANCE: return distance_side( c, o, pos );
        case ACN_PAIR_INSIDE:  /* objects.c:1096-1099 */
            return ( obj_side( c, o->child0, pos ) + obj_side( c, o->child1, pos ) == -2 ) ? -1 : 1;
        case ACN_PAIR_OUTSIDE: /* objects.c:1253-1256 */
            return ( obj_side( c, o->child0, pos ) + obj_side( c, o->child1, pos ) == 2 ) ? 1 : -1;
        case ACN_NEG:          /* objects.c:1341-1344 */
            return -1 * obj_side( c, o->child0, pos );
        case ACN_SCALE:        /* objects.c:1439-1443 */
        {
            COST( c, ACN_F_SIDE_SCALE, 0 );
            m3 rax = node_rax( o );
            v3 p = m3_mlv( &rax, v3_sub( pos, v3_ld( o->pos ) ) );
            return obj_side( c, o->child0, v3_mld( p, V( o->prm[ 0 ], o->prm[ 1 ], o->prm[ 2 ] ) ) );
        }
        default: return 1;
    }
}

/* envelope_s_fov objects.c:66-84 == obj_sphere_s_fov objects.c:619-637 */
static cone_t sphere_fov( v3 center, double radius, v3 pos )
{
    cone_t cne;
    v3 diff = v3_sub( center, pos );
    cne.ray.d = v3_of_length( diff, 1.0 );
    cne.ray.p = pos;
    double diff_sqr = v3_sqr( diff );
    double radius_sqr = f3_sqr( radius );
    if( diff_sqr > radius_sqr )
    {
        cne.cos_rs = M_SQRT( 1.0 - ( radius_sqr / diff_sqr ) );
    }
    else
    {
        cne.cos_rs = -1;
    }
    return cne;
}

/* obj_fov objects.c:254-259 -> per-type fov (plane :520-527, sphere :619-637, pairs :1035-1045, :1192-1202) */
static cone_t obj_fov( const acn_node* o, v3 pos )
{
    cone_t cne;
    switch( o->type )
    {
        case ACN_PLANE:
            cne.ray.p = pos;
            cne.ray.d = v3_neg( v3_ld( o->rax + 6 ) );
            cne.cos_rs = v3_mlv( v3_sub( v3_ld( o->pos ), pos ), cne.ray.d ) > 0 ? 0 : 1;
            return cne;
        case ACN_SPHERE:
            return sphere_fov( v3_ld( o->pos ), o->prm[ 0 ], pos );
        default: /* pairs; other types are rejected at validation (ACN_ERR_NO_FOV) */
            if( has_env( o ) ) return sphere_fov( v3_ld( o->env_pos ), o->env_radius, pos );
            cne.ray.d = v3_of_length( v3_sub( v3_ld( o->pos ), pos ), 1.0 );
            cne.ray.p = pos;
            cne.cos_rs = 0;
            return cne;
    }
}

/* obj_projection: plane objects.c:514-518, sphere :602-617, distance :893-896 (other types have none and are
 * rejected at validation when they carry a chess texture) */
static void obj_projection( const acn_node* o, v3 pos, double* px, double* py )
{
    if( o->type == ACN_PLANE )
    {
        v3 p = v3_sub( pos, v3_ld( o->pos ) );
        *px = v3_mlv( p, v3_ld( o->rax ) );
        *py = v3_mlv( p, v3_ld( o->rax + 3 ) );
    }
    else if( o->type == ACN_SPHERE )
    {
        v3 r = v3_of_length( v3_sub( pos, v3_ld( o->pos ) ), 1.0 );
        double x = v3_mlv( r, v3_ld( o->rax ) );
        double y = v3_mlv( r, v3_mlx( v3_ld( o->rax + 6 ), v3_ld( o->rax ) ) );
        double z = v3_mlv( r, v3_ld( o->rax + 6 ) );
        double azimuth = M_ATAN2( x, y );
        z = z >  1.0 ?  1.0 : z;
        z = z < -1.0 ? -1.0 : z;
        *px = azimuth;
        *py = M_ASIN( z );
    }
    else
    {
        *px = 0; *py = 0;
    }
}

/* obj_color objects.c:411-422 with txm_plain_s_clr / txm_chess_s_clr textures.c:99-102, 142-148 */
static v3 obj_color( const acn_flat_scene* sc, const acn_node* o, v3 pos )
{
    if( o->texture < 0 ) return v3_ld( o->color );
    const acn_texture* t = &sc->textures[ o->texture ];
    if( t->kind == ACN_TXM_PLAIN ) return v3_ld( t->color1 );
    double px, py;
    obj_projection( o, pos, &px, &py );
    int64_t x = M_LLRINT( px * t->scale );
    int64_t y = M_LLRINT( py * t->scale );
    return ( ( x ^ y ) & 1 ) ? v3_ld( t->color1 ) : v3_ld( t->color2 );
}

/* ---- src/compound.c:215-299 -------------------------------------------------------------------------------------- */
static double compound_ray_hit( ctx_t* c, int cmp, const ray_t* ray, v3* p_nor, int* hit_obj )
{
    const acn_node* o = &c->sc->nodes[ cmp ];
    COUNT( c, ORC_N_NODE_VISIT );
    if( has_env( o ) && !env_ray_hits( c, o, ray ) ) return F3_INF;
    v3 nor = { 0, 0, 0 };
    double min_a = F3_INF;
    for( int i = 0; i < o->child1; i++ )
    {
        int element = c->sc->elems[ o->child0 + i ];
        int hit_obj_l = -1;
        double a = F3_INF;
        if( c->sc->nodes[ element ].type == ACN_COMPOUND )
        {
            a = compound_ray_hit( c, element, ray, &nor, &hit_obj_l );
        }
        else
        {
            hit_obj_l = element;
            a = obj_ray_hit( c, hit_obj_l, ray, &nor );
        }

        if( a < min_a )
        {
            min_a = a;
            if( p_nor ) *p_nor = nor;
            if( hit_obj ) *hit_obj = hit_obj_l;
        }
    }
    return min_a;
}

static double compound_ray_trans_hit( ctx_t* c, int cmp, const ray_t* ray, trans_t* trans )
{
    const acn_node* o = &c->sc->nodes[ cmp ];
    COUNT( c, ORC_N_TRANS_RAY );
    COUNT( c, ORC_N_NODE_VISIT );
    if( has_env( o ) && !env_ray_hits( c, o, ray ) ) return F3_INF;
    v3 nor = { 0, 0, 0 };
    double min_a = F3_INF;
    for( int i = 0; i < o->child1; i++ )
    {
        int element = c->sc->elems[ o->child0 + i ];
        int hit_obj = -1;
        double a = F3_INF;
        if( c->sc->nodes[ element ].type == ACN_COMPOUND )
        {
            a = compound_ray_hit( c, element, ray, &nor, &hit_obj );
        }
        else
        {
            hit_obj = element;
            a = obj_ray_hit( c, hit_obj, ray, &nor );
        }

        if( a < F3_INF )
        {
            COST( c, ACN_F_TRANS_RESOLVE, 0 );
            if( a < min_a - F3_EPS )
            {
                min_a = a;
                if( v3_mlv( nor, ray->d ) > 0 )
                {
                    trans->exit_nor = nor;
                    trans->exit_obj = hit_obj;
                    trans->enter_obj = -1;
                }
                else
                {
                    trans->exit_nor = v3_neg( nor );
                    trans->exit_obj = -1;
                    trans->enter_obj = hit_obj;
                }
            }
            else if( f3_abs( a - min_a ) < F3_EPS )
            {
                min_a = a < min_a ? a : min_a;
                if( v3_mlv( nor, ray->d ) > 0 )
                {
                    trans->exit_obj = hit_obj;
                }
                else
                {
                    trans->enter_obj = hit_obj;
                }
            }
        }
    }
    return min_a;
}

/* ---- src/scene.c:362-382 ----------------------------------------------------------------------------------------- */
static double scene_trans_hit( ctx_t* c, const ray_t* r, trans_t* trans )
{
    double min_a = F3_INF;
    double a;
    trans_t trans_l = { { 0, 0, 0 }, -1, -1 };

    if( ( a = compound_ray_trans_hit( c, c->sc->light_root, r, &trans_l ) ) < min_a )
    {
        min_a = a;
        *trans = trans_l;
    }

    if( ( a = compound_ray_trans_hit( c, c->sc->matter_root, r, &trans_l ) ) < min_a )
    {
        min_a = a;
        *trans = trans_l;
    }
    return min_a;
}

/* ---- src/scene.c:394-416 ----------------------------------------------------------------------------------------- */
static double oren_nayar_weight( ctx_t* c, double weight, double theta_i, double on_a, double on_b, v3 out_d, v3 nor, v3 ray_prj )
{
    COUNT( c, ORC_N_OREN_NAYAR );
    COST( c, ACN_F_OREN_NAYAR, ACN_T_OREN_NAYAR );
    double theta_r = M_ACOS( weight );
    double cos_phi = -v3_mlv( v3_of_length( v3_orthogonal_projection( out_d, nor ), 1.0 ), ray_prj );
    return weight *
    (
        on_a +
        (
            on_b *
            f3_max( cos_phi, 0 ) *
            M_SIN( f3_max( theta_i, theta_r ) ) *
            M_TAN( f3_min( theta_i, theta_r ) )
        )
    );
}

/* ---- src/scene.c:420-667 ----------------------------------------------------------------------------------------- */
static v3 scene_lum( ctx_t* c, const ray_t* ray, double offs, trans_t* trans, uint64_t depth, double intensity )
{
    const acn_params* scene = &c->sc->params;
    const acn_node* nodes = c->sc->nodes;
    v3 bg = v3_ld( scene->background_color );
    v3 lum = { 0, 0, 0 };
    if( depth == 0 || intensity < scene->trace_min_intensity ) return lum;
    COUNT( c, ORC_N_LUM );
    COST( c, ACN_F_LUM_FIXED, 0 );

    v3 pos = ray_pos( ray, offs );
    const acn_node* enter_obj = trans->enter_obj >= 0 ? &nodes[ trans->enter_obj ] : NULL;
    const acn_node* exit_obj  = trans->exit_obj  >= 0 ? &nodes[ trans->exit_obj  ] : NULL;

    if( enter_obj && enter_obj->radiance > 0 )
    {
        double diff_sqr = v3_diff_sqr( pos, v3_ld( enter_obj->pos ) );
        double light_intensity = ( diff_sqr > 0 ) ? ( enter_obj->radiance / diff_sqr ) : F3_MAG;
        COST( c, ACN_F_EMISSION, 0 );
        if( shard_skips_terms( c ) ) return lum;
        return v3_mlf( obj_color( c->sc, enter_obj, pos ), light_intensity * intensity );
    }

    double trans_refractive_index = 1.0;
    double fresnel_reflectivity = 0;
    double chromatic_reflectivity = 0;
    double diffuse_reflectivity = 0;
    double on_a = 1.0;
    double on_b = 0.0;
    int transparent = 0;

    if( enter_obj )
    {
        trans_refractive_index = enter_obj->refractive_index;
        fresnel_reflectivity   = enter_obj->fresnel_reflectivity && enter_obj->refractive_index != 1.0;
        chromatic_reflectivity = enter_obj->chromatic_reflectivity;
        diffuse_reflectivity   = enter_obj->diffuse_reflectivity;
        transparent            = v3_sqr( v3_ld( enter_obj->transparency ) ) > 0;
        double sigma           = enter_obj->sigma;
        if( sigma > 0 )
        {
            double sigma_sqr = f3_sqr( sigma );
            on_a = 1.0 - 0.5 * sigma_sqr / ( sigma_sqr + 0.33 );
            on_b = 0.45 * sigma_sqr / ( sigma_sqr + 0.09 );
        }
    }

    if( exit_obj )
    {
        trans_refractive_index /= exit_obj->refractive_index;
        fresnel_reflectivity = 1.0;
        diffuse_reflectivity = chromatic_reflectivity = 0;
        transparent = 1;
    }

    /* obj_color( trans->enter_obj, pos ) (objects.c:411-422; texture fields unsupported). The reference
     * dereferences a NULL enter_obj here when trace_min_intensity == 0; the restatement uses white. */
    v3 enter_color = enter_obj ? obj_color( c->sc, enter_obj, pos ) : V( 1, 1, 1 );

    /* fresnel reflection :473-495 */
    if( fresnel_reflectivity > 0 && intensity >= scene->trace_min_intensity )
    {
        COUNT( c, ORC_N_FRESNEL );
        COST( c, ACN_F_FRESNEL_REFL, 0 );
        ray_t out;
        out.p = pos;
        double reflectance = fresnel_reflection( ray->d, trans->exit_nor, trans_refractive_index, &out.d ) * fresnel_reflectivity;

        trans_t trans_l = { { 0, 0, 0 }, -1, -1 };
        double a;
        v3 lum_l = { 0, 0, 0 };
        if( ( a = scene_trans_hit( c, &out, &trans_l ) ) < F3_INF )
        {
            lum_l = scene_lum( c, &out, a, &trans_l, depth - 1, reflectance * intensity );
        }
        else if( !shard_skips_terms( c ) )
        {
            lum_l = v3_mlf( bg, reflectance * intensity );
        }
        lum = v3_add( lum, lum_l );
        intensity *= ( 1.0 - reflectance );
    }

    /* chromatic reflection :498-523 */
    if( chromatic_reflectivity > 0 && intensity >= scene->trace_min_intensity )
    {
        ray_t out;
        out.p = pos;
        COST( c, ACN_F_REFLECTION, 0 );
        out.d = v3_reflection( ray->d, trans->exit_nor );
        trans_t trans_l = { { 0, 0, 0 }, -1, -1 };
        double a;
        v3 lum_l = { 0, 0, 0 };
        if( ( a = scene_trans_hit( c, &out, &trans_l ) ) < F3_INF )
        {
            lum_l = scene_lum( c, &out, a, &trans_l, depth - 1, chromatic_reflectivity * intensity );
        }
        else if( !shard_skips_terms( c ) )
        {
            lum_l = v3_mlf( bg, chromatic_reflectivity * intensity );
        }
        lum_l.x *= enter_color.x;
        lum_l.y *= enter_color.y;
        lum_l.z *= enter_color.z;
        lum = v3_add( lum, lum_l );
        intensity *= ( 1.0 - chromatic_reflectivity );
    }

    /* diffuse reflection :526-630 */
    if( intensity * diffuse_reflectivity >= scene->trace_min_intensity )
    {
        double diffuse_intensity = intensity * diffuse_reflectivity;
        ray_t surface = { pos, v3_neg( trans->exit_nor ) };
        COST( c, ACN_F_SHADE_DIFFUSE + ACN_F_SEED, ACN_T_SHADE_DIFFUSE + ACN_T_SEED );

        double theta_i = M_ACOS( -v3_mlv( ray->d, surface.d ) );
        v3 ray_projection = v3_of_length( v3_orthogonal_projection( ray->d, surface.d ), 1.0 );

        uint64_t rv = v3_random_seed( surface.p, 3294479285ull ) + v3_random_seed( surface.d, 3247146734ull );

        v3 lum_l = { 0, 0, 0 };

        const acn_node* light = &nodes[ c->sc->light_root ];
        for( int i = 0; i < light->child1; i++ )
        {
            v3 cl_sum = { 0, 0, 0 };
            ray_t out = surface;
            int light_idx = c->sc->elems[ light->child0 + i ];
            const acn_node* light_src = &nodes[ light_idx ];
            COST( c, ACN_F_FOV + ACN_F_FRAME, 0 );
            cone_t fov_to_src = obj_fov( light_src, pos );
            m3 src_con = m3_transposed( m3_con_z( fov_to_src.ray.d ) );
            double cyl_hgt = 1 - fov_to_src.cos_rs; /* areal_coverage vectors.h:362 */
            v3 color = obj_color( c->sc, light_src, v3_ld( light_src->pos ) );
            uint64_t direct_samples = ( uint64_t )( scene->direct_samples * diffuse_intensity );
            direct_samples = ( direct_samples == 0 ) ? 1 : direct_samples;

            for( uint64_t j = 0; j < direct_samples; j++ )
            {
                COUNT( c, ORC_N_CAP_SAMPLE );
                COST( c, ACN_F_CAP_SAMPLE, ACN_T_CAP_SAMPLE );
                out.d = m3_mlv( &src_con, v3_random_sphere_cap( &rv, cyl_hgt ) );
                if( shard_skips_sample( c, j, direct_samples ) ) continue;   /* the draws were made: the stream stays in step */
                double weight = v3_mlv( out.d, surface.d );
                if( weight <= 0 ) continue;

                double a = obj_ray_hit( c, light_idx, &out, NULL );
                if( a >= F3_INF ) continue;

                if( on_b > 0 ) weight = oren_nayar_weight( c, weight, theta_i, on_a, on_b, out.d, surface.d, ray_projection );

                COUNT( c, ORC_N_SHADOW_RAY );
                if( compound_ray_hit( c, c->sc->matter_root, &out, NULL, NULL ) > a )
                {
                    COST( c, ACN_F_DIRECT_TAIL, 0 );
                    v3 hit_pos = ray_pos( &out, a );
                    double diff_sqr = v3_diff_sqr( hit_pos, v3_ld( light_src->pos ) );
                    double local_intensity = ( diff_sqr > 0 ) ? ( light_src->radiance / diff_sqr ) : F3_MAG;
                    cl_sum = v3_add( cl_sum, v3_mlf( color, local_intensity * weight * diffuse_intensity ) );
                }
            }
            lum_l = v3_add( lum_l, v3_mlf( cl_sum, 2.0 * cyl_hgt / direct_samples ) );
        }

        /* path tracing :584-621 */
        if( scene->path_samples && depth > 10 )
        {
            v3 cl_sum = { 0, 0, 0 };
            ray_t out = surface;
            COST( c, ACN_F_FRAME, 0 );
            m3 out_con = m3_transposed( m3_con_z( surface.d ) );

            uint64_t path_samples = ( uint64_t )( scene->path_samples * diffuse_intensity );
            path_samples = ( path_samples == 0 ) ? 1 : path_samples;

            for( uint64_t i = 0; i < path_samples; i++ )
            {
                COUNT( c, ORC_N_CAP_SAMPLE );
                COST( c, ACN_F_CAP_SAMPLE, ACN_T_CAP_SAMPLE );
                out.d = m3_mlv( &out_con, v3_random_sphere_cap( &rv, 1.0 ) );
                if( shard_skips_sample( c, i, path_samples ) ) continue;
                double weight = v3_mlv( out.d, surface.d );
                if( weight <= 0 ) continue;
                COST( c, ACN_F_PATH_TAIL, 0 );

                if( on_b > 0 ) weight = oren_nayar_weight( c, weight, theta_i, on_a, on_b, out.d, surface.d, ray_projection );

                trans_t trans_l = { { 0, 0, 0 }, -1, -1 };
                double a = compound_ray_trans_hit( c, c->sc->matter_root, &out, &trans_l );

                if( a < scene->max_path_length )
                {
                    c->level++;
                    v3 lum_c = scene_lum( c, &out, a, &trans_l, depth - 10, weight * diffuse_intensity );
                    c->level--;
                    cl_sum = v3_add( cl_sum, lum_c );
                }
                else
                {
                    cl_sum = v3_add( cl_sum, v3_mlf( bg, weight * diffuse_intensity ) );
                }
            }
            lum_l = v3_add( lum_l, v3_mlf( cl_sum, 2.0 / path_samples ) );
        }

        lum_l.x *= enter_color.x;
        lum_l.y *= enter_color.y;
        lum_l.z *= enter_color.z;
        lum = v3_add( lum, lum_l );

        intensity *= ( 1.0 - diffuse_reflectivity );
    }

    /* refraction :633-653 */
    if( transparent && intensity >= scene->trace_min_intensity )
    {
        ray_t out;
        out.p = ray_pos( ray, offs + 2.0 * F3_EPS );
        COST( c, ACN_F_FRESNEL_REFR, 0 );
        fresnel_refraction( ray->d, trans->exit_nor, trans_refractive_index, &out.d );

        trans_t trans_l = { { 0, 0, 0 }, -1, -1 };
        double a;
        v3 lum_l = { 0, 0, 0 };
        if( ( a = scene_trans_hit( c, &out, &trans_l ) ) < F3_INF )
        {
            lum_l = scene_lum( c, &out, a, &trans_l, depth - 1, intensity );
        }
        else if( !shard_skips_terms( c ) )
        {
            lum_l = v3_mlf( bg, intensity );
        }
        lum = v3_add( lum, lum_l );
    }

    /* exiting object :656-664 */
    if( exit_obj )
    {
        if( offs > 0 ) COST( c, ACN_F_ABSORB, ACN_T_ABSORB );
        double rf = offs > 0 ? M_POW( exit_obj->transparency[ 0 ], offs ) : 1.0;
        double gf = offs > 0 ? M_POW( exit_obj->transparency[ 1 ], offs ) : 1.0;
        double bf = offs > 0 ? M_POW( exit_obj->transparency[ 2 ], offs ) : 1.0;
        lum.x *= rf;
        lum.y *= gf;
        lum.z *= bf;
    }

    return lum;
}

/* vectors.h:372-384 */
static v3 cl_sat( v3 o, double gamma )
{
    double x = M_POW( o.x, gamma );
    double y = M_POW( o.y, gamma );
    double z = M_POW( o.z, gamma );
    x = x > 0.0 ? x < 1.0 ? x : 1.0 : 0.0;
    y = y > 0.0 ? y < 1.0 ? y : 1.0 : 0.0;
    z = z > 0.0 ? z < 1.0 ? z : 1.0 : 0.0;
    return V( x, y, z );
}

/* ---- src/scene.c:956-1013: one sample position ------------------------------------------------------------------- */
typedef struct { m3 camera_rotation; double unit_f; } camera_t;

static camera_t camera_setup( const acn_params* s )
{
    camera_t cam;
    uint64_t unit_sz = ( s->image_height >> 1 );
    cam.unit_f = 1.0 / unit_sz;
    v3 ry = v3_of_length( v3_ld( s->camera_view_direction ), 1 );
    v3 rz = v3_of_length( v3_ld( s->camera_top_direction ), 1 );
    rz = v3_von( ry, rz );
    v3 rx = v3_mlx( ry, rz );
    m3 r = { rx, ry, rz };
    cam.camera_rotation = m3_transposed( r );
    return cam;
}

static v3 sample_position( ctx_t* c, const camera_t* cam, double monitor_x, double monitor_y, int linear )
{
    const acn_params* s = &c->sc->params;
    uint64_t width = s->image_width, height = s->image_height;
    double z = cam->unit_f * ( ( height >> 1 ) - monitor_y );
    double x = cam->unit_f * ( monitor_x - ( width >> 1 ) );
    v3 d = V( x, s->camera_focal_length, z );
    d = v3_of_length( d, 1.0 );

    ray_t ray;
    ray.p = v3_ld( s->camera_position );
    ray.d = m3_mlv( &cam->camera_rotation, d );
    COST( c, ACN_F_CAMERA_RAY, 0 );

    v3 out_clr = shard_skips_terms( c ) ? V( 0, 0, 0 ) : v3_ld( s->background_color );
    trans_t trans_l = { { 0, 0, 0 }, -1, -1 };
    double offs = scene_trans_hit( c, &ray, &trans_l );
    if( offs < F3_INF )
    {
        out_clr = scene_lum( c, &ray, offs, &trans_l, s->trace_depth, 1.0 );
    }
    return linear ? out_clr : cl_sat( out_clr, s->gamma );
}

/* ---- validation (what the reference would abort on) -------------------------------------------------------------- */
static int validate( const acn_flat_scene* sc )
{
    if( !sc || sc->abi_version != ACN_ABI_VERSION || !sc->nodes ) return ACN_ERR_ARG;
    if( sc->light_root < 0 || sc->matter_root < 0 || ( uint32_t )sc->light_root >= sc->n_nodes || ( uint32_t )sc->matter_root >= sc->n_nodes ) return ACN_ERR_ARG;
    if( sc->params.experimental_level != 0 ) return ACN_ERR_UNSUPPORTED;
    if( sc->params.image_height < 2 ) return ACN_ERR_ARG;
    for( uint32_t i = 0; i < sc->n_nodes; i++ )
    {
        const acn_node* n = &sc->nodes[ i ];
        if( n->texture != -1 )
        {
            if( n->texture < 0 || ( uint32_t )n->texture >= sc->n_textures || !sc->textures ) return ACN_ERR_ARG;
            const acn_texture* t = &sc->textures[ n->texture ];
            if( t->kind != ACN_TXM_PLAIN && t->kind != ACN_TXM_CHESS ) return ACN_ERR_ARG;
            if( t->kind == ACN_TXM_CHESS && n->type != ACN_PLANE && n->type != ACN_SPHERE && n->type != ACN_DISTANCE ) return ACN_ERR_UNSUPPORTED;
        }
        switch( n->type )
        {
            case ACN_PLANE: case ACN_SPHERE: case ACN_SQUAROID: case ACN_DISTANCE: break;
            case ACN_PAIR_INSIDE: case ACN_PAIR_OUTSIDE:
                if( n->child1 < 0 || ( uint32_t )n->child1 >= sc->n_nodes ) return ACN_ERR_ARG; /* fallthrough */
            case ACN_NEG: case ACN_SCALE:
                if( n->child0 < 0 || ( uint32_t )n->child0 >= sc->n_nodes ) return ACN_ERR_ARG;
                break;
            case ACN_COMPOUND:
                if( n->child1 < 0 || n->child0 < 0 || ( uint32_t )( n->child0 + n->child1 ) > sc->n_elems ) return ACN_ERR_ARG;
                for( int k = 0; k < n->child1; k++ )
                {
                    int e = sc->elems[ n->child0 + k ];
                    if( e < 0 || ( uint32_t )e >= sc->n_nodes ) return ACN_ERR_ARG;
                }
                break;
            default: return ACN_ERR_ARG;
        }
    }
    const acn_node* light = &sc->nodes[ sc->light_root ];
    if( light->type != ACN_COMPOUND || sc->nodes[ sc->matter_root ].type != ACN_COMPOUND ) return ACN_ERR_ARG;
    for( int k = 0; k < light->child1; k++ )
    {
        int t = sc->nodes[ sc->elems[ light->child0 + k ] ].type;
        if( t == ACN_COMPOUND ) return ACN_ERR_ARG; /* assert scene.c:547 */
        if( t != ACN_PLANE && t != ACN_SPHERE && t != ACN_PAIR_INSIDE && t != ACN_PAIR_OUTSIDE ) return ACN_ERR_NO_FOV;
    }
    return ACN_OK;
}

/* ---- pixel farm: src/scene.c:944-952, 1017-1028 ------------------------------------------------------------------ */
typedef struct
{
    const acn_flat_scene* sc;
    const double* pos_xy;
    double* out_rgb;
    size_t n;
    size_t* index;
    pthread_mutex_t* mutex;
    int linear;
    int count;
    uint32_t rank, world;
    uint64_t cnt[ ORC_N_COUNTERS ];
} farm_t;

static void* farm_func( void* arg )
{
    farm_t* f = arg;
    ctx_t c = { f->sc, f->count ? f->cnt : NULL, f->rank, f->world, 0 };
    camera_t cam = camera_setup( &f->sc->params );
    for( ;; )
    {
        pthread_mutex_lock( f->mutex );
        size_t first = *f->index;
        *f->index += 16;
        pthread_mutex_unlock( f->mutex );
        if( first >= f->n ) break;
        size_t last = first + 16 < f->n ? first + 16 : f->n;
        for( size_t i = first; i < last; i++ )
        {
            v3 clr = sample_position( &c, &cam, f->pos_xy[ i * 2 ], f->pos_xy[ i * 2 + 1 ], f->linear );
            f->out_rgb[ i * 3 + 0 ] = clr.x;
            f->out_rgb[ i * 3 + 1 ] = clr.y;
            f->out_rgb[ i * 3 + 2 ] = clr.z;
        }
    }
    return NULL;
}

int acn_oracle_render_positions( const acn_flat_scene* scene, const double* pos_xy, size_t n, double* out_rgb,
                                 uint32_t flags, int threads, uint64_t* counters )
{
    return acn_oracle_render_positions_shard( scene, pos_xy, n, out_rgb, flags, threads, counters, 0, 1 );
}

int acn_oracle_render_positions_shard( const acn_flat_scene* scene, const double* pos_xy, size_t n, double* out_rgb,
                                       uint32_t flags, int threads, uint64_t* counters, uint32_t rank, uint32_t world )
{
    if( world > 1 && ( rank >= world || !( flags & ACN_OPT_LINEAR_OUT ) ) ) return ACN_ERR_ARG;   /* partial sums are linear */
    int st = validate( scene );
    if( st != ACN_OK ) return st;
    if( n && ( !pos_xy || !out_rgb ) ) return ACN_ERR_ARG;
    if( threads < 1 ) threads = 1;
    if( threads > 256 ) threads = 256;
    size_t index = 0;
    pthread_mutex_t mutex = PTHREAD_MUTEX_INITIALIZER;
    farm_t* farms = calloc( threads, sizeof( farm_t ) );
    pthread_t* th = calloc( threads, sizeof( pthread_t ) );
    for( int t = 0; t < threads; t++ )
    {
        farm_t* f = &farms[ t ];
        f->sc = scene; f->pos_xy = pos_xy; f->out_rgb = out_rgb; f->n = n;
        f->index = &index; f->mutex = &mutex;
        f->linear = ( flags & ACN_OPT_LINEAR_OUT ) != 0;
        f->count = counters != NULL;
        f->rank = rank; f->world = world;
    }
    if( threads == 1 )
    {
        farm_func( &farms[ 0 ] );
    }
    else
    {
        for( int t = 0; t < threads; t++ ) pthread_create( &th[ t ], NULL, farm_func, &farms[ t ] );
        for( int t = 0; t < threads; t++ ) pthread_join( th[ t ], NULL );
    }
    if( counters )
    {
        memset( counters, 0, sizeof( uint64_t ) * ORC_N_COUNTERS );
        for( int t = 0; t < threads; t++ ) for( int k = 0; k < ORC_N_COUNTERS; k++ ) counters[ k ] += farms[ t ].cnt[ k ];
    }
    free( th );
    free( farms );
    return ACN_OK;
}

/* ---- src/objects.c:286-310 (obj_ray_exit), :312-363 (obj_estimate_envelope) -------------------------------------- */
static double obj_ray_exit( ctx_t* c, int node, const ray_t* ray, v3* p_nor )
{
    v3 nor = { 0, 0, 0 };
    double a = obj_ray_hit( c, node, ray, &nor );
    if( a >= F3_INF ) return F3_INF;
    ray_t ray_l = *ray;
    double sum = 0;
    while( a < F3_INF )
    {
        a += F3_EPS * 2;
        sum += a;
        ray_l.p = ray_pos( &ray_l, a );
        a = obj_ray_hit( c, node, &ray_l, &nor );
    }
    if( v3_mlv( nor, ray->d ) > 0 )
    {
        if( p_nor ) *p_nor = nor;
        return sum;
    }
    return F3_INF;
}

int acn_oracle_estimate_envelope( const acn_flat_scene* scene, int32_t node, uint64_t samples, uint32_t rseed,
                                  double radius_factor, double* out )
{
    if( !scene || node < 0 || ( uint32_t )node >= scene->n_nodes || scene->nodes[ node ].type == ACN_COMPOUND ) return ACN_ERR_ARG;
    ctx_t c = { scene, NULL };
    const acn_node* hdr = &scene->nodes[ node ];
    v3* pos_arr = malloc( sizeof( v3 ) * ( samples ? samples : 1 ) );
    size_t size = 0;
    v3 sum = { 0, 0, 0 };
    uint64_t rv = rseed;
    ray_t ray;
    ray.p = v3_ld( hdr->pos );
    for( uint64_t i = 0; i < samples; i++ )
    {
        ray.d = v3_random_sphere_belt( &rv, 1.0 );
        double a = obj_ray_exit( &c, node, &ray, NULL );
        if( a < F3_INF )
        {
            v3 pos = ray_pos( &ray, a );
            pos_arr[ size++ ] = pos;
            sum = v3_add( sum, ray_pos( &ray, a ) );
            ray.p = v3_mlf( sum, ( 1.0 / size ) );
            ray.p.x += F3_EPS * f3_rnd0( &rv );
            ray.p.y += F3_EPS * f3_rnd0( &rv );
            ray.p.z += F3_EPS * f3_rnd0( &rv );
        }
    }
    double radius = F3_MAG;
    if( size > 0 )
    {
        double max_r2 = 0;
        for( size_t i = 0; i < size; i++ )
        {
            double r = v3_diff_sqr( ray.p, pos_arr[ i ] );
            max_r2 = r > max_r2 ? r : max_r2;
        }
        radius = M_SQRT( max_r2 ) * radius_factor;
    }
    free( pos_arr );
    out[ 0 ] = ray.p.x; out[ 1 ] = ray.p.y; out[ 2 ] = ray.p.z; out[ 3 ] = radius;
    return ACN_OK;
}

/* ---- leaf exports for known-answer tests ------------------------------------------------------------------------- */
static void st3( double* d, v3 v ) { d[ 0 ] = v.x; d[ 1 ] = v.y; d[ 2 ] = v.z; }

double acn_oracle_sphere_ray_hit( const double* pos3, double r, const double* ray_p3, const double* ray_d3, double* nor3 )
{
    ray_t ray = { v3_ld( ray_p3 ), v3_ld( ray_d3 ) };
    v3 nor = { 0, 0, 0 };
    double a = sphere_ray_hit( v3_ld( pos3 ), r, &ray, &nor );
    if( nor3 ) st3( nor3, nor );
    return a;
}

double acn_oracle_plane_ray_hit( const double* pos3, const double* nor3, const double* ray_p3, const double* ray_d3 )
{
    ray_t ray = { v3_ld( ray_p3 ), v3_ld( ray_d3 ) };
    return plane_ray_hit( v3_ld( pos3 ), v3_ld( nor3 ), &ray, NULL );
}

double acn_oracle_fresnel_reflection( const double* dir3, const double* exit_nor3, double trix, double* out_dir3 )
{
    v3 d = { 0, 0, 0 };
    double r = fresnel_reflection( v3_ld( dir3 ), v3_ld( exit_nor3 ), trix, &d );
    if( out_dir3 ) st3( out_dir3, d );
    return r;
}

void acn_oracle_fresnel_refraction( const double* dir3, const double* exit_nor3, double trix, double* out_dir3 )
{
    v3 d = { 0, 0, 0 };
    fresnel_refraction( v3_ld( dir3 ), v3_ld( exit_nor3 ), trix, &d );
    st3( out_dir3, d );
}

double acn_oracle_obj_ray_hit( const acn_flat_scene* scene, int32_t node, const double* ray_p3, const double* ray_d3, double* nor3 )
{
    ctx_t c = { scene, NULL };
    ray_t ray = { v3_ld( ray_p3 ), v3_ld( ray_d3 ) };
    v3 nor = { 0, 0, 0 };
    double a = obj_ray_hit( &c, node, &ray, nor3 ? &nor : NULL );
    if( nor3 ) st3( nor3, nor );
    return a;
}

int acn_oracle_obj_side( const acn_flat_scene* scene, int32_t node, const double* pos3 )
{
    ctx_t c = { scene, NULL };
    return obj_side( &c, node, v3_ld( pos3 ) );
}

double acn_oracle_trans_hit( const acn_flat_scene* scene, const double* ray_p3, const double* ray_d3,
                             double* exit_nor3, int32_t* exit_obj, int32_t* enter_obj )
{
    ctx_t c = { scene, NULL };
    ray_t ray = { v3_ld( ray_p3 ), v3_ld( ray_d3 ) };
    trans_t t = { { 0, 0, 0 }, -1, -1 };
    double a = scene_trans_hit( &c, &ray, &t );
    if( exit_nor3 ) st3( exit_nor3, t.exit_nor );
    if( exit_obj ) *exit_obj = t.exit_obj;
    if( enter_obj ) *enter_obj = t.enter_obj;
    return a;
}

uint64_t acn_oracle_random_seed( const double* v, uint64_t rv ) { return v3_random_seed( v3_ld( v ), rv ); }

void acn_oracle_sphere_cap( uint64_t* rv, double h, double* out3 ) { st3( out3, v3_random_sphere_cap( rv, h ) ); }
