/* acn_scene.c -- host-side object / compound / scene model and flattening (see include/acn_scene.h).
 *
 * This is scene ASSEMBLY: it runs once per scene, stays plain C on the host, and its
 * only job is to reproduce the reference's construction rules so that the flattened scene handed to the GPU
 * path has the contents the reference's object graph would have.  No ray is traced here.
 * Reference lines are cited per function (paths relative to /root/reference).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acn_scene.h"

struct acn_obj
{
    int type;
    /* properties_s (objects.h:51-78) */
    acn_v3 pos;
    acn_m3 rax;
    acn_v3 color;
    double radiance, refractive_index;
    double fresnel_reflectivity, chromatic_reflectivity, diffuse_reflectivity, sigma, surface_roughness;
    acn_v3 transparency;
    int    has_env;
    acn_v3 env_pos;
    double env_radius;
    int    has_tex;          /* prp.texture_field (deep-copied with the properties) */
    acn_texture tex;
    /* type parameters */
    double prm[ 4 ];
    int sdf_kind;
    int cycles;
    acn_obj* o1;
    acn_obj* o2;
    /* compound_s (compound.c:36-50) */
    acn_obj** data;
    size_t size, space;
};

/* ---- small vector helpers (vectors.h) ---- */
static acn_v3 V( double x, double y, double z ) { acn_v3 v = { x, y, z }; return v; }
static acn_v3 v_add( acn_v3 a, acn_v3 b ) { return V( a.x + b.x, a.y + b.y, a.z + b.z ); }
static acn_v3 v_sub( acn_v3 a, acn_v3 b ) { return V( a.x - b.x, a.y - b.y, a.z - b.z ); }
static acn_v3 v_mlf( acn_v3 a, double f ) { return V( a.x * f, a.y * f, a.z * f ); }
static double v_sqr( acn_v3 a ) { return ( a.x * a.x ) + ( a.y * a.y ) + ( a.z * a.z ); }
static acn_v3 v_of_length( acn_v3 o, double a )   /* vectors.h:148-154 */
{
    double r_sqr = v_sqr( o );
    if( fabs( r_sqr - 1.0 ) < 1E-8 ) return o;
    double f = r_sqr > 0 ? ( a / sqrt( r_sqr ) ) : 0;
    return V( o.x * f, o.y * f, o.z * f );
}
static acn_v3 m_mlv( const acn_m3* o, acn_v3 v )   /* vectors.h:256-265 */
{
    return V( o->x.x * v.x + o->x.y * v.y + o->x.z * v.z,
              o->y.x * v.x + o->y.y * v.y + o->y.z * v.z,
              o->z.x * v.x + o->z.y * v.y + o->z.z * v.z );
}
static acn_m3 m_mlm( const acn_m3* o, const acn_m3* a )   /* vectors.h:279-282 */
{
    acn_m3 r = { m_mlv( o, a->x ), m_mlv( o, a->y ), m_mlv( o, a->z ) };
    return r;
}

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* vectors.h:290-309 with the degree conversion of closures.c:104-127 */
acn_m3 acn_rotx( double deg )
{
    double a = ( M_PI / 180.0 ) * deg, sa = sin( a ), ca = cos( a );
    acn_m3 m = { { 1, 0, 0 }, { 0, ca, -sa }, { 0, sa, ca } };
    return m;
}
acn_m3 acn_roty( double deg )
{
    double a = ( M_PI / 180.0 ) * deg, sa = sin( a ), ca = cos( a );
    acn_m3 m = { { ca, 0, sa }, { 0, 1, 0 }, { -sa, 0, ca } };
    return m;
}
acn_m3 acn_rotz( double deg )
{
    double a = ( M_PI / 180.0 ) * deg, sa = sin( a ), ca = cos( a );
    acn_m3 m = { { ca, -sa, 0 }, { sa, ca, 0 }, { 0, 0, 1 } };
    return m;
}

/* ---- creation ---- */

/* properties_s_init_a objects.c:167-177: everything zero except pos, rax, color */
static acn_obj* obj_create( int type )
{
    acn_obj* o = calloc( 1, sizeof( acn_obj ) );
    o->type = type;
    o->rax.x = V( 1, 0, 0 );
    o->rax.y = V( 0, 1, 0 );
    o->rax.z = V( 0, 0, 1 );
    o->color = V( 0.7, 0.7, 0.7 );
    return o;
}

int acn_obj_type( const acn_obj* o ) { return o->type; }

acn_obj* acn_obj_plane_s_create( void ) { return obj_create( ACN_PLANE ); }

acn_obj* acn_obj_sphere_s_create( double radius )
{
    acn_obj* o = obj_create( ACN_SPHERE );
    o->prm[ 0 ] = radius;
    return o;
}

acn_obj* acn_obj_squaroid_s_create_squaroid( double a, double b, double c, double r )
{
    acn_obj* o = obj_create( ACN_SQUAROID );
    o->prm[ 0 ] = a; o->prm[ 1 ] = b; o->prm[ 2 ] = c; o->prm[ 3 ] = r;
    return o;
}

static double inv_sqr_or_1( double r ) { return ( r != 0 ) ? 1.0 / ( r * r ) : 1.0; }

/* objects.c:723-776 */
acn_obj* acn_obj_squaroid_s_create_ellipsoid( double rx, double ry, double rz )
{
    return acn_obj_squaroid_s_create_squaroid( inv_sqr_or_1( rx ), inv_sqr_or_1( ry ), inv_sqr_or_1( rz ), -1 );
}
acn_obj* acn_obj_squaroid_s_create_hyperboloid1( double rx, double ry, double rz )
{
    return acn_obj_squaroid_s_create_squaroid( inv_sqr_or_1( rx ), inv_sqr_or_1( ry ), -inv_sqr_or_1( rz ), -1 );
}
acn_obj* acn_obj_squaroid_s_create_hyperboloid2( double rx, double ry, double rz )
{
    return acn_obj_squaroid_s_create_squaroid( inv_sqr_or_1( rx ), inv_sqr_or_1( ry ), -inv_sqr_or_1( rz ), 1 );
}
acn_obj* acn_obj_squaroid_s_create_cone( double rx, double ry, double rz )
{
    return acn_obj_squaroid_s_create_squaroid( inv_sqr_or_1( rx ), inv_sqr_or_1( ry ), -inv_sqr_or_1( rz ), 0 );
}
acn_obj* acn_obj_squaroid_s_create_cylinder( double rx, double ry )
{
    return acn_obj_squaroid_s_create_squaroid( inv_sqr_or_1( rx ), inv_sqr_or_1( ry ), 0, -1 );
}

/* closures.c:568-591 */
acn_obj* acn_obj_torus_create( double radius1, double radius2 )
{
    acn_obj* o = obj_create( ACN_DISTANCE );
    o->prm[ 0 ] = 1.0;                 /* inv_scale */
    o->cycles   = 200;
    o->sdf_kind = ACN_SDF_TORUS;
    o->prm[ 1 ] = radius2 / radius1;   /* ex_radius */
    acn_obj_scale( o, radius1 );
    acn_obj_set_envelope( o, V( 0, 0, 0 ), ( radius1 + radius2 ) * 1.01 );
    return o;
}

/* objects.c:853-861 (defaults of the def string), 1691-1710 */
acn_obj* acn_obj_distance_s_create( void )
{
    acn_obj* o = obj_create( ACN_DISTANCE );
    o->prm[ 0 ] = 1.0;   /* inv_scale */
    o->cycles   = 200;
    o->sdf_kind = ACN_SDF_SPHERE;
    return o;
}
int acn_obj_set_distance_function( acn_obj* o, int sdf_kind, double ex_radius )
{
    if( !o || o->type != ACN_DISTANCE ) return ACN_ERR_ARG;
    if( sdf_kind != ACN_SDF_SPHERE && sdf_kind != ACN_SDF_TORUS ) return ACN_ERR_ARG;
    o->sdf_kind = sdf_kind;
    o->prm[ 1 ] = sdf_kind == ACN_SDF_TORUS ? ex_radius : 0.0;
    return ACN_OK;
}

acn_obj* acn_obj_clone( const acn_obj* o )
{
    if( !o ) return NULL;
    acn_obj* c = malloc( sizeof( acn_obj ) );
    *c = *o;
    c->o1 = acn_obj_clone( o->o1 );
    c->o2 = acn_obj_clone( o->o2 );
    c->data = NULL;
    c->space = 0;
    if( o->size )
    {
        c->data = malloc( sizeof( acn_obj* ) * o->size );
        c->space = o->size;
        for( size_t i = 0; i < o->size; i++ ) c->data[ i ] = acn_obj_clone( o->data[ i ] );
    }
    return c;
}

void acn_obj_discard( acn_obj* o )
{
    if( !o ) return;
    acn_obj_discard( o->o1 );
    acn_obj_discard( o->o2 );
    for( size_t i = 0; i < o->size; i++ ) acn_obj_discard( o->data[ i ] );
    free( o->data );
    free( o );
}

/* properties_s_copy (deep, includes the envelope) */
static void prp_copy( acn_obj* dst, const acn_obj* src )
{
    dst->pos = src->pos; dst->rax = src->rax; dst->color = src->color;
    dst->radiance = src->radiance; dst->refractive_index = src->refractive_index;
    dst->fresnel_reflectivity = src->fresnel_reflectivity; dst->chromatic_reflectivity = src->chromatic_reflectivity;
    dst->diffuse_reflectivity = src->diffuse_reflectivity; dst->sigma = src->sigma;
    dst->surface_roughness = src->surface_roughness; dst->transparency = src->transparency;
    dst->has_env = src->has_env; dst->env_pos = src->env_pos; dst->env_radius = src->env_radius;
    dst->has_tex = src->has_tex; dst->tex = src->tex;
}

/* objects.c:1011-1018 */
acn_obj* acn_obj_pair_inside_s_create_pair( const acn_obj* o1, const acn_obj* o2 )
{
    acn_obj* o = obj_create( ACN_PAIR_INSIDE );
    prp_copy( o, o1 );
    o->o1 = acn_obj_clone( o1 );
    o->o2 = acn_obj_clone( o2 );
    return o;
}

/* objects.c:1161-1176 */
acn_obj* acn_obj_pair_outside_s_create_pair( const acn_obj* o1, const acn_obj* o2 )
{
    acn_obj* o = obj_create( ACN_PAIR_OUTSIDE );
    prp_copy( o, o1 );
    o->o1 = acn_obj_clone( o1 );
    o->o2 = acn_obj_clone( o2 );
    o->has_env = 0;
    return o;
}

/* objects.c:1315-1321 */
acn_obj* acn_obj_neg_s_create_neg( const acn_obj* o1 )
{
    acn_obj* o = obj_create( ACN_NEG );
    prp_copy( o, o1 );
    o->o1 = acn_obj_clone( o1 );
    return o;
}

/* objects.c:1388-1407 */
acn_obj* acn_obj_scale_s_create_scale( const acn_obj* o1, acn_v3 scale )
{
    acn_obj* o = obj_create( ACN_SCALE );
    prp_copy( o, o1 );
    o->pos = V( 0, 0, 0 );
    o->rax.x = V( 1, 0, 0 ); o->rax.y = V( 0, 1, 0 ); o->rax.z = V( 0, 0, 1 );
    if( o->has_env )
    {
        o->env_pos = V( o->env_pos.x * scale.x, o->env_pos.y * scale.y, o->env_pos.z * scale.z );
        double v = ( scale.x > scale.y ) ? scale.x : scale.y;
        v = ( v > scale.z ) ? v : scale.z;
        o->env_radius *= v;
    }
    o->o1 = acn_obj_clone( o1 );
    o->prm[ 0 ] = ( scale.x != 0 ) ? ( 1.0 / scale.x ) : 1.0;
    o->prm[ 1 ] = ( scale.y != 0 ) ? ( 1.0 / scale.y ) : 1.0;
    o->prm[ 2 ] = ( scale.z != 0 ) ? ( 1.0 / scale.z ) : 1.0;
    return o;
}

/* container.c:376-410 */
static acn_obj* composite( acn_obj* const* list, size_t start, size_t size, int inside )
{
    if( size == 1 ) return acn_obj_clone( list[ start ] );
    acn_obj* a = composite( list, start, size >> 1, inside );
    acn_obj* b = composite( list, start + ( size >> 1 ), size - ( size >> 1 ), inside );
    acn_obj* r = inside ? acn_obj_pair_inside_s_create_pair( a, b ) : acn_obj_pair_outside_s_create_pair( a, b );
    acn_obj_discard( a );
    acn_obj_discard( b );
    return r;
}

acn_obj* acn_create_inside_composite( acn_obj* const* list, size_t size ) { return size ? composite( list, 0, size, 1 ) : NULL; }
acn_obj* acn_create_outside_composite( acn_obj* const* list, size_t size ) { return size ? composite( list, 0, size, 0 ) : NULL; }

/* ---- transforms ---- */

/* properties_s_move/rotate/scale objects.c:179-196; envelope_s_* :44-59 */
static void prp_move( acn_obj* o, acn_v3 vec )
{
    o->pos = v_add( o->pos, vec );
    if( o->has_env ) o->env_pos = v_add( o->env_pos, vec );
}
static void prp_rotate( acn_obj* o, const acn_m3* mat )
{
    o->rax = m_mlm( mat, &o->rax );
    o->pos = m_mlv( mat, o->pos );
    if( o->has_env ) o->env_pos = m_mlv( mat, o->env_pos );
}
static void prp_scale( acn_obj* o, double fac )
{
    o->pos = v_mlf( o->pos, fac );
    if( o->has_env ) { o->env_pos = v_mlf( o->env_pos, fac ); o->env_radius *= fac; }
}

void acn_obj_move( acn_obj* o, acn_v3 vec )
{
    switch( o->type )
    {
        case ACN_COMPOUND: /* compound.c:301-320 */
            if( o->has_env ) o->env_pos = v_add( o->env_pos, vec );
            for( size_t i = 0; i < o->size; i++ ) acn_obj_move( o->data[ i ], vec );
            break;
        case ACN_PAIR_INSIDE: case ACN_PAIR_OUTSIDE: /* objects.c:1101-1106, 1258-1263 */
            prp_move( o, vec ); acn_obj_move( o->o1, vec ); acn_obj_move( o->o2, vec ); break;
        case ACN_NEG: /* objects.c:1346 */
            prp_move( o, vec ); acn_obj_move( o->o1, vec ); break;
        default: /* incl. ACN_SCALE objects.c:1445-1448: only itself */
            prp_move( o, vec ); break;
    }
}

void acn_obj_rotate( acn_obj* o, const acn_m3* mat )
{
    switch( o->type )
    {
        case ACN_COMPOUND: /* compound.c:322-341 */
            if( o->has_env ) o->env_pos = m_mlv( mat, o->env_pos );
            for( size_t i = 0; i < o->size; i++ ) acn_obj_rotate( o->data[ i ], mat );
            break;
        case ACN_PAIR_INSIDE: case ACN_PAIR_OUTSIDE:
            prp_rotate( o, mat ); acn_obj_rotate( o->o1, mat ); acn_obj_rotate( o->o2, mat ); break;
        case ACN_NEG:
            prp_rotate( o, mat ); acn_obj_rotate( o->o1, mat ); break;
        default:
            prp_rotate( o, mat ); break;
    }
}

void acn_obj_scale( acn_obj* o, double fac )
{
    switch( o->type )
    {
        case ACN_COMPOUND: /* compound.c:343-362 */
            if( o->has_env ) { o->env_pos = v_mlf( o->env_pos, fac ); o->env_radius *= fac; }
            for( size_t i = 0; i < o->size; i++ ) acn_obj_scale( o->data[ i ], fac );
            break;
        case ACN_SPHERE:   prp_scale( o, fac ); o->prm[ 0 ] *= fac; break;                 /* objects.c:661 */
        case ACN_SQUAROID: prp_scale( o, fac ); o->prm[ 3 ] *= ( fac * fac ); break;       /* objects.c:831 */
        case ACN_DISTANCE: prp_scale( o, fac ); o->prm[ 0 ] *= 1.0 / fac; break;           /* objects.c:970 */
        case ACN_PAIR_INSIDE: case ACN_PAIR_OUTSIDE:
            prp_scale( o, fac ); acn_obj_scale( o->o1, fac ); acn_obj_scale( o->o2, fac ); break;
        case ACN_NEG:
            prp_scale( o, fac ); acn_obj_scale( o->o1, fac ); break;
        case ACN_SCALE: /* objects.c:1455-1459 */
        {
            prp_scale( o, fac );
            double f = ( fac != 0 ) ? 1.0 / fac : 1.0;
            o->prm[ 0 ] *= f; o->prm[ 1 ] *= f; o->prm[ 2 ] *= f;
            break;
        }
        default: prp_scale( o, fac ); break;
    }
}

/* ---- properties ---- */
void acn_obj_set_color( acn_obj* o, acn_v3 c ) { o->color = c; }
void acn_obj_set_transparency( acn_obj* o, acn_v3 c ) { o->transparency = c; }
void acn_obj_set_refractive_index( acn_obj* o, double v )   /* objects.c:436-448 */
{
    o->refractive_index = v;
    o->fresnel_reflectivity = ( v == 1.0 ) ? 0.0 : 1.0;
}
void acn_obj_set_radiance( acn_obj* o, double v ) { o->radiance = v; }
void acn_obj_set_fresnel_reflectivity( acn_obj* o, double v ) { o->fresnel_reflectivity = v; }
void acn_obj_set_chromatic_reflectivity( acn_obj* o, double v ) { o->chromatic_reflectivity = v; }
void acn_obj_set_diffuse_reflectivity( acn_obj* o, double v ) { o->diffuse_reflectivity = v; }
void acn_obj_set_sigma( acn_obj* o, double v ) { o->sigma = v; }
void acn_obj_set_surface_roughness( acn_obj* o, double v ) { o->surface_roughness = v; }
double acn_obj_radiance( const acn_obj* o ) { return o->radiance; }
void   acn_obj_get_pos( const acn_obj* o, double* pos3 ) { pos3[ 0 ] = o->pos.x; pos3[ 1 ] = o->pos.y; pos3[ 2 ] = o->pos.z; }
double acn_obj_sphere_s_get_radius( const acn_obj* o ) { return o->type == ACN_SPHERE ? o->prm[ 0 ] : 0; }   /* objects.c:601-604 */

/* The reflected leaf members a script can reach as `obj.name` (objects.c:571-577 sphere, 685-694 squaroid) */
static double* obj_field( const acn_obj* o, const char* name )
{
    acn_obj* m = ( acn_obj* )o;
    if( o->type == ACN_SPHERE && !strcmp( name, "radius" ) ) return &m->prm[ 0 ];
    if( o->type == ACN_SQUAROID && name[ 0 ] && !name[ 1 ] )
    {
        if( name[ 0 ] == 'a' ) return &m->prm[ 0 ];
        if( name[ 0 ] == 'b' ) return &m->prm[ 1 ];
        if( name[ 0 ] == 'c' ) return &m->prm[ 2 ];
        if( name[ 0 ] == 'r' ) return &m->prm[ 3 ];
    }
    if( o->type == ACN_DISTANCE && !strcmp( name, "inv_scale" ) ) return &m->prm[ 0 ];
    return NULL;
}
int acn_obj_get_field( const acn_obj* o, const char* name, double* value )
{
    if( o->type == ACN_DISTANCE && !strcmp( name, "cycles" ) ) { *value = ( double )o->cycles; return 1; }
    double* p = obj_field( o, name );
    if( !p ) return 0;
    *value = *p;
    return 1;
}
int acn_obj_set_field( acn_obj* o, const char* name, double value )
{
    if( o->type == ACN_DISTANCE && !strcmp( name, "cycles" ) ) { o->cycles = value > 0 ? ( int )value : 0; return 1; }
    double* p = obj_field( o, name );
    if( !p ) return 0;
    *p = value;
    return 1;
}

void acn_obj_set_texture_field_plain( acn_obj* o, acn_v3 color )
{
    memset( &o->tex, 0, sizeof( o->tex ) );
    o->has_tex = 1; o->tex.kind = ACN_TXM_PLAIN;
    o->tex.color1[ 0 ] = color.x; o->tex.color1[ 1 ] = color.y; o->tex.color1[ 2 ] = color.z;
    o->tex.scale = 1.0;
}

void acn_obj_set_texture_field_chess( acn_obj* o, acn_v3 c1, acn_v3 c2, double scale )
{
    memset( &o->tex, 0, sizeof( o->tex ) );
    o->has_tex = 1; o->tex.kind = ACN_TXM_CHESS;
    o->tex.color1[ 0 ] = c1.x; o->tex.color1[ 1 ] = c1.y; o->tex.color1[ 2 ] = c1.z;
    o->tex.color2[ 0 ] = c2.x; o->tex.color2[ 1 ] = c2.y; o->tex.color2[ 2 ] = c2.z;
    o->tex.scale = scale;
}

void acn_obj_clear_texture_field( acn_obj* o ) { o->has_tex = 0; }

void acn_obj_set_envelope( acn_obj* o, acn_v3 pos, double radius )
{
    o->has_env = 1; o->env_pos = pos; o->env_radius = radius;
}

int acn_obj_get_envelope( const acn_obj* o, double* out )
{
    if( !o->has_env ) return 0;
    out[ 0 ] = o->env_pos.x; out[ 1 ] = o->env_pos.y; out[ 2 ] = o->env_pos.z; out[ 3 ] = o->env_radius;
    return 1;
}

/* objects.c:1582-1690 */
int acn_obj_set_material( acn_obj* o, const char* name )
{
    static const struct { const char* name; double n; double t[ 3 ]; double fresnel, chromatic, diffuse; int set_sigma; double sigma; int set_color; double c[ 3 ]; } tab[] =
    {
        { "transparent",      1.0,  { 1, 1, 1 },        1, 0, 0, 0, 0,    0, { 0, 0, 0 } },
        { "glass",            1.46, { 0.8, 0.9, 0.9 },  1, 0, 0, 0, 0,    0, { 0, 0, 0 } },
        { "water",            1.32, { 0.5, 0.9, 0.99 }, 1, 0, 0, 0, 0,    0, { 0, 0, 0 } },
        { "sapphire",         1.76, { 0.7, 0.7, 0.7 },  1, 0, 0, 0, 0,    0, { 0, 0, 0 } },
        { "diamond",          2.42, { 0.8, 0.8, 0.8 },  1, 0, 0, 0, 0,    0, { 0, 0, 0 } },
        { "diffuse",          1.0,  { 0, 0, 0 },        0, 0, 1, 1, 0.29, 0, { 0, 0, 0 } },
        { "diffuse_polished", 1.5,  { 0, 0, 0 },        1, 0, 1, 1, 0.29, 0, { 0, 0, 0 } },
        { "perfect_mirror",   1.0,  { 0, 0, 0 },        0, 1, 0, 0, 0,    1, { 1, 1, 1 } },
        { "mirror",           1.0,  { 0, 0, 0 },        0, 1, 0, 0, 0,    1, { 0.92, 0.94, 0.87 } },
        { "gold",             1.0,  { 0, 0, 0 },        0, 1, 0, 0, 0,    1, { 0.83, 0.69, 0.22 } },
        { "silver",           1.0,  { 0, 0, 0 },        0, 1, 0, 0, 0,    1, { 0.8, 0.8, 0.8 } },
    };
    for( size_t i = 0; i < sizeof( tab ) / sizeof( tab[ 0 ] ); i++ )
    {
        if( strcmp( name, tab[ i ].name ) == 0 )
        {
            o->refractive_index = tab[ i ].n;
            o->transparency = V( tab[ i ].t[ 0 ], tab[ i ].t[ 1 ], tab[ i ].t[ 2 ] );
            o->fresnel_reflectivity = tab[ i ].fresnel;
            o->chromatic_reflectivity = tab[ i ].chromatic;
            o->diffuse_reflectivity = tab[ i ].diffuse;
            if( tab[ i ].set_sigma ) o->sigma = tab[ i ].sigma;
            if( tab[ i ].set_color ) o->color = V( tab[ i ].c[ 0 ], tab[ i ].c[ 1 ], tab[ i ].c[ 2 ] );
            return ACN_OK;
        }
    }
    return ACN_ERR_ARG;
}

/* ---- compound_s ---- */
acn_obj* acn_compound_s_create( void ) { return obj_create( ACN_COMPOUND ); }
size_t acn_compound_s_get_size( const acn_obj* c ) { return c ? c->size : 0; }

void acn_compound_s_clear( acn_obj* c )
{
    for( size_t i = 0; i < c->size; i++ ) acn_obj_discard( c->data[ i ] );
    c->size = 0;
}

static void compound_append( acn_obj* c, acn_obj* owned )
{
    if( c->size == c->space )
    {
        c->space = c->space ? c->space * 2 : 8;
        c->data = realloc( c->data, sizeof( acn_obj* ) * c->space );
    }
    c->data[ c->size++ ] = owned;
}

/* objects.c:105-136 */
static void envelope_of_pair( acn_v3 p1, double r1, acn_v3 p2, double r2, acn_v3* pos, double* radius )
{
    acn_v3 diff = v_sub( p1, p2 );
    double d = sqrt( v_sqr( diff ) );
    double rmax = r1 > r2 ? r1 : r2;
    double rmin = r1 < r2 ? r1 : r2;
    if( rmin + d <= rmax )
    {
        if( r1 > r2 ) { *pos = p1; *radius = r1; } else { *pos = p2; *radius = r2; }
    }
    else
    {
        acn_v3 q1 = v_add( p1, v_of_length( diff, r1 ) );
        acn_v3 q2 = v_sub( p2, v_of_length( diff, r2 ) );
        *pos = v_mlf( v_add( q1, q2 ), 0.5 );
        *radius = ( r1 + r2 + d ) * 0.5;
    }
}

/* compound.c:140-207 */
void acn_compound_s_push( acn_obj* o, const acn_obj* object )
{
    if( !object ) return;
    if( object->type != ACN_COMPOUND )
    {
        acn_obj* dst = acn_obj_clone( object );
        compound_append( o, dst );
        if( o->has_env )
        {
            if( dst->has_env )
            {
                envelope_of_pair( o->env_pos, o->env_radius, dst->env_pos, dst->env_radius, &o->env_pos, &o->env_radius );
            }
            else
            {
                o->has_env = 0;
            }
        }
        else if( o->size == 1 )
        {
            o->has_env = dst->has_env; o->env_pos = dst->env_pos; o->env_radius = dst->env_radius;
        }
    }
    else
    {
        if( object->has_env )
        {
            compound_append( o, acn_obj_clone( object ) );
        }
        else
        {
            for( size_t i = 0; i < object->size; i++ ) acn_compound_s_push( o, object->data[ i ] );
        }
    }
}

void acn_compound_s_set_sphere_envelopes( acn_obj* c, double factor )
{
    for( size_t i = 0; i < c->size; i++ )
    {
        acn_obj* e = c->data[ i ];
        if( e->type == ACN_COMPOUND ) acn_compound_s_set_sphere_envelopes( e, factor );
        else if( e->type == ACN_SPHERE && !e->has_env ) acn_obj_set_envelope( e, e->pos, e->prm[ 0 ] * factor );
    }
}

/* ---- auto envelope: objects.c:470-476 (1000 samples, seed 123, factor 1.1), compound.c:73-107 ---- */
static acn_envelope_estimator_fn envelope_estimator_g = NULL;
void acn_set_envelope_estimator( acn_envelope_estimator_fn fn ) { envelope_estimator_g = fn; }

int acn_obj_set_auto_envelope( acn_obj* o )
{
    if( o->type != ACN_COMPOUND )
    {
        acn_flat_scene f;
        int32_t node = -1;
        int st = acn_obj_flatten( o, &f, &node );
        if( st != ACN_OK ) return st;
        double env[ 4 ] = { 0, 0, 0, 0 };
        if( envelope_estimator_g ) st = envelope_estimator_g( &f, node, 1000, 123, 1.1, env );
        else
        {
            acn_scene_handle* h = NULL;
            st = acn_scene_upload( &f, 0, &h );
            if( st == ACN_OK ) st = acn_estimate_envelope( h, node, 1000, 123, 1.1, env );
            if( h ) acn_scene_free( h );
        }
        acn_flat_scene_free( &f );
        if( st != ACN_OK ) return st;
        acn_obj_set_envelope( o, V( env[ 0 ], env[ 1 ], env[ 2 ] ), env[ 3 ] );
        return ACN_OK;
    }
    o->has_env = 0;
    for( size_t i = 0; i < o->size; i++ )
    {
        acn_obj* e = o->data[ i ];
        if( !e->has_env )
        {
            int st = acn_obj_set_auto_envelope( e );
            if( st != ACN_OK ) return st;
        }
        if( o->has_env )
        {
            envelope_of_pair( o->env_pos, o->env_radius, e->env_pos, e->env_radius, &o->env_pos, &o->env_radius );
        }
        else
        {
            o->has_env = 1; o->env_pos = e->env_pos; o->env_radius = e->env_radius;
        }
    }
    return ACN_OK;
}

/* ---- scene_s ---- */
acn_scene* acn_scene_s_create( void )   /* scene.c:185-213, 217-223 */
{
    acn_scene* s = calloc( 1, sizeof( acn_scene ) );
    s->threads = 10;
    s->prm.image_width = 800;
    s->prm.image_height = 600;
    s->prm.gamma = 1.0;
    s->gradient_threshold = 0.1;
    s->gradient_samples = 10;
    s->gradient_cycles = 1;
    s->prm.camera_focal_length = 1.0;
    s->prm.trace_depth = 11;
    s->prm.trace_min_intensity = 0;
    s->prm.direct_samples = 100;
    s->prm.path_samples = 0;
    s->prm.max_path_length = 1E+30;
    s->prm.experimental_level = 0;
    s->light = acn_compound_s_create();
    s->matter = acn_compound_s_create();
    return s;
}

void acn_scene_s_discard( acn_scene* o )
{
    if( !o ) return;
    acn_obj_discard( o->light );
    acn_obj_discard( o->matter );
    free( o );
}

void acn_scene_s_clear( acn_scene* o )   /* scene.c:671-675 */
{
    acn_compound_s_clear( o->light );  o->light->has_env = 0;
    acn_compound_s_clear( o->matter ); o->matter->has_env = 0;
}

size_t acn_scene_s_push( acn_scene* o, const acn_obj* object )   /* scene.c:238-279 */
{
    if( object->type != ACN_COMPOUND )
    {
        if( acn_obj_radiance( object ) > 0 ) acn_compound_s_push( o->light, object );
        else                                 acn_compound_s_push( o->matter, object );
        return 1;
    }
    acn_compound_s_push( o->matter, object );
    return 0;
}

size_t acn_scene_s_objects( const acn_scene* o )   /* scene.c:283-289 */
{
    return acn_compound_s_get_size( o->light ) + acn_compound_s_get_size( o->matter );
}

/* ---- flattening ---- */
typedef struct
{
    acn_node* nodes; uint32_t n_nodes, node_space;
    int32_t*  elems; uint32_t n_elems, elem_space;
    acn_texture* tex; uint32_t n_tex, tex_space;
} flat_builder;

static int32_t fb_new_node( flat_builder* b )
{
    if( b->n_nodes == b->node_space )
    {
        b->node_space = b->node_space ? b->node_space * 2 : 64;
        b->nodes = realloc( b->nodes, sizeof( acn_node ) * b->node_space );
    }
    memset( &b->nodes[ b->n_nodes ], 0, sizeof( acn_node ) );
    return ( int32_t )b->n_nodes++;
}

static int32_t fb_new_elems( flat_builder* b, uint32_t n )
{
    while( b->n_elems + n > b->elem_space )
    {
        b->elem_space = b->elem_space ? b->elem_space * 2 : 64;
        b->elems = realloc( b->elems, sizeof( int32_t ) * b->elem_space );
    }
    int32_t first = ( int32_t )b->n_elems;
    b->n_elems += n;
    return first;
}

static int32_t fb_add( flat_builder* b, const acn_obj* o )
{
    int32_t idx = fb_new_node( b );
    acn_node n;
    memset( &n, 0, sizeof( n ) );
    n.type = o->type;
    n.flags = o->has_env ? ACN_NODE_HAS_ENVELOPE : 0;
    n.child0 = n.child1 = -1;
    n.sdf_kind = o->sdf_kind;
    n.cycles = o->cycles;
    n.texture = -1;
    if( o->has_tex )
    {
        if( b->n_tex == b->tex_space )
        {
            b->tex_space = b->tex_space ? b->tex_space * 2 : 8;
            b->tex = realloc( b->tex, sizeof( acn_texture ) * b->tex_space );
        }
        n.texture = ( int32_t )b->n_tex;
        b->tex[ b->n_tex++ ] = o->tex;
    }
    n.pos[ 0 ] = o->pos.x; n.pos[ 1 ] = o->pos.y; n.pos[ 2 ] = o->pos.z;
    n.rax[ 0 ] = o->rax.x.x; n.rax[ 1 ] = o->rax.x.y; n.rax[ 2 ] = o->rax.x.z;
    n.rax[ 3 ] = o->rax.y.x; n.rax[ 4 ] = o->rax.y.y; n.rax[ 5 ] = o->rax.y.z;
    n.rax[ 6 ] = o->rax.z.x; n.rax[ 7 ] = o->rax.z.y; n.rax[ 8 ] = o->rax.z.z;
    n.env_pos[ 0 ] = o->env_pos.x; n.env_pos[ 1 ] = o->env_pos.y; n.env_pos[ 2 ] = o->env_pos.z;
    n.env_radius = o->env_radius;
    memcpy( n.prm, o->prm, sizeof( n.prm ) );
    n.color[ 0 ] = o->color.x; n.color[ 1 ] = o->color.y; n.color[ 2 ] = o->color.z;
    n.radiance = o->radiance;
    n.refractive_index = o->refractive_index;
    n.fresnel_reflectivity = o->fresnel_reflectivity;
    n.chromatic_reflectivity = o->chromatic_reflectivity;
    n.diffuse_reflectivity = o->diffuse_reflectivity;
    n.sigma = o->sigma;
    n.surface_roughness = o->surface_roughness;
    n.transparency[ 0 ] = o->transparency.x; n.transparency[ 1 ] = o->transparency.y; n.transparency[ 2 ] = o->transparency.z;

    if( o->type == ACN_COMPOUND )
    {
        int32_t first = fb_new_elems( b, ( uint32_t )o->size );
        n.child0 = first;
        n.child1 = ( int32_t )o->size;
        for( size_t i = 0; i < o->size; i++ )
        {
            int32_t e = fb_add( b, o->data[ i ] );
            b->elems[ first + i ] = e;
        }
    }
    else
    {
        if( o->o1 ) n.child0 = fb_add( b, o->o1 );
        if( o->o2 ) n.child1 = fb_add( b, o->o2 );
    }
    b->nodes[ idx ] = n;
    return idx;
}

int acn_scene_s_flatten( const acn_scene* o, acn_flat_scene* out )
{
    if( !o || !out ) return ACN_ERR_ARG;
    flat_builder b;
    memset( &b, 0, sizeof( b ) );
    memset( out, 0, sizeof( *out ) );
    out->abi_version = ACN_ABI_VERSION;
    out->light_root  = fb_add( &b, o->light );
    out->matter_root = fb_add( &b, o->matter );
    out->n_nodes = b.n_nodes;
    out->n_elems = b.n_elems;
    out->nodes = b.nodes;
    out->elems = b.elems ? b.elems : calloc( 1, sizeof( int32_t ) );
    out->params = o->prm;
    out->n_textures = b.n_tex;
    out->textures = b.tex;
    return ACN_OK;
}

int acn_obj_flatten( const acn_obj* obj, acn_flat_scene* out, int32_t* node_of_obj )
{
    acn_scene* s = acn_scene_s_create();
    acn_obj* c = acn_obj_clone( obj );
    compound_append( s->matter, c ); /* no push rules: keep the object exactly as it is */
    int st = acn_scene_s_flatten( s, out );
    if( st == ACN_OK && node_of_obj ) *node_of_obj = out->elems[ out->nodes[ out->matter_root ].child0 ];
    acn_scene_s_discard( s );
    return st;
}

void acn_flat_scene_free( acn_flat_scene* f )
{
    if( !f ) return;
    free( ( void* )f->nodes );
    free( ( void* )f->elems );
    free( ( void* )f->textures );
    f->nodes = NULL; f->elems = NULL; f->textures = NULL; f->n_nodes = f->n_elems = f->n_textures = 0;
}
