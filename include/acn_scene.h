/* acn_scene.h -- host-side (plain C) mirror of the part of Actinon's object / scene API that produces the
 * hot path's input and consumes its output.  It exists so the path can be driven, tested and benchmarked
 * without the reference's foundation library `beth` (absent offline): same function names (prefixed acn_),
 * same argument meaning, same construction rules.  Library: libactinon_host.so (links libactinon_hip.so).
 *
 * Reference surface mirrored (paths relative to /root/reference):
 *   object ctors / transforms / material presets   src/objects.c:393-476,713-776,1011-1018,1161-1176,
 *                                                   1315-1321,1388-1407,1582-1690; src/closures.c:460-593
 *   balanced CSG composites                         src/container.c:376-410
 *   compound_s push / envelope rules                src/compound.c:62-207
 *   scene_s fields, push, clear, objects            src/scene.c:153-291 ; src/scene.h:45-57
 *   lum_machine_s_run, scene_s_create_image_file    src/scene.c:1017-1165
 *   image_cps_s_write_pnm                           src/scene.c:122-137
 */
#ifndef ACN_SCENE_H
#define ACN_SCENE_H

#include "actinon_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct acn_obj acn_obj;       /* any obj_*_s or a compound_s */
typedef struct acn_scene acn_scene;   /* scene_s */

typedef struct acn_v3 { double x, y, z; } acn_v3;
typedef struct acn_m3 { acn_v3 x, y, z; } acn_m3;

/* closures.c:29-139 */
acn_m3 acn_rotx( double degrees );
acn_m3 acn_roty( double degrees );
acn_m3 acn_rotz( double degrees );

/* ---- object constructors ---- */
acn_obj* acn_obj_plane_s_create( void );                                            /* closures.c:460-466 */
acn_obj* acn_obj_sphere_s_create( double radius );                                  /* closures.c:472-480 */
acn_obj* acn_obj_squaroid_s_create_squaroid( double a, double b, double c, double r );
acn_obj* acn_obj_squaroid_s_create_ellipsoid( double rx, double ry, double rz );
acn_obj* acn_obj_squaroid_s_create_hyperboloid1( double rx, double ry, double rz );
acn_obj* acn_obj_squaroid_s_create_hyperboloid2( double rx, double ry, double rz );
acn_obj* acn_obj_squaroid_s_create_cone( double rx, double ry, double rz );
acn_obj* acn_obj_squaroid_s_create_cylinder( double rx, double ry );
acn_obj* acn_obj_torus_create( double radius1, double radius2 );                    /* closures.c:568-591 */
/* beth_object( "obj_distance_s" ) (objects.c:853-861: inv_scale 1, cycles 200, no envelope) and its set_distance_function
 * (objects.c:1691-1710) with distance_sphere_s / distance_torus_s (distance.c:30-92): sdf_kind is enum acn_sdf_kind,
 * ex_radius the torus' ex-planar radius (ignored for the sphere).  A distance object without a function is a unit sphere here
 * (the reference calls a null pointer).  ACN_ERR_ARG: o is no distance object, or the kind is unknown. */
acn_obj* acn_obj_distance_s_create( void );
int      acn_obj_set_distance_function( acn_obj* o, int sdf_kind, double ex_radius );
acn_obj* acn_obj_pair_inside_s_create_pair( const acn_obj* o1, const acn_obj* o2 );   /* script operator &  */
acn_obj* acn_obj_pair_outside_s_create_pair( const acn_obj* o1, const acn_obj* o2 );  /* script operator |  */
acn_obj* acn_obj_neg_s_create_neg( const acn_obj* o1 );                               /* script operator !  */
acn_obj* acn_obj_scale_s_create_scale( const acn_obj* o1, acn_v3 scale );             /* obj * vec          */
/* balanced trees over list[start .. start+size) ; deep-clones the elements */
acn_obj* acn_create_inside_composite( acn_obj* const* list, size_t size );
acn_obj* acn_create_outside_composite( acn_obj* const* list, size_t size );

acn_obj* acn_obj_clone( const acn_obj* o );
void     acn_obj_discard( acn_obj* o );
int      acn_obj_type( const acn_obj* o );                                           /* enum acn_node_type */

/* ---- transforms (objects and compounds) ---- */
void acn_obj_move( acn_obj* o, acn_v3 vec );
void acn_obj_rotate( acn_obj* o, const acn_m3* mat );
void acn_obj_scale( acn_obj* o, double fac );

/* ---- properties (objects.c:424-476, 1463-1690) ---- */
void acn_obj_set_color( acn_obj* o, acn_v3 color );
void acn_obj_set_transparency( acn_obj* o, acn_v3 color );
void acn_obj_set_refractive_index( acn_obj* o, double v );
void acn_obj_set_radiance( acn_obj* o, double v );
void acn_obj_set_fresnel_reflectivity( acn_obj* o, double v );
void acn_obj_set_chromatic_reflectivity( acn_obj* o, double v );
void acn_obj_set_diffuse_reflectivity( acn_obj* o, double v );
void acn_obj_set_sigma( acn_obj* o, double v );
void acn_obj_set_surface_roughness( acn_obj* o, double v );
/* obj_set_texture_field (objects.c:450-456) with txm_plain_s / txm_chess_s (textures.c) */
void acn_obj_set_texture_field_plain( acn_obj* o, acn_v3 color );
void acn_obj_set_texture_field_chess( acn_obj* o, acn_v3 color1, acn_v3 color2, double scale );
void acn_obj_clear_texture_field( acn_obj* o );
int  acn_obj_set_material( acn_obj* o, const char* name );      /* 0 ok, ACN_ERR_ARG unknown preset */
void acn_obj_set_envelope( acn_obj* o, acn_v3 pos, double radius );   /* objects and compounds */
int  acn_obj_set_auto_envelope( acn_obj* o );                   /* objects.c:470-476 / compound.c:73-107; runs on the GPU */
/* Test seam: replaces the GPU estimator behind acn_obj_set_auto_envelope (same arguments as acn_estimate_envelope,
 * scene given flat instead of resident).  The fixture generator under tests/golden/ plugs the oracle's estimator
 * in where no GPU exists; product code never sets it.  NULL restores the GPU estimator. */
typedef int ( *acn_envelope_estimator_fn )( const acn_flat_scene* scene, int32_t node, uint64_t samples, uint32_t rseed,
                                            double radius_factor, double* pos3_radius );
void acn_set_envelope_estimator( acn_envelope_estimator_fn fn );
double acn_obj_radiance( const acn_obj* o );
void   acn_obj_get_pos( const acn_obj* o, double* pos3 );              /* prp.pos */
double acn_obj_sphere_s_get_radius( const acn_obj* o );                /* 0 when o is no sphere */
/* members reachable from scripts as `obj.name`: sphere "radius"; squaroid "a" "b" "c" "r"; distance object "inv_scale",
 * "cycles" (an integer: the value is truncated). Return 1 if present. */
int    acn_obj_get_field( const acn_obj* o, const char* name, double* value );
int    acn_obj_set_field( acn_obj* o, const char* name, double value );
int  acn_obj_get_envelope( const acn_obj* o, double* pos3_radius ); /* 1 if present */

/* ---- compound_s ---- */
acn_obj* acn_compound_s_create( void );
void     acn_compound_s_push( acn_obj* compound, const acn_obj* object );   /* compound.c:140-213 (copies) */
size_t   acn_compound_s_get_size( const acn_obj* compound );
void     acn_compound_s_clear( acn_obj* compound );
/* CPU-only substitute for the Monte-Carlo estimator on leaf spheres: envelope = ( pos, radius * factor ) for every
 * sphere element (recursively) that has none. NOT reference behaviour; used where no GPU is available. */
void     acn_compound_s_set_sphere_envelopes( acn_obj* compound, double factor );

/* ---- scene_s ---- */
struct acn_scene
{
    uint64_t   threads;              /* ignored by the GPU path */
    double     gradient_threshold;
    uint64_t   gradient_samples;
    uint64_t   gradient_cycles;
    acn_params prm;                  /* the fields the hot path reads */
    acn_obj*   light;
    acn_obj*   matter;
    int        device;               /* HIP device used by lum_machine_s_run (default 0) */
};

acn_scene* acn_scene_s_create( void );              /* defaults of scene.c:185-213 */
void       acn_scene_s_discard( acn_scene* o );
void       acn_scene_s_clear( acn_scene* o );
size_t     acn_scene_s_push( acn_scene* o, const acn_obj* object );
size_t     acn_scene_s_objects( const acn_scene* o );

/* Flattening: walks light then matter in array order; fills *out (arrays owned by the returned block; free
 * with acn_flat_scene_free). */
int  acn_scene_s_flatten( const acn_scene* o, acn_flat_scene* out );
void acn_flat_scene_free( acn_flat_scene* f );
/* single object as a one-element matter compound (for estimators / unit tests) */
int  acn_obj_flatten( const acn_obj* o, acn_flat_scene* out, int32_t* node_of_obj );

/* lum_s (scene.c:682-687) and the seam */
typedef struct acn_lum { double pos_x, pos_y; double clr[3]; double weight; } acn_lum;
int acn_lum_machine_s_run( const acn_scene* scene, acn_lum* lum_arr, size_t n );

extern int acn_scene_s_overwrite_output_files_g;    /* scene.h:35  (actinon -f) */
extern int acn_scene_s_automatic_recover_g;         /* scene.h:36  (actinon -r): resume from `<file>.tmp.lum_image` */
/* Render driver: main pass + gradient cycles, writes `file` (PNM P6) after every pass. Returns acn_status.
 * SIGINT during a gradient cycle stops softly: the accumulated image goes to `<file>.tmp.lum_image` and the call
 * returns ACN_ERR_CANCELLED; a later call with acn_scene_s_automatic_recover_g resumes at that cycle. */
int acn_scene_s_create_image_file( acn_scene* o, const char* file );
/* image_cps_s_write_pnm on an RGB float image already gamma-saturated (values in [0,1]) */
int acn_write_pnm( const char* file, const double* rgb, size_t w, size_t h );
/* cps_from_cl scene.c:76-82 */
uint32_t acn_cps_from_cl( const double* cl3 );

/* ---- scenes of BASELINE.json, built by direct calls mirroring the .acn scripts ---- */
acn_scene* acn_scene_primitives( void );      /* src_acn/primitives.acn   */
acn_scene* acn_scene_wine_glass( void );      /* src_acn/wine_glass.acn   */
acn_scene* acn_scene_diamond( void );         /* src_acn/diamond.acn      */
acn_scene* acn_scene_many_spheres( int levels, int exact_envelopes ); /* src_acn/many_spheres.acn */

#ifdef __cplusplus
}
#endif
#endif
