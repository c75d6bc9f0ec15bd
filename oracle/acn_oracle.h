/* acn_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain recursive C, fp64) of Actinon's trace/radiance hot path, operating on the same
 * flattened scene the GPU library consumes (include/actinon_hip.h).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this; the product (actinon_amd/, libactinon_hip.so,
 * libactinon_host.so) never links, imports or calls it.
 *
 * PARITY STATUS: the reference (johsteffens/actinon) cannot be built here (it needs the absent library
 * `beth`; building it behind hand-written stand-ins is not allowed), it ships no tests, golden vectors or
 * fixtures, and its Monte-Carlo LCG constants live in beth.  This oracle is therefore pinned only
 *   (a) statistically, against block means of the reference's own shipped renders (image/NAME.png ->
 *       tests/golden/ref_image_blocks.json, tests/test_oracle_vs_reference_images.py), and
 *   (b) by the three known-answer values SURVEY.md 8(c) recorded from the verbatim-compiled gmath.c.
 * Bit-level / sample-stream parity with an upstream build is UNPINNED.
 *
 * Arithmetic: IEEE binary64, expressions in the reference's order, no FP contraction.  Transcendentals come
 * from actinon_amd/csrc/acn_detmath.h (default; bit-identical to the GPU) or, with -DACN_ORACLE_LIBM, from
 * libm as in the reference.
 */
#ifndef ACN_ORACLE_H
#define ACN_ORACLE_H

#include "actinon_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* event counters */
enum
{
    ORC_N_LUM = 0,        /* scene_s_lum calls */
    ORC_N_TRANS_RAY,      /* compound_s_ray_trans_hit calls on a root compound */
    ORC_N_SHADOW_RAY,     /* compound_s_ray_hit(matter) occlusion tests */
    ORC_N_OBJ_HIT,        /* obj_ray_hit calls (all levels) */
    ORC_N_ENV_TEST,       /* envelope_s_ray_hits / envelope side tests */
    ORC_N_PLANE_HIT,
    ORC_N_SPHERE_HIT,     /* sphere objects only */
    ORC_N_SQUAROID_HIT,
    ORC_N_SDF_RAY,
    ORC_N_SDF_EVAL,
    ORC_N_PAIR_HIT,
    ORC_N_SIDE,           /* obj_side calls (all levels) */
    ORC_N_CAP_SAMPLE,
    ORC_N_OREN_NAYAR,
    ORC_N_FRESNEL,
    ORC_N_NODE_VISIT,     /* nodes touched by ray traversal */
    ORC_N_FLOP,           /* F_alg: sum of event costs (actinon_amd/csrc/acn_costs.h) */
    ORC_N_TRANSC,         /* T_alg: transcendental calls */
    ORC_N_COUNTERS
};

/* out_rgb[i] = cl_s_sat( lum( pos_xy[i] ) ); flags: ACN_OPT_LINEAR_OUT. threads >= 1 (pthread pixel farm,
 * src/scene.c:1017-1028). counters (nullable): ORC_N_COUNTERS sums over all positions. Returns acn_status. */
int acn_oracle_render_positions( const acn_flat_scene* scene, const double* pos_xy, size_t n, double* out_rgb,
                                 uint32_t flags, int threads, uint64_t* counters );

/* the same for rank `rank` of `world` of a sample-sharded call (ACN_SHARD_SAMPLES, include/actinon_hip.h): linear
 * partial radiance; the sum over the ranks is the unsharded result up to floating-point reassociation */
int acn_oracle_render_positions_shard( const acn_flat_scene* scene, const double* pos_xy, size_t n, double* out_rgb,
                                       uint32_t flags, int threads, uint64_t* counters, uint32_t rank, uint32_t world );

/* obj_estimate_envelope (src/objects.c:312-363): out = pos[3], radius */
int acn_oracle_estimate_envelope( const acn_flat_scene* scene, int32_t node, uint64_t samples, uint32_t rseed,
                                  double radius_factor, double* out_pos3_radius );

/* leaf functions exposed for known-answer tests */
double acn_oracle_sphere_ray_hit( const double* pos3, double r, const double* ray_p3, const double* ray_d3, double* nor3 );
double acn_oracle_plane_ray_hit( const double* pos3, const double* nor3, const double* ray_p3, const double* ray_d3 );
double acn_oracle_fresnel_reflection( const double* dir3, const double* exit_nor3, double trix, double* out_dir3 );
void   acn_oracle_fresnel_refraction( const double* dir3, const double* exit_nor3, double trix, double* out_dir3 );
double acn_oracle_obj_ray_hit( const acn_flat_scene* scene, int32_t node, const double* ray_p3, const double* ray_d3, double* nor3 );
int    acn_oracle_obj_side( const acn_flat_scene* scene, int32_t node, const double* pos3 );
double acn_oracle_trans_hit( const acn_flat_scene* scene, const double* ray_p3, const double* ray_d3,
                             double* exit_nor3, int32_t* exit_obj, int32_t* enter_obj );
uint64_t acn_oracle_random_seed( const double* v3, uint64_t rv );
void   acn_oracle_sphere_cap( uint64_t* rv, double h, double* out3 );
/* 0: detmath, 1: libm */
int    acn_oracle_math_mode( void );

#ifdef __cplusplus
}
#endif
#endif
