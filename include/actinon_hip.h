/* actinon_hip.h -- C ABI of libactinon_hip.so: the MI355X (gfx950) replacement for Actinon's
 * per-sample trace/radiance path.
 *
 * Drop-in seam (reference, all paths relative to /root/reference):
 *   void lum_machine_s_run( const scene_s* scene, lum_arr_s* lum_arr )      src/scene.c:1017-1028
 * called from exactly one place, scene_s_create_image_file (src/scene.c:1141).  The reference has no
 * FFI layer; this header promotes that internal function boundary to a C ABI:
 *
 *   reference input                                  ->  this ABI
 *   scene_s render fields (src/scene.c:153-183)      ->  acn_params
 *   scene->light / scene->matter compound_s graphs   ->  acn_flat_scene.nodes / .elems (POD, index-linked)
 *   lum_arr->data[i].pos  (v2d_s, src/scene.c:682)   ->  pos_xy[i*2 .. i*2+1]
 *   lum_arr->data[i].clr  (cl_s after cl_s_sat)      ->  out_rgb[i*3 .. i*3+2]
 *
 * Everything is plain pointers + sizes; no torch / HIP types appear in a signature (streams and device
 * pointers are passed as void*).  All floating point is IEEE binary64, as in the reference (f3_t).
 */
#ifndef ACTINON_HIP_H
#define ACTINON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACN_ABI_VERSION 2

/* ------------------------------------------------------------------------------------------------------------------ */
/* Flattened scene.  One acn_node per reference object (obj_*_s, src/objects.c) or compound_s (src/compound.c:36-50).
 * Pointers of the reference become int32 indices into nodes[]; a compound's element array becomes a slice of elems[]. */

enum acn_node_type
{
    ACN_PLANE        = 1, /* obj_plane_s         src/objects.c:480-547   */
    ACN_SPHERE       = 2, /* obj_sphere_s        src/objects.c:552-661   */
    ACN_SQUAROID     = 3, /* obj_squaroid_s      src/objects.c:669-831   */
    ACN_DISTANCE     = 4, /* obj_distance_s      src/objects.c:836-970   */
    ACN_PAIR_INSIDE  = 5, /* obj_pair_inside_s   src/objects.c:975-1120  */
    ACN_PAIR_OUTSIDE = 6, /* obj_pair_outside_s  src/objects.c:1125-1277 */
    ACN_NEG          = 7, /* obj_neg_s           src/objects.c:1282-1348 */
    ACN_SCALE        = 8, /* obj_scale_s         src/objects.c:1353-1459 */
    ACN_COMPOUND     = 9  /* compound_s          src/compound.c:36-50    */
};

enum acn_sdf_kind
{
    ACN_SDF_SPHERE = 0,   /* distance_sphere_s_call  src/distance.c:39-42 */
    ACN_SDF_TORUS  = 1    /* distance_torus_s_call   src/distance.c:83-92 */
};

#define ACN_NODE_HAS_ENVELOPE 1u

/* 304 bytes, 16-byte aligned.  prp fields follow properties_s (src/objects.h:51-78). */
typedef struct acn_node
{
    int32_t  type;        /* enum acn_node_type */
    uint32_t flags;       /* ACN_NODE_HAS_ENVELOPE */
    int32_t  child0;      /* pair: o1 | neg, scale: o1 | compound: first index into elems[] | else -1 */
    int32_t  child1;      /* pair: o2 | compound: number of elements | else -1 */
    int32_t  sdf_kind;    /* ACN_DISTANCE: enum acn_sdf_kind */
    int32_t  cycles;      /* ACN_DISTANCE: obj_distance_s.cycles */
    int32_t  texture;     /* prp.texture_field: index into acn_flat_scene.textures, -1 = none */
    int32_t  reserved;

    double pos[3];        /* prp.pos */
    double rax[9];        /* prp.rax, rows x,y,z */
    double env_pos[3];    /* prp.envelope->pos   (compound: compound_s.envelope) */
    double env_radius;    /* prp.envelope->radius */
    double prm[4];        /* sphere: radius,-,-,- | squaroid: a,b,c,r | distance: inv_scale,ex_radius,-,- | scale: inv_scale.xyz,- */

    double color[3];      /* prp.color */
    double radiance;
    double refractive_index;
    double fresnel_reflectivity;
    double chromatic_reflectivity;
    double diffuse_reflectivity;
    double sigma;
    double surface_roughness;
    double transparency[3];
    double pad_;
} acn_node;

/* Texture fields (src/textures.c): the colour of an object's surface as a function of position, obj_color
 * src/objects.c:411-422.  ACN_TXM_CHESS uses obj_projection, which only planes (objects.c:514-518), spheres (:602-617)
 * and distance objects (:893-896) implement; on any other object type it is rejected at upload. */
enum acn_texture_kind
{
    ACN_TXM_PLAIN = 0,    /* txm_plain_s src/textures.c:62-102 : color1 */
    ACN_TXM_CHESS = 1     /* txm_chess_s src/textures.c:118-148: ( llrint( p.x*scale ) ^ llrint( p.y*scale ) ) & 1 ? color1 : color2 */
};

typedef struct acn_texture
{
    int32_t kind;
    int32_t reserved;
    double  color1[3];
    double  color2[3];
    double  scale;
} acn_texture;

/* Render parameters: the scene_s fields the hot path reads (src/scene.c:153-183; defaults :185-213). */
typedef struct acn_params
{
    uint64_t image_width;
    uint64_t image_height;
    double   gamma;
    double   background_color[3];
    double   camera_position[3];
    double   camera_view_direction[3];
    double   camera_top_direction[3];
    double   camera_focal_length;
    uint64_t trace_depth;
    double   trace_min_intensity;
    uint64_t direct_samples;
    uint64_t path_samples;
    double   max_path_length;
    int64_t  experimental_level;   /* must be 0 (src/scene.c:1000-1007) */
} acn_params;

typedef struct acn_flat_scene
{
    uint32_t        abi_version;   /* ACN_ABI_VERSION */
    uint32_t        n_nodes;
    uint32_t        n_elems;
    int32_t         light_root;    /* node index of scene->light  (an ACN_COMPOUND) */
    int32_t         matter_root;   /* node index of scene->matter (an ACN_COMPOUND) */
    uint32_t        reserved;
    const acn_node* nodes;
    const int32_t*  elems;
    acn_params      params;
    uint32_t        n_textures;
    uint32_t        reserved2;
    const acn_texture* textures;
} acn_flat_scene;

/* The 64-bit LCG triple the reference takes from beth (bcore_lcg00/01/02_u3; src/vectors.h:45-48,185-189).
 * beth is not available offline, so the constants are DECLARED here; stream-exact parity with an upstream
 * build is therefore unpinned. x' = a*x + c (mod 2^64). */
#ifndef ACN_LCG00_A   /* (the statistical pin of the test oracle builds it with other triples too: tests/test_reference_images.py) */
#define ACN_LCG00_A 6364136223846793005ull
#define ACN_LCG00_C 1442695040888963407ull
#define ACN_LCG01_A 2862933555777941757ull
#define ACN_LCG01_C 3037000493ull
#define ACN_LCG02_A 3202034522624059733ull
#define ACN_LCG02_C 4354685564936845319ull
#endif

/* ------------------------------------------------------------------------------------------------------------------ */
/* Render call */

#define ACN_OPT_LINEAR_OUT 1u   /* skip cl_s_sat (src/vectors.h:372-384): caller accumulates / reduces first */
#define ACN_OPT_COUNT_WORK 2u   /* run the instrumented kernels: acn_last_counters() reports rays / samples / hit tests */
#define ACN_OPT_STAGE_TIMING 4u /* record HIP events around every launch: acn_last_stage_ms() reports per-stage times */

/* Sharding of one call over the ranks of a multi-GPU job (one process per GPU).  The reference has one process and
 * no counterpart; what shards is what its pixel farm makes independent (src/scene.c:976-1011): sample positions, and
 * inside a position the iterations of the outermost sample loops (src/scene.c:556,596).
 *   ACN_SHARD_SAMPLES  every rank renders every position of the call, but of each OUTERMOST direct-light loop and path
 *                      loop only the iterations [ n * rank / world, n * ( rank + 1 ) / world ) -- n, the normalisation
 *                      2 * cyl_hgt / n resp. 2 / n and the LCG stream position of an iteration stay those of the whole
 *                      loop.  Terms that are under no such loop (emission and background reached through specular
 *                      chains, src/scene.c:432-437,488-491,511-514,648-651; a camera ray that hits nothing) come from
 *                      rank 0 alone.  Out: LINEAR partial radiance (use ACN_OPT_LINEAR_OUT); the caller sum-reduces the
 *                      ranks' buffers and then applies acn_resolve_dev.  For few positions with many samples.
 * Whole positions are sharded with the acn_shard_tile_* functions below (disjoint supports, no floating-point reduce). */
#define ACN_SHARD_NONE    0u
#define ACN_SHARD_SAMPLES 1u

typedef struct acn_render_opts
{
    uint32_t flags;
    uint32_t struct_size;          /* sizeof( acn_render_opts ) as the CALLER was compiled; 0 = the base layout of this struct
                                      (ACN_RENDER_OPTS_BASE_SIZE = 40 bytes: flags .. reserved2 -- a zero-initialised struct with
                                      the shard members set keeps its meaning).  The library reads no member beyond it, so a host
                                      built against an older header keeps working when members are appended (ACN_RENDER_OPTS_INIT) */
    const volatile int* cancel;    /* optional; polled between launches; the SIGINT flag of src/scene.c:893,978 */
    void*    stream;               /* optional hipStream_t; NULL = the handle's own stream */
    uint32_t shard_mode;           /* ACN_SHARD_* */
    uint32_t shard_rank;           /* 0 .. shard_world - 1 */
    uint32_t shard_world;          /* 0 or 1: the call is not sharded */
    uint32_t reserved2;
} acn_render_opts;
#define ACN_RENDER_OPTS_BASE_SIZE 40
#define ACN_RENDER_OPTS_INIT { 0u, ( uint32_t )sizeof( acn_render_opts ), 0, 0, ACN_SHARD_NONE, 0u, 0u, 0u }

typedef struct acn_scene_handle acn_scene_handle;

enum acn_status
{
    ACN_OK              =  0,
    ACN_ERR_ARG         = -1,  /* malformed scene / argument */
    ACN_ERR_UNSUPPORTED = -2,  /* experimental_level != 0 (src/scene.c:1004-1007), projection-less chess texture, too-deep CSG */
    ACN_ERR_NO_FOV      = -3,  /* light object without fov function (src/objects.c:254-258) */
    ACN_ERR_DEVICE      = -4,  /* HIP failure, no GPU */
    ACN_ERR_CANCELLED   = -5
};

/* Number of visible HIP devices (0 if none). */
int acn_device_count( void );

/* Validates the flat scene, converts it to the device layout and makes it resident on `device`.
 * Replaces nothing in the reference (which traverses host pointers); it is the price of the seam. */
int acn_scene_upload( const acn_flat_scene* scene, int device, acn_scene_handle** out );
void acn_scene_free( acn_scene_handle* h );

/* lum_machine_s_run counterpart on host buffers (src/scene.c:1017): out_rgb[i] = cl_s_sat( lum( pos_xy[i] ) ).
 * Synchronous. Returns acn_status. */
int acn_render_positions( acn_scene_handle* h, const double* pos_xy, size_t n, double* out_rgb,
                          const acn_render_opts* opts );

/* Same on device-resident buffers (d_pos_xy, d_out_rgb are device pointers on the handle's device).
 * Work is enqueued on opts->stream (NULL = the handle's own stream, then the call also waits for completion).
 * The call synchronises that stream once per chunk of positions (queue-overflow check); on return the last kernels may
 * still be in flight on a caller-provided stream. */
int acn_render_positions_dev( acn_scene_handle* h, const void* d_pos_xy, size_t n, void* d_out_rgb,
                              const acn_render_opts* opts );

/* Main-pass helper: pos = (i+0.5, j+0.5) row-major (src/scene.c:1110-1119) generated on the device,
 * for the pixel sub-range [first, first+count) of the image_width x image_height raster. */
int acn_render_main_pass_dev( acn_scene_handle* h, size_t first, size_t count, void* d_out_rgb,
                              const acn_render_opts* opts );

/* Sharding of whole positions: the n positions of a call (or pixels of a frame) are cut into tiles of ACN_SHARD_TILE
 * consecutive positions dealt round-robin to the ranks -- interleaving balances sky, floor and glass between them.
 * These three are plain arithmetic (no GPU): */
#define ACN_SHARD_TILE 256
/* number of positions rank `rank` of `world` owns */
size_t acn_shard_tile_count( size_t n, uint32_t rank, uint32_t world );
/* common length of the ranks' parts for an all-gather: the largest count, i.e. ceil( ceil( n / TILE ) / world ) * TILE */
size_t acn_shard_tile_padded( size_t n, uint32_t world );
/* index in [ 0, n ) of the i-th position of the rank's part (i < acn_shard_tile_count) */
size_t acn_shard_tile_index( size_t n, uint32_t rank, uint32_t world, size_t i );

/* Main pass of rank `rank`: renders the rank's tiles of the pixel range [ first, first + count ) (acn_render_main_pass_dev's
 * positions) into d_part, [ acn_shard_tile_padded( count, world ) ][ 3 ] f64, part order, zero behind the rank's count. */
int acn_render_main_pass_shard_dev( acn_scene_handle* h, size_t first, size_t count, uint32_t rank, uint32_t world,
                                    void* d_part, const acn_render_opts* opts );
/* After the all-gather of the ranks' parts (d_gathered: [ world ][ padded ][ 3 ] f64, rank-major): the frame
 * d_frame[ count ][ 3 ] in position order.  Every value is copied, none is added: bit-identical to one GPU. */
int acn_shard_unpack_dev( acn_scene_handle* h, const void* d_gathered, size_t count, uint32_t world, void* d_frame,
                          const acn_render_opts* opts );

/* cl_s_sat (src/vectors.h:372-384) + cps_from_cl (src/scene.c:76-82) on a device-resident LINEAR radiance buffer,
 * e.g. after the cross-GPU sum-reduce: d_out_rgb (nullable) receives the gamma-saturated colours [n][3] f64,
 * d_out_rgb8 (nullable) the packed 8-bit pixels [n][3] u8 in PNM order. In-place (d_out_rgb == d_linear_rgb) allowed. */
int acn_resolve_dev( acn_scene_handle* h, const void* d_linear_rgb, size_t n, void* d_out_rgb, void* d_out_rgb8,
                     const acn_render_opts* opts );

/* Timing of the kernels of the last render call on this handle (HIP events on the launch stream), ms. */
int acn_last_kernel_ms( acn_scene_handle* h, double* trace_ms );

/* Per-stage device time of the last render call (HIP events on the launch stream; the per-stage values [0..2], [13]
 * are zero unless the call had ACN_OPT_STAGE_TIMING, the total [3] is always measured; when the call ran on concurrent
 * lanes (ACN_LANES, default 6 for large calls) the per-stage values are SUMS over the lanes and can exceed the total)
 * and pipeline statistics:
 * out[0] walk kernels ms, [1] shade kernels ms, [2] finalize ms, [3] total ms, [4..6] launches per stage, [7] chunks,
 * [8] overflow retries, [9] path levels run, [10] peak shading tasks, [11] peak child hits, [12] slots of the largest queue,
 * [13] hard-ray kernels ms, [14] their launches, [15] hard rays, [16] rays traced by the specular walk (camera rays
 * included), [17] path-sample hits shaded (levels >= 1), [18] host synchronisations inside the pipeline (one per chunk),
 * [19] 64-ray steps of the walk kernel's waves ([16] / ( 64 * [19] ) is its lane occupancy), [20] ACN_FLAG_* bits seen
 * (8: a pixel contribution exceeded the fixed-point clamp of 16384), [21] rays the walk finished on the waves' private
 * stacks instead of in generation passes, [22] specular rays whose radiance is zero on a hit (depth 0 or intensity below
 * trace_min_intensity, src/scene.c:430) and that were therefore answered by an any-hit probe instead of a walk, [23] bytes
 * of device memory the call's work queues and ray stacks occupy (all lanes), [24] how often they were (re)allocated
 * since the upload. n <= 25. */
int acn_last_stage_ms( acn_scene_handle* h, double* out, int n );

/* Work counters of the last render call (rays cast, node visits ...), see DESIGN.md: [0..7] events, [8] flop and
 * [9] transcendental calls by the cost table (ACN_OPT_COUNT_WORK).  [10 + 16 * kernel + phase] (kernel 0 walk, 1 hard
 * shadow, 2 hard path, 3 shade): shader-clock ticks the kernel's waves spent per phase, filled only by a diagnostic build
 * of the library (EXTRA_DEFS=-DACN_PHASE_TIMERS), zero otherwise. n <= 74. */
int acn_last_counters( acn_scene_handle* h, uint64_t* out, int n );

/* MC bounding-sphere estimate of src/objects.c:312-363 for node `node` of an uploaded scene (GPU). */
int acn_estimate_envelope( acn_scene_handle* h, int32_t node, uint64_t samples, uint32_t rseed,
                           double radius_factor, double* out_pos3_radius );

/* Test hook: evaluates the deterministic fp64 kernels of csrc/acn_detmath.h on the device.
 * op: 0 sin 1 cos 2 tan 3 acos 4 log 5 exp 6 pow(x,y) 7 sqrt 8 div(x/y) 9 u64->f64 (x bits) 10 frexp-mantissa */
int acn_detmath_eval( int device, int op, const double* x, const double* y, double* out, size_t n );

const char* acn_last_error( void );

#ifdef __cplusplus
}
#endif

#endif /* ACTINON_HIP_H */
