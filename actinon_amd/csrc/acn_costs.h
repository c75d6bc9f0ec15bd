/* acn_costs.h -- the unit-cost table of SURVEY.md App. B: fp64 operations ( add / sub / mul / div / sqrt / compare-select
 * = 1 each ) and transcendental calls per EVENT of the reference's algorithm, derived by counting operations in the cited
 * reference lines.  Both the instrumented device kernels (ACN_OPT_COUNT_WORK) and the test oracle tally
 *     F_alg = sum over events of count * cost          T_alg = sum of count * transcendentals
 * with these numbers, at the same places of the algorithm, so that "achieved fp64 FLOP/s" in bench.py is counted work
 * over measured time rather than a guess.  They are definitions for bookkeeping; they need not match machine
 * instruction counts.  Costs the table does not list are marked (+). */
#ifndef ACN_COSTS_H
#define ACN_COSTS_H

#define ACN_F_PLANE_HIT          15   /* gmath.h:38-45 */
#define ACN_F_SPHERE_MISS        17   /* gmath.h:64-85: no finite offset */
#define ACN_F_SPHERE_HIT         21   /*   finite offset, no normal */
#define ACN_F_SPHERE_HIT_NOR     42   /*   finite offset + unit normal (vectors.h:148-154) */
#define ACN_F_ENV_MISS           17   /* objects.c:90-93 = sphere_ray_hit < inf */
#define ACN_F_ENV_HIT            21
#define ACN_F_SQUAROID_MISS      66   /* objects.c:778-821 */
#define ACN_F_SQUAROID_HIT       78   /* (+) finite offset, no normal */
#define ACN_F_SQUAROID_HIT_NOR  103
#define ACN_F_SDF_TORUS          20   /* distance.c:83-92 (2 sqrt) */
#define ACN_F_SDF_SPHERE          7   /* distance.c:39-42 */
#define ACN_F_SDF_RAY            45   /* objects.c:903-959 fixed part */
#define ACN_F_SDF_STEP            8   /*   per marching step, the SDF evaluation itself not included */
#define ACN_F_SDF_NORMAL         24   /* (+) forward differences + normalisation, the 4 SDF evaluations not included */
#define ACN_F_SIDE_PLANE          8   /* gmath.h:52-55 */
#define ACN_F_SIDE_SPHERE         9   /* gmath.h:93-97; also an envelope's side test */
#define ACN_F_SIDE_SQUAROID      27   /* objects.c:823-827 */
#define ACN_F_SIDE_SDF           21   /* objects.c:961-966, the SDF evaluation not included */
#define ACN_F_SIDE_SCALE         21   /* (+) objects.c:1439-1443 */
#define ACN_F_PAIR_STEP           8   /* objects.c:1057-1092: per candidate accepted or rejected */
#define ACN_F_SCALE_WRAP         52   /* objects.c:1418-1437 */
#define ACN_F_TRANS_RESOLVE       8   /* compound.c:266-296: per finite candidate */
#define ACN_F_FRESNEL_REFL       55   /* gmath.c:68-91 + vectors.h:238-241 */
#define ACN_F_FRESNEL_REFR       23   /* gmath.c:94-113 */
#define ACN_F_REFLECTION         20   /* (+) vectors.h:238-241 alone (chromatic reflection) */
#define ACN_F_CAP_SAMPLE         27   /* vectors.h:197-206 + m3d_s_mlv */
#define ACN_T_CAP_SAMPLE          2   /*   sin, cos */
#define ACN_F_FRAME              45   /* vectors.h:157-175,315-322 + transpose */
#define ACN_F_FOV                20   /* (+) objects.c:619-637 */
#define ACN_F_OREN_NAYAR         35   /* scene.c:394-416 */
#define ACN_T_OREN_NAYAR          3   /*   acos, sin, tan */
#define ACN_F_OREN_NAYAR_DIRECT  17   /* (+) the device's closed form of the same weight in the direct-light loop (oren_nayar_weight_direct): no transcendental */
#define ACN_F_SEED               18   /* vectors.h:177-190, two calls: 12 integer mul/add + 6 conversions */
#define ACN_T_SEED                6   /*   frexp */
#define ACN_F_SHADE_DIFFUSE      40   /* scene.c:526-537 */
#define ACN_T_SHADE_DIFFUSE       1   /*   acos */
#define ACN_F_DIRECT_TAIL        23   /* scene.c:571-574, unoccluded sample */
#define ACN_F_PATH_TAIL           6   /* (+) scene.c:608-616 */
#define ACN_F_ROUGHNESS          25   /* objects.c:266-282 */
#define ACN_T_ROUGHNESS           6   /*   3 log + 3 frexp */
#define ACN_F_ABSORB              3   /* scene.c:656-664 */
#define ACN_T_ABSORB              3   /*   pow */
#define ACN_F_SAT                 6   /* vectors.h:372-384 */
#define ACN_T_SAT                 3   /*   pow */
#define ACN_F_CAMERA_RAY         35   /* scene.c:980-990 */
#define ACN_F_EMISSION           12   /* (+) scene.c:432-437 */
#define ACN_F_LUM_FIXED          20   /* (+) scene.c:440-470, material set-up of a scene_s_lum call */

#endif
