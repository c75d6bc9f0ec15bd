/* acn_interp.h -- interpreter for Actinon scene scripts (*.acn), the front-end that feeds the render seam.
 *
 * SURVEY.md 8 row (f-2).  The script language is defined operationally by the reference's evaluator
 * (src/interpreter.c:207-511 tokens, 1412-1730 expressions, 1734-1850 statements, 1896-1923 calls) and by the
 * per-type member tables (src/scene.c:293-331, src/container.c:156-231,423-518, src/compound.c:380-455,
 * src/objects.c:1463-1725, src/closures.c).  This implementation follows those RULES (operator binding,
 * reference/clone semantics, literal rounding) with its own data model on top of include/acn_scene.h; it
 * is host-side scene assembly, plain C, and contains no ray code.  Part of libactinon_host.so.
 */
#ifndef ACN_INTERP_H
#define ACN_INTERP_H

#include "acn_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* What `x.set_auto_envelope()` / `(@) x` do (objects.c:470-476, compound.c:73-107). */
enum
{
    ACN_AUTOENV_GPU    = 0,   /* Monte-Carlo estimator on the GPU (acn_obj_set_auto_envelope); needs libactinon_hip + a device */
    ACN_AUTOENV_SKIP   = 1,   /* leave objects without envelope (same image, slower trace); for GPU-less tooling and tests */
};

/* scene.create_image( file ) hook (scene.c:313-325).  Return an acn_status; non-zero aborts the script. */
typedef int ( *acn_create_image_fn )( void* ctx, acn_scene* scene, const char* file );

typedef struct acn_interp_opts
{
    acn_create_image_fn on_create_image;   /* NULL: acn_scene_s_create_image_file (render + write PNM) */
    void*               ctx;
    int                 auto_envelope;     /* ACN_AUTOENV_* */
    int                 readonly_fs;       /* 1: file_touch / file_delete / file_rename do nothing and return false */
    int                 argc;              /* script-visible `program_args` (main.c:84-91) */
    const char* const*  argv;
} acn_interp_opts;

/* Runs the script (mclosure_s_interpret, interpreter.c:1934-2020). Returns ACN_OK or ACN_ERR_ARG with a
 * "file:line: message" text available from acn_interp_last_error(). */
int acn_interpret_file( const char* path, const acn_interp_opts* opts );
/* Same, script text given in memory; `name` is used for messages, #source_file_name and relative #parse. */
int acn_interpret_string( const char* text, const char* name, const acn_interp_opts* opts );
const char* acn_interp_last_error( void );

/* Convenience for tools and tests: interpret `path`, do not render, return a deep copy of the scene as it was
 * at the first create_image call (NULL on error or if the script never calls it).  Runs with readonly_fs. */
acn_scene* acn_scene_from_script( const char* path, int auto_envelope );
acn_scene* acn_scene_s_clone( const acn_scene* o );

#ifdef __cplusplus
}
#endif
#endif
