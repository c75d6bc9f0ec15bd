/* actinon_hip -- command line front-end: interprets an .acn script and renders its create_image calls on the GPU.
 * Same invocation as the reference's binary (/root/reference/src/main.c:76-122):
 *     actinon_hip <script file> [-f] [-r] [script arguments ...]
 *   -f   overwrite existing output files            (scene_s_overwrite_output_files_g)
 *   -r   resume from <image>.tmp.lum_image if there (scene_s_automatic_recover_g)
 * Everything else is handed to the script as `program_args` (index 0: program, 1: script file).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acn_interp.h"

int main( int argc, const char** argv )
{
    /* the library's concurrent lanes want their streams on distinct hardware queues (ROCm maps streams onto
     * GPU_MAX_HW_QUEUES queues, default 4); set here, in the program, before anything initialises HIP */
    setenv( "GPU_MAX_HW_QUEUES", "8", 0 );
    printf( "ACTINON-HIP: ray tracer, MI355X render path.\n\n" );
    if( argc < 2 )
    {
        printf( "Usage: actinon_hip <script file> [-f] [-r]\n" );
        return 1;
    }
    const char** args = calloc( ( size_t )argc, sizeof( char* ) );
    int n = 0;
    for( int i = 0; i < 2; i++ ) args[ n++ ] = argv[ i ];
    for( int i = 2; i < argc; i++ )
    {
        if( !strcmp( argv[ i ], "-f" ) ) acn_scene_s_overwrite_output_files_g = 1;
        if( !strcmp( argv[ i ], "-r" ) ) acn_scene_s_automatic_recover_g = 1;
        else args[ n++ ] = argv[ i ];
    }
    if( acn_device_count() <= 0 )
    {
        fprintf( stderr, "No HIP device: this program has no CPU render path.\n" );
        return 2;
    }
    acn_interp_opts opts;
    memset( &opts, 0, sizeof( opts ) );
    opts.argc = n;
    opts.argv = args;
    printf( "Processing '%s'\n", argv[ 1 ] );
    int st = acn_interpret_file( argv[ 1 ], &opts );
    if( st != ACN_OK ) fprintf( stderr, "%s\n", acn_interp_last_error() );
    free( args );
    return st == ACN_OK ? 0 : 3;
}
