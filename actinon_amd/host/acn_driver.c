/* acn_driver.c -- host render driver: the caller side of the seam (see include/acn_scene.h).
 *
 *   acn_lum_machine_s_run          <- lum_machine_s_run            src/scene.c:1017-1028 (now: one GPU call)
 *   acn_scene_s_create_image_file  <- scene_s_create_image_file    src/scene.c:1032-1165
 *   lum_image accumulation         <- lum_image_s_*                src/scene.c:744-885
 *   acn_write_pnm / acn_cps_from_cl<- image_cps_s_write_pnm, cps_from_cl   src/scene.c:76-82,122-137
 *
 *   SIGINT soft stop + recovery file <- signal_callabck, scene.c:887-895, 1046-1091, 1138-1147
 *
 * The recovery file `<image>.tmp.lum_image` holds the accumulated luminance image, the gradient cycle to resume
 * at and the position generator's state.  The reference serialises it with beth's binary markup; this driver
 * writes its own little-endian layout (struct recovery_header below), so the two programs cannot read each
 * other's recovery files.  The reference asks on stdin before overwriting or recovering; a library cannot, so
 * the two globals below decide (scene.h:35-36).
 */
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acn_scene.h"

int acn_scene_s_overwrite_output_files_g = 0;
int acn_scene_s_automatic_recover_g = 0;

static volatile int signal_received_g = 0;
static void signal_callback( int sig ) { signal_received_g = sig; }

uint32_t acn_cps_from_cl( const double* cl )
{
    uint8_t r = cl[ 0 ] > 0.0 ? cl[ 0 ] < 1.0 ? ( uint8_t )( cl[ 0 ] * 256 ) : 255 : 0;
    uint8_t g = cl[ 1 ] > 0.0 ? cl[ 1 ] < 1.0 ? ( uint8_t )( cl[ 1 ] * 256 ) : 255 : 0;
    uint8_t b = cl[ 2 ] > 0.0 ? cl[ 2 ] < 1.0 ? ( uint8_t )( cl[ 2 ] * 256 ) : 255 : 0;
    return ( uint32_t )r | ( ( uint32_t )g ) << 8 | ( ( uint32_t )b ) << 16;
}

int acn_write_pnm( const char* file, const double* rgb, size_t w, size_t h )
{
    FILE* f = fopen( file, "wb" );
    if( !f ) return ACN_ERR_ARG;
    fprintf( f, "P6\n%zu %zu\n255\n", w, h );
    uint8_t* row = malloc( w * 3 );
    for( size_t j = 0; j < h; j++ )
    {
        for( size_t i = 0; i < w; i++ )
        {
            uint32_t v = acn_cps_from_cl( rgb + ( j * w + i ) * 3 );
            row[ i * 3 + 0 ] = ( uint8_t )v;
            row[ i * 3 + 1 ] = ( uint8_t )( v >> 8 );
            row[ i * 3 + 2 ] = ( uint8_t )( v >> 16 );
        }
        fwrite( row, 1, w * 3, f );
    }
    free( row );
    fclose( f );
    return ACN_OK;
}

static int run_on_handle( acn_scene_handle* h, acn_lum* lum_arr, size_t n, const volatile int* cancel )
{
    if( n == 0 ) return ACN_OK;
    double* pos = malloc( sizeof( double ) * 2 * n );
    double* clr = malloc( sizeof( double ) * 3 * n );
    for( size_t i = 0; i < n; i++ ) { pos[ i * 2 ] = lum_arr[ i ].pos_x; pos[ i * 2 + 1 ] = lum_arr[ i ].pos_y; }
    acn_render_opts opts;
    memset( &opts, 0, sizeof( opts ) );
    opts.struct_size = ( uint32_t )sizeof( opts );
    opts.cancel = cancel;
    int st = acn_render_positions( h, pos, n, clr, &opts );
    if( st == ACN_OK )
    {
        for( size_t i = 0; i < n; i++ ) memcpy( lum_arr[ i ].clr, clr + i * 3, sizeof( double ) * 3 );
    }
    free( pos );
    free( clr );
    return st;
}

int acn_lum_machine_s_run( const acn_scene* scene, acn_lum* lum_arr, size_t n )
{
    acn_flat_scene f;
    int st = acn_scene_s_flatten( scene, &f );
    if( st != ACN_OK ) return st;
    acn_scene_handle* h = NULL;
    st = acn_scene_upload( &f, scene->device, &h );
    if( st == ACN_OK ) st = run_on_handle( h, lum_arr, n, NULL );
    if( h ) acn_scene_free( h );
    acn_flat_scene_free( &f );
    return st;
}

/* lum_image_s scene.c:744-862 */
typedef struct { size_t width, height; acn_lum* data; } lum_image;

static void lum_image_push( lum_image* o, const acn_lum* lum )   /* scene.c:804-813 */
{
    int32_t x = ( int32_t )( lum->pos_x / lum->weight );
    int32_t y = ( int32_t )( lum->pos_y / lum->weight );
    if( x >= 0 && ( size_t )x < o->width && y >= 0 && ( size_t )y < o->height )
    {
        acn_lum* d = &o->data[ ( size_t )y * o->width + x ];
        d->pos_x += lum->pos_x; d->pos_y += lum->pos_y;
        d->clr[ 0 ] += lum->clr[ 0 ]; d->clr[ 1 ] += lum->clr[ 1 ]; d->clr[ 2 ] += lum->clr[ 2 ];
        d->weight += lum->weight;
    }
}

static void lum_image_avg_clr( const lum_image* o, int64_t x, int64_t y, double* clr )   /* scene.c:824-835 */
{
    acn_lum lum;
    memset( &lum, 0, sizeof( lum ) );
    if( x >= 0 && ( size_t )x < o->width && y >= 0 && ( size_t )y < o->height ) lum = o->data[ ( size_t )y * o->width + x ];
    double f = ( lum.weight > 0 ) ? 1.0 / lum.weight : 1.0;
    clr[ 0 ] = lum.clr[ 0 ] * f; clr[ 1 ] = lum.clr[ 1 ] * f; clr[ 2 ] = lum.clr[ 2 ] * f;
}

static double lum_image_clr_dev( const lum_image* o, const double* ref, int64_t x, int64_t y )   /* scene.c:839-844 */
{
    if( x < 0 || ( size_t )x >= o->width ) return 0;
    if( y < 0 || ( size_t )y >= o->height ) return 0;
    double c[ 3 ];
    lum_image_avg_clr( o, x, y, c );
    double dx = ref[ 0 ] - c[ 0 ], dy = ref[ 1 ] - c[ 1 ], dz = ref[ 2 ] - c[ 2 ];
    return ( dx * dx ) + ( dy * dy ) + ( dz * dz );
}

static double lum_image_sqr_grad( const lum_image* o, int64_t x, int64_t y )   /* scene.c:848-862 */
{
    double g0 = 0, g1;
    double v[ 3 ];
    lum_image_avg_clr( o, x, y, v );
    static const int dx[ 8 ] = { -1, -1, -1, 0, 0, 1, 1, 1 };
    static const int dy[ 8 ] = { -1, 0, 1, -1, 1, -1, 0, 1 };
    for( int k = 0; k < 8; k++ )
    {
        g1 = lum_image_clr_dev( o, v, x + dx[ k ], y + dy[ k ] );
        g0 = g1 > g0 ? g1 : g0;
    }
    return g0;
}

static int lum_image_write( const lum_image* o, const char* file )   /* scene.c:866-885 */
{
    double* rgb = malloc( sizeof( double ) * 3 * o->width * o->height );
    for( size_t j = 0; j < o->height; j++ )
        for( size_t i = 0; i < o->width; i++ )
            lum_image_avg_clr( o, ( int64_t )i, ( int64_t )j, rgb + ( j * o->width + i ) * 3 );
    int st = acn_write_pnm( file, rgb, o->width, o->height );
    free( rgb );
    return st;
}

/* recovery file (lum_image_s: width, height, gradient_cycle, rval, data; scene.c:744-760) */
typedef struct recovery_header
{
    char     magic[ 8 ];       /* "ACNLUM1\0" */
    uint64_t width, height;
    uint64_t gradient_cycle;   /* the cycle that was interrupted = the one to redo */
    uint64_t rval;             /* generator state at the START of that cycle */
} recovery_header;

static int recovery_save( const char* path, const lum_image* img, uint64_t cycle, uint64_t rval )
{
    FILE* f = fopen( path, "wb" );
    if( !f ) return ACN_ERR_ARG;
    recovery_header hd;
    memset( &hd, 0, sizeof( hd ) );
    memcpy( hd.magic, "ACNLUM1", 8 );
    hd.width = img->width; hd.height = img->height; hd.gradient_cycle = cycle; hd.rval = rval;
    int ok = fwrite( &hd, sizeof( hd ), 1, f ) == 1 && fwrite( img->data, sizeof( acn_lum ), img->width * img->height, f ) == img->width * img->height;
    fclose( f );
    return ok ? ACN_OK : ACN_ERR_ARG;
}

/* 1: recovered, 0: not usable (size changed / not ours) */
static int recovery_load( const char* path, lum_image* img, uint64_t* cycle, uint64_t* rval )
{
    FILE* f = fopen( path, "rb" );
    if( !f ) return 0;
    recovery_header hd;
    int ok = fread( &hd, sizeof( hd ), 1, f ) == 1 && memcmp( hd.magic, "ACNLUM1", 8 ) == 0;
    if( ok && ( hd.width != img->width || hd.height != img->height ) )
    {
        printf( "Image size has changed. Starting from cycle 0.\n" );
        ok = 0;
    }
    if( ok ) ok = fread( img->data, sizeof( acn_lum ), img->width * img->height, f ) == img->width * img->height;
    fclose( f );
    if( !ok ) { memset( img->data, 0, sizeof( acn_lum ) * img->width * img->height ); return 0; }
    *cycle = hd.gradient_cycle; *rval = hd.rval;
    return 1;
}

static int file_exists( const char* path ) { FILE* f = fopen( path, "rb" ); if( f ) fclose( f ); return f != NULL; }

int acn_scene_s_create_image_file( acn_scene* o, const char* file )
{
    if( !acn_scene_s_overwrite_output_files_g && file_exists( file ) )
    {
        fprintf( stderr, "Image file '%s' exists (set acn_scene_s_overwrite_output_files_g).\n", file );
        return ACN_ERR_ARG;
    }
    size_t tl = strlen( file ) + 32;
    char* tmp_file = malloc( tl );
    snprintf( tmp_file, tl, "%s.tmp.lum_image", file );
    printf( "Number of objects: %zu\n", acn_scene_s_objects( o ) );

    acn_flat_scene flat;
    int st = acn_scene_s_flatten( o, &flat );
    if( st != ACN_OK ) { free( tmp_file ); return st; }
    acn_scene_handle* h = NULL;
    st = acn_scene_upload( &flat, o->device, &h );
    if( st != ACN_OK ) { acn_flat_scene_free( &flat ); free( tmp_file ); return st; }

    size_t w = o->prm.image_width, hgt = o->prm.image_height;
    lum_image img = { w, hgt, calloc( w * hgt, sizeof( acn_lum ) ) };
    uint64_t rval = 21943294;   /* scene.c:799 */
    uint64_t first_cycle = 0;
    double sqr_gradient_threshold = o->gradient_threshold * o->gradient_threshold;

    if( file_exists( tmp_file ) )   /* scene.c:1069-1091 */
    {
        if( acn_scene_s_automatic_recover_g )
        {
            if( recovery_load( tmp_file, &img, &first_cycle, &rval ) ) printf( "Recovered from file %s: resuming at gradient cycle %llu\n", tmp_file, ( unsigned long long )first_cycle );
            else { first_cycle = 0; rval = 21943294; }
        }
        else printf( "Recovery file %s ignored (set acn_scene_s_automatic_recover_g / -r to resume from it).\n", tmp_file );
    }

    signal_received_g = 0;
    void ( *old_handler )( int ) = signal( SIGINT, signal_callback );

    acn_lum* arr = NULL;
    size_t arr_space = 0;
    printf( "Rendering ...\n" );
    for( uint64_t cycle = first_cycle; cycle <= o->gradient_cycles && st == ACN_OK; cycle++ )
    {
        uint64_t rval_at_start = rval;
        size_t n = 0;
#define PUSH_POS( px, py ) do { \
            if( n == arr_space ) { arr_space = arr_space ? arr_space * 2 : 256; arr = realloc( arr, sizeof( acn_lum ) * arr_space ); } \
            memset( &arr[ n ], 0, sizeof( acn_lum ) ); arr[ n ].pos_x = ( px ); arr[ n ].pos_y = ( py ); arr[ n ].weight = 1.0; n++; } while( 0 )
        if( cycle == 0 )
        {
            printf( "\n\tmain image: " );
            for( size_t j = 0; j < hgt; j++ ) for( size_t i = 0; i < w; i++ ) PUSH_POS( i + 0.5, j + 0.5 );
        }
        else
        {
            printf( "\n\tgradient pass %3llu: ", ( unsigned long long )cycle );
            for( int64_t j = 0; j < ( int64_t )hgt; j++ )
            {
                for( int64_t i = 0; i < ( int64_t )w; i++ )
                {
                    if( lum_image_sqr_grad( &img, i, j ) > sqr_gradient_threshold )
                    {
                        for( uint64_t k = 0; k < o->gradient_samples; k++ )
                        {
                            /* f3_rnd1 vectors.h:48 with the declared lcg00 */
                            double dx = ( double )( rval = rval * ACN_LCG00_A + ACN_LCG00_C ) * ( 1.0 / 0xFFFFFFFFFFFFFFFFull );
                            double dy = ( double )( rval = rval * ACN_LCG00_A + ACN_LCG00_C ) * ( 1.0 / 0xFFFFFFFFFFFFFFFFull );
                            PUSH_POS( i + dx, j + dy );
                        }
                    }
                }
            }
        }
#undef PUSH_POS
        fflush( stdout );
        st = run_on_handle( h, arr, n, &signal_received_g );
        if( st == ACN_ERR_CANCELLED || signal_received_g == SIGINT )   /* scene.c:1138-1147 */
        {
            printf( "\nSIGINT received\n" );
            st = ACN_ERR_CANCELLED;
            if( cycle > 0 )
            {
                printf( "Saving result from last gradient cycle to file %s\n", tmp_file );
                if( recovery_save( tmp_file, &img, cycle, rval_at_start ) != ACN_OK ) fprintf( stderr, "Cannot write %s\n", tmp_file );
            }
            break;
        }
        if( st != ACN_OK ) break;
        for( size_t i = 0; i < n; i++ ) lum_image_push( &img, &arr[ i ] );
        st = lum_image_write( &img, file );
    }
    printf( "\n" );
    signal( SIGINT, old_handler == SIG_ERR ? SIG_DFL : old_handler );
    free( tmp_file );
    free( arr );
    free( img.data );
    acn_scene_free( h );
    acn_flat_scene_free( &flat );
    return st;
}
