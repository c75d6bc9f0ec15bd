/* acn_scenes.c -- the BASELINE.json scenes, built by direct calls that mirror the reference's .acn scripts
 * statement by statement (the .acn interpreter itself is out of scope for the hot path).
 *
 * Script semantics relied on (src/interpreter.c): `def x = e` clones (:1660); `obj + vec` / `obj - vec` clone and
 * move (:818-924, :1708); `obj * num|rot` clone and scale|rotate, `obj * vec` wraps in obj_scale_s (:765-768);
 * `*` `/` evaluate left to right (:1696-1697), `+ - & | :` evaluate their right side first, i.e. are
 * right-associative (:1708-1715); a prefix `!` `-` binds to the next operand only (:1680-1690);
 * `.move/.rotate/.scale/.set_*` mutate in place; `list.push(x)` stores a copy; pushing a list to the scene
 * pushes its elements in order, recursively (src/scene.c:269-277).
 */
#include <stdlib.h>

#include "acn_scene.h"

static acn_v3 vec( double x, double y, double z ) { acn_v3 v = { x, y, z }; return v; }
static acn_v3 vecx( double v ) { return vec( v, 0, 0 ); }
static acn_v3 vecy( double v ) { return vec( 0, v, 0 ); }
static acn_v3 vecz( double v ) { return vec( 0, 0, v ); }
/* `o - vec`: add( o, (-1) * vec ) */
static acn_v3 neg( acn_v3 v ) { return vec( v.x * -1, v.y * -1, v.z * -1 ); }

/* expression temporaries: every helper returns a NEW object and leaves its arguments alone */
static acn_obj* add_v( const acn_obj* o, acn_v3 v ) { acn_obj* r = acn_obj_clone( o ); acn_obj_move( r, v ); return r; }
static acn_obj* sub_v( const acn_obj* o, acn_v3 v ) { return add_v( o, neg( v ) ); }
static acn_obj* mul_f( const acn_obj* o, double f ) { acn_obj* r = acn_obj_clone( o ); acn_obj_scale( r, f ); return r; }
static acn_obj* mul_m( const acn_obj* o, acn_m3 m ) { acn_obj* r = acn_obj_clone( o ); acn_obj_rotate( r, &m ); return r; }

/* consuming variants: free the temporaries passed in */
static acn_obj* AND( acn_obj* a, acn_obj* b ) { acn_obj* r = acn_obj_pair_inside_s_create_pair( a, b ); acn_obj_discard( a ); acn_obj_discard( b ); return r; }
static acn_obj* OR( acn_obj* a, acn_obj* b ) { acn_obj* r = acn_obj_pair_outside_s_create_pair( a, b ); acn_obj_discard( a ); acn_obj_discard( b ); return r; }
static acn_obj* NOT( acn_obj* a ) { acn_obj* r = acn_obj_neg_s_create_neg( a ); acn_obj_discard( a ); return r; }
static acn_obj* C( const acn_obj* a ) { return acn_obj_clone( a ); }
static void set_envelope_sphere( acn_obj* o, acn_obj* sphere_tmp )   /* x.set_envelope( <sphere expr> ) objects.c:1527-1533 */
{
    acn_flat_scene f; int32_t n;
    acn_obj_flatten( sphere_tmp, &f, &n );
    acn_obj_set_envelope( o, vec( f.nodes[ n ].pos[ 0 ], f.nodes[ n ].pos[ 1 ], f.nodes[ n ].pos[ 2 ] ), f.nodes[ n ].prm[ 0 ] );
    acn_flat_scene_free( &f );
    acn_obj_discard( sphere_tmp );
}

static void scene_push_take( acn_scene* s, acn_obj* o ) { acn_scene_s_push( s, o ); acn_obj_discard( o ); }

/* a script list ( [] ) of objects */
typedef struct { acn_obj* d[ 128 ]; size_t n; } list_t;
static void list_push( list_t* l, acn_obj* owned ) { l->d[ l->n++ ] = owned; }
static void list_move( list_t* l, acn_v3 v ) { for( size_t i = 0; i < l->n; i++ ) acn_obj_move( l->d[ i ], v ); }
static void list_rotate( list_t* l, acn_m3 m ) { for( size_t i = 0; i < l->n; i++ ) acn_obj_rotate( l->d[ i ], &m ); }
static void list_clear( list_t* l ) { for( size_t i = 0; i < l->n; i++ ) acn_obj_discard( l->d[ i ] ); l->n = 0; }

/* ================================================================================================================== */
/* src_acn/primitives.acn */
acn_scene* acn_scene_primitives( void )
{
    acn_scene* scene = acn_scene_s_create();
    scene->threads = 30;
    scene->prm.image_width = 400;
    scene->prm.image_height = 400;
    scene->prm.gamma = 1.0;
    scene->gradient_cycles = 30;
    scene->gradient_samples = 2;
    scene->gradient_threshold = 0.03;
    scene->prm.trace_depth = 25;
    scene->prm.trace_min_intensity = 0.03;
    scene->prm.direct_samples = 30;
    scene->prm.path_samples = 30;
    scene->prm.max_path_length = 1;
    double cam[ 3 ] = { 0, -10, 0 };
    for( int k = 0; k < 3; k++ ) { scene->prm.camera_position[ k ] = cam[ k ]; scene->prm.camera_view_direction[ k ] = 0 + cam[ k ] * -1; }
    scene->prm.camera_top_direction[ 2 ] = 1;
    scene->prm.camera_focal_length = 4;
    scene->prm.background_color[ 0 ] = 0.4; scene->prm.background_color[ 1 ] = 0.4; scene->prm.background_color[ 2 ] = 0.4;

    /* create_light( 0.5, 30 ) + vec( 0, -4, 4 ) */
    {
        acn_obj* sph = acn_obj_sphere_s_create( 1.0 );   /* def sph = obj_sphere_s; default radius 1.0 */
        acn_obj* light = mul_f( sph, 0.5 );
        acn_obj_set_radiance( light, 30 );
        scene_push_take( scene, add_v( light, vec( 0, -4, 4 ) ) );
        acn_obj_discard( light ); acn_obj_discard( sph );
    }
    /* create_floor( -1 ) */
    {
        acn_obj* plane = acn_obj_plane_s_create();
        acn_obj_set_material( plane, "diffuse_polished" );
        acn_obj_set_color( plane, vec( 0.6, 0.4, 0.2 ) );
        acn_obj_set_refractive_index( plane, 1.2 );
        acn_obj_move( plane, vec( 0, 0, -1 ) );
        scene_push_take( scene, plane );
    }
    /* create_matter() */
    {
        acn_obj* sph = acn_obj_sphere_s_create( 0.5 );
        acn_obj* el1 = acn_obj_squaroid_s_create_ellipsoid( 0.3, 0.3, 0.5 );
        acn_obj* el2 = acn_obj_squaroid_s_create_ellipsoid( 0.5, 0.5, 0.3 );
        acn_obj* tor0 = acn_obj_torus_create( 0.35, 0.15 );
        acn_obj* tor = mul_m( tor0, acn_rotx( 90 ) );
        acn_obj* cyl = acn_obj_squaroid_s_create_cylinder( 0.4, 0.4 );
        acn_obj* cne = acn_obj_squaroid_s_create_cone( 0.2, 0.2, 2 );
        acn_obj* hyp = acn_obj_squaroid_s_create_hyperboloid1( 0.2, 0.2, 1 );

        list_t set = { { 0 }, 0 };
        list_push( &set, C( sph ) );
        list_push( &set, add_v( el1, vecz( 1.1 ) ) );
        list_move( &set, vecx( -1.1 ) );
        list_push( &set, C( tor ) );
        list_push( &set, add_v( el2, vecz( 1 ) ) );
        list_move( &set, vecx( -1 ) );
        list_push( &set, C( hyp ) );
        list_move( &set, vecx( -0.9 ) );
        list_push( &set, C( cne ) );
        list_move( &set, vecx( -1 ) );
        list_push( &set, C( cyl ) );
        for( size_t i = 0; i < set.n; i++ )
        {
            acn_obj_set_material( set.d[ i ], "diffuse_polished" );
            acn_obj_set_color( set.d[ i ], vec( 0.6, 0.7, 0.8 ) );
        }
        list_move( &set, vecx( 2 ) );
        for( size_t i = 0; i < set.n; i++ ) acn_scene_s_push( scene, set.d[ i ] );
        list_clear( &set );
        acn_obj_discard( sph ); acn_obj_discard( el1 ); acn_obj_discard( el2 ); acn_obj_discard( tor0 ); acn_obj_discard( tor );
        acn_obj_discard( cyl ); acn_obj_discard( cne ); acn_obj_discard( hyp );
    }
    return scene;
}

/* ================================================================================================================== */
/* src_acn/wine_glass.acn */
acn_scene* acn_scene_wine_glass( void )
{
    acn_scene* scene = acn_scene_s_create();
    scene->threads = 64;
    scene->prm.image_width = 400;
    scene->prm.image_height = 400;
    scene->prm.gamma = 0.9;
    scene->gradient_cycles = 100;
    scene->gradient_samples = 2;
    scene->gradient_threshold = 0.03;
    scene->prm.trace_depth = 25;
    scene->prm.trace_min_intensity = 0.03;
    scene->prm.direct_samples = 200;
    scene->prm.path_samples = 500;
    double cam[ 3 ] = { 0, -10, 7 };
    for( int k = 0; k < 3; k++ ) { scene->prm.camera_position[ k ] = cam[ k ]; scene->prm.camera_view_direction[ k ] = 0 + cam[ k ] * -1; }
    scene->prm.camera_top_direction[ 2 ] = 1;
    scene->prm.camera_focal_length = 4;
    scene->prm.background_color[ 0 ] = 0.4; scene->prm.background_color[ 1 ] = 0.4; scene->prm.background_color[ 2 ] = 0.3;

    /* create_lights() */
    {
        acn_obj* base_light = acn_obj_sphere_s_create( 1 );
        acn_obj_set_radiance( base_light, 40 );
        acn_obj_set_color( base_light, vec( 1.0, 1.0, 1.0 ) );
        acn_obj_move( base_light, vec( -2, 2, 5 ) );
        scene_push_take( scene, base_light );
    }
    /* create_floor() */
    {
        acn_obj* plane = acn_obj_plane_s_create();
        acn_obj_set_material( plane, "diffuse" );
        acn_obj_set_color( plane, vec( 0.8, 0.7, 0.6 ) );
        acn_obj_set_refractive_index( plane, 1.0 );
        acn_obj_move( plane, vec( 0, 0, -1 ) );
        scene_push_take( scene, plane );
    }
    /* create_glass() */
    {
        acn_obj* sphere   = acn_obj_sphere_s_create( 1 );
        acn_obj* cover    = acn_obj_plane_s_create();
        acn_obj* cylinder = acn_obj_squaroid_s_create_cylinder( 1, 1 );

        acn_obj* outer_sphere = C( sphere );
        acn_obj* bowl = AND( C( outer_sphere ), add_v( cover, vecz( 0.6 ) ) );
        acn_obj* inner_sphere = mul_f( outer_sphere, 0.96 );
        acn_obj* liquid_sphere = C( inner_sphere );
        acn_obj* liquid = AND( C( liquid_sphere ), sub_v( cover, vecz( 0.3 ) ) );

        acn_obj_move( inner_sphere, vecz( 0.02 ) );
        acn_obj_move( liquid, vecz( 0.02 ) );

        bowl = AND( bowl, NOT( C( inner_sphere ) ) );

        set_envelope_sphere( bowl, mul_f( outer_sphere, 1.01 ) );
        set_envelope_sphere( liquid, mul_f( outer_sphere, 1.01 ) );

        acn_obj_move( bowl, vecz( 2 ) );
        acn_obj_move( liquid, vecz( 2 ) );

        acn_obj* neckcyl = AND( mul_f( cylinder, 0.08 ), AND( add_v( cover, vecz( 0.5 ) ), NOT( sub_v( cover, vecz( 0.5 ) ) ) ) );
        acn_obj* pearl = mul_f( sphere, 0.15 );
        acn_obj* neck = OR( neckcyl, OR( add_v( pearl, vecz( 0.45 ) ), sub_v( pearl, vecz( 0.45 ) ) ) );
        set_envelope_sphere( neck, mul_f( sphere, 1.2 ) );
        acn_obj_move( neck, vecz( 0.55 ) );

        acn_obj* bottom = AND( mul_f( sphere, 3 ), AND( mul_f( cylinder, 0.8 ), NOT( add_v( cover, vecz( 2.85 ) ) ) ) );
        {
            acn_obj* t = mul_f( sphere, 0.85 );
            set_envelope_sphere( bottom, add_v( t, vecz( 2.85 ) ) );
            acn_obj_discard( t );
        }
        acn_obj_move( bottom, vecz( -2.85 ) );

        acn_obj* glass = OR( bowl, OR( neck, bottom ) );
        {
            acn_obj* t = mul_f( sphere, 1.7 );
            set_envelope_sphere( glass, add_v( t, vecz( 1.5 ) ) );
            acn_obj_discard( t );
        }
        acn_obj_set_material( glass, "glass" );
        acn_obj_set_material( liquid, "water" );
        acn_obj_set_transparency( liquid, vec( 0.177, 9.61E-6, 9.54E-7 ) );

        /* def wine = glass : liquid; wine.move( vecz( -0.999 ) ); */
        acn_obj_move( glass, vecz( -0.999 ) );
        acn_obj_move( liquid, vecz( -0.999 ) );
        acn_scene_s_push( scene, glass );
        acn_scene_s_push( scene, liquid );

        acn_obj_discard( glass ); acn_obj_discard( liquid ); acn_obj_discard( pearl );
        acn_obj_discard( inner_sphere ); acn_obj_discard( liquid_sphere ); acn_obj_discard( outer_sphere );
        acn_obj_discard( sphere ); acn_obj_discard( cover ); acn_obj_discard( cylinder );
    }
    return scene;
}

/* ================================================================================================================== */
/* src_acn/diamond.acn */
static void append_to_cuts( list_t* cuts, acn_obj* cut_side_owned, int num_cuts )
{
    for( int i = 0; i < num_cuts; i++ )
    {
        /* i * 360 / num_cuts: (i*360) integer, then * inverse(num_cuts) = * (1.0/num_cuts)  interpreter.c:1697,987 */
        double deg = ( double )( i * 360 ) * ( 1.0 / num_cuts );
        list_push( cuts, mul_m( cut_side_owned, acn_rotz( deg ) ) );
    }
    acn_obj_discard( cut_side_owned );
}

static acn_obj* sub_v_take( acn_obj* o, acn_v3 v ) { acn_obj_move( o, neg( v ) ); return o; }
static acn_obj* add_v_take( acn_obj* o, acn_v3 v ) { acn_obj_move( o, v ); return o; }
static acn_obj* mul_m_take( acn_obj* o, acn_m3 m ) { acn_obj_rotate( o, &m ); return o; }

acn_scene* acn_scene_diamond( void )
{
    acn_scene* scene = acn_scene_s_create();
    scene->threads = 10;
    scene->prm.image_width = 400;
    scene->prm.image_height = 400;
    scene->prm.gamma = 1.0;
    scene->gradient_cycles = 100;
    scene->gradient_samples = 2;
    scene->gradient_threshold = 0.03;
    scene->prm.trace_depth = 25;
    scene->prm.trace_min_intensity = 0.03;
    scene->prm.direct_samples = 50;
    scene->prm.path_samples = 50;
    scene->prm.max_path_length = 0.1;
    double cam[ 3 ] = { 0, -0.5, 0.1 };
    double tgt[ 3 ] = { 0, 0, -0.05 };
    for( int k = 0; k < 3; k++ ) { scene->prm.camera_position[ k ] = cam[ k ]; scene->prm.camera_view_direction[ k ] = tgt[ k ] + cam[ k ] * -1; }
    scene->prm.camera_top_direction[ 2 ] = 1;
    scene->prm.camera_focal_length = 4;
    scene->prm.background_color[ 0 ] = 0.6; scene->prm.background_color[ 1 ] = 0.7; scene->prm.background_color[ 2 ] = 0.8;

    /* create_light( 0.025, 0.1 ) + vec( 0.04, 0.04, 0.125 ) */
    {
        acn_obj* sph = acn_obj_sphere_s_create( 1 );
        acn_obj* light = mul_f( sph, 0.025 );
        acn_obj_set_radiance( light, 0.1 );
        scene_push_take( scene, add_v( light, vec( 0.04, 0.04, 0.125 ) ) );
        acn_obj_discard( light ); acn_obj_discard( sph );
    }
    double floor_offset = -0.075;
    /* create_floor( floor_offset ) */
    {
        acn_obj* plane = acn_obj_plane_s_create();
        acn_obj_set_material( plane, "diffuse" );
        acn_obj_set_color( plane, vec( 0.6, 0.4, 0.2 ) );
        acn_obj_set_refractive_index( plane, 1.2 );
        acn_obj_move( plane, vec( 0, 0, floor_offset ) );
        scene_push_take( scene, plane );
    }
    double plate_height = 0.01;
    list_t all = { { 0 }, 0 };   /* diamond_on_plate, flattened in push order */

    /* create_plate( floor_offset, 0.1, plate_height ) */
    {
        double floor_offs = floor_offset, radius = 0.1, height = plate_height;
        double cloth_height = 0.00075;
        acn_obj* cover = acn_obj_plane_s_create();
        acn_obj* cone = acn_obj_squaroid_s_create_cone( 1, 1, 1 );
        acn_obj* plate = AND( C( cone ), AND( sub_v( cover, vecz( radius + cloth_height ) ), NOT( sub_v( cover, vecz( radius + height ) ) ) ) );
        acn_obj_move( plate, vecz( floor_offs + ( radius + height ) ) );
        double chw = radius * 0.65;
        list_t lst = { { 0 }, 0 };
        list_push( &lst, C( cover ) );
        list_push( &lst, sub_v_take( mul_m( cover, acn_rotx(  90 ) ), vecy( chw ) ) );
        list_push( &lst, add_v_take( mul_m( cover, acn_rotx( -90 ) ), vecy( chw ) ) );
        list_push( &lst, add_v_take( mul_m( cover, acn_roty(  90 ) ), vecx( chw ) ) );
        list_push( &lst, sub_v_take( mul_m( cover, acn_roty( -90 ) ), vecx( chw ) ) );
        acn_obj* cloth = acn_create_inside_composite( lst.d, lst.n );
        list_clear( &lst );
        acn_obj_move( cloth, vecz( floor_offs + height ) );
        { acn_m3 m = acn_rotz( 45 ); acn_obj_rotate( cloth, &m ); }
        acn_obj_set_material( cloth, "diffuse" );
        acn_obj_set_color( cloth, vec( 0.9, 0.1, 0.2 ) );
        acn_obj_set_material( plate, "diffuse_polished" );
        acn_obj_set_color( plate, vec( 0.2, 0.1, 0.05 ) );
        list_push( &all, plate );
        list_push( &all, cloth );
        acn_obj_discard( cover ); acn_obj_discard( cone );
    }
    /* create_diamond_on_stand( floor_offset + plate_height - 0.004 ) */
    {
        double bottom_zoffs = floor_offset + ( plate_height + 0.004 * -1 );
        /* def diamond = create_diamond( 0.0472 ) */
        acn_obj* diamond;
        {
            double radius = 0.0472;
            acn_obj* plane = acn_obj_plane_s_create();
            acn_obj_set_color( plane, vec( 1, 0.5, 0.3 ) );
            acn_obj* sphere = acn_obj_sphere_s_create( 1 );
            acn_obj* plate = add_v( plane, vecz( 0.24 ) );
            acn_obj* side = mul_m( plane, acn_rotx( 90 ) );
            list_t cuts = { { 0 }, 0 };
            append_to_cuts( &cuts, sub_v_take( mul_m( side, acn_rotx( 37 - 90 ) ), vecy( 1.00 ) ), 16 );
            append_to_cuts( &cuts, mul_m_take( sub_v_take( mul_m( side, acn_rotx( 32 - 90 ) ), vecy( 1.02 ) ), acn_rotz(  180 * ( 1.0 / 16 ) ) ), 8 );
            append_to_cuts( &cuts, mul_m_take( sub_v_take( mul_m( side, acn_rotx( 26 - 90 ) ), vecy( 1.12 ) ), acn_rotz( -180 * ( 1.0 / 16 ) ) ), 8 );
            append_to_cuts( &cuts, sub_v_take( mul_m( side, acn_rotx( 90 - 42 ) ), vecy( 1.00 ) ), 16 );
            append_to_cuts( &cuts, mul_m_take( sub_v_take( mul_m( side, acn_rotx( 90 - 40 ) ), vecy( 1.02 ) ), acn_rotz(  180 * ( 1.0 / 16 ) ) ), 8 );
            acn_obj* gem = AND( acn_create_inside_composite( cuts.d, cuts.n ), AND( C( plate ), C( sphere ) ) );
            list_clear( &cuts );
            acn_obj_set_material( gem, "diamond" );
            set_envelope_sphere( gem, mul_f( sphere, 1.01 ) );
            acn_obj_scale( gem, radius );
            diamond = gem;
            acn_obj_discard( plane ); acn_obj_discard( sphere ); acn_obj_discard( plate ); acn_obj_discard( side );
        }
        /* def stand = create_stand( bottom_zoffs, 0.05 ) */
        list_t stand = { { 0 }, 0 };
        {
            double radius = 0.05;
            acn_obj* sph = acn_obj_sphere_s_create( 1 );
            acn_obj* cov = acn_obj_plane_s_create();
            list_t lst = { { 0 }, 0 };
            list_push( &lst, mul_f( sph, radius ) );
            list_push( &lst, NOT( sub_v( cov, vecz( radius * 0.1 ) ) ) );
            list_push( &lst, NOT( add_v_take( mul_f( sph, radius ), vecz( radius * 0.4 ) ) ) );
            list_push( &lst, add_v( cov, vecz( radius * 0.08 ) ) );
            acn_obj* ring = acn_create_inside_composite( lst.d, lst.n );
            acn_obj_set_color( ring, vec( 0.83, 0.68, 0.22 ) );
            acn_obj_set_chromatic_reflectivity( ring, 0.7 );
            list_clear( &lst );

            list_push( &lst, mul_f( sph, radius ) );
            list_push( &lst, NOT( add_v_take( mul_f( sph, radius ), vec( -radius * 0.05, -radius * 0.2, 0 ) ) ) );
            list_push( &lst, NOT( add_v_take( mul_f( sph, radius ), vec( -radius * 0.05,  radius * 0.2, 0 ) ) ) );
            list_push( &lst, mul_m_take( mul_m( cov, acn_rotx(  90 ) ), acn_rotz( -7 ) ) );
            list_push( &lst, mul_m_take( mul_m( cov, acn_rotx( -90 ) ), acn_rotz(  7 ) ) );
            list_push( &lst, add_v_take( mul_m( cov, acn_roty( -90 ) ), vecx( radius * 0.5 ) ) );
            list_push( &lst, NOT( C( cov ) ) );
            acn_obj* bar = acn_create_inside_composite( lst.d, lst.n );
            list_clear( &lst );
            {
                acn_obj* t = mul_f( sph, radius );
                acn_obj_scale( t, 0.8 );
                acn_obj_move( t, vecx( radius ) );
                set_envelope_sphere( bar, mul_m_take( t, acn_roty( -45 ) ) );
            }
            acn_obj_move( bar, vec( radius * 0.45, 0, bottom_zoffs ) );
            acn_obj_set_color( bar, vec( 0.83, 0.68, 0.22 ) );
            acn_obj_set_chromatic_reflectivity( bar, 0.7 );

            acn_obj_move( ring, vec( 0, 0, bottom_zoffs + radius * 0.85 ) );

            list_push( &stand, ring );
            list_push( &stand, C( bar ) );
            list_rotate( &stand, acn_rotz( 90 ) );
            list_push( &stand, C( bar ) );
            list_rotate( &stand, acn_rotz( 90 ) );
            list_push( &stand, C( bar ) );
            list_rotate( &stand, acn_rotz( 90 ) );
            list_push( &stand, C( bar ) );
            acn_obj_discard( bar ); acn_obj_discard( sph ); acn_obj_discard( cov );
        }
        acn_obj_move( diamond, vecz( 0.046 + bottom_zoffs ) );
        list_push( &all, diamond );
        for( size_t i = 0; i < stand.n; i++ ) list_push( &all, stand.d[ i ] );
    }
    list_rotate( &all, acn_rotz( 30 ) );
    for( size_t i = 0; i < all.n; i++ ) acn_scene_s_push( scene, all.d[ i ] );
    list_clear( &all );
    return scene;
}

/* ================================================================================================================== */
/* src_acn/many_spheres.acn; levels = 5 in the script. exact_envelopes != 0 replaces the Monte-Carlo
 * set_auto_envelope() of the 8 seed spheres (GPU) by the sphere itself scaled 1.1 (CPU-only builds). */
acn_scene* acn_scene_many_spheres( int levels, int exact_envelopes )
{
    acn_scene* scene = acn_scene_s_create();
    scene->threads = 10;
    scene->prm.image_width = 600;
    scene->prm.image_height = 600;
    scene->prm.gamma = 0.7;
    scene->gradient_cycles = 20;
    scene->gradient_samples = 2;
    scene->gradient_threshold = 0.03;
    scene->prm.trace_depth = 25;
    scene->prm.trace_min_intensity = 0.05;
    scene->prm.direct_samples = 20;
    scene->prm.path_samples = 20;
    scene->prm.max_path_length = 1;
    double cam[ 3 ] = { 0, -10, 5 };
    for( int k = 0; k < 3; k++ ) { scene->prm.camera_position[ k ] = cam[ k ]; scene->prm.camera_view_direction[ k ] = 0 + cam[ k ] * -1; }
    scene->prm.camera_top_direction[ 2 ] = 1;
    scene->prm.camera_focal_length = 4;
    scene->prm.background_color[ 0 ] = 0.4; scene->prm.background_color[ 1 ] = 0.5; scene->prm.background_color[ 2 ] = 0.6;

    {
        acn_obj* sph = acn_obj_sphere_s_create( 1.0 );
        acn_obj* light = mul_f( sph, 0.5 );
        acn_obj_set_radiance( light, 30 );
        scene_push_take( scene, add_v( light, vec( -4, -4, 4 ) ) );
        acn_obj_discard( light ); acn_obj_discard( sph );
    }
    {
        acn_obj* plane = acn_obj_plane_s_create();
        acn_obj_set_material( plane, "diffuse_polished" );
        acn_obj_set_color( plane, vec( 0.6, 0.6, 0.5 ) );
        acn_obj_move( plane, vec( 0, 0, -2 ) );
        scene_push_take( scene, plane );
    }
    /* create_matter( 0.025, 2.4, 5 ) */
    {
        double d = 0.025, f = 2.4;
        acn_obj* sph = acn_obj_sphere_s_create( d );
        acn_obj_set_material( sph, "diffuse_polished" );
        acn_obj_set_color( sph, vec( 0.9, 0.8, 0.6 ) );
        acn_obj* cmp = acn_compound_s_create();
        acn_compound_s_push( cmp, sph );
        acn_obj_discard( sph );
        for( int i = 0; i < levels; i++ )
        {
            acn_obj_move( cmp, neg( vec( d, d, d ) ) );
            for( int axis = 0; axis < 3; axis++ )
            {
                acn_obj* shifted = add_v( cmp, axis == 0 ? vecx( d * 2 ) : axis == 1 ? vecy( d * 2 ) : vecz( d * 2 ) );
                acn_obj* n = acn_compound_s_create();
                acn_compound_s_push( n, cmp );
                acn_compound_s_push( n, shifted );
                acn_obj_discard( cmp ); acn_obj_discard( shifted );
                cmp = n;
            }
            if( exact_envelopes ) acn_compound_s_set_sphere_envelopes( cmp, 1.1 );
            if( acn_obj_set_auto_envelope( cmp ) != ACN_OK ) { acn_obj_discard( cmp ); acn_scene_s_discard( scene ); return NULL; }
            d *= f;
        }
        { acn_m3 m = acn_rotz( 20 ); acn_obj_rotate( cmp, &m ); }
        scene_push_take( scene, cmp );
    }
    return scene;
}
