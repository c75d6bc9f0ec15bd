/* acn_interp.c -- interpreter for Actinon scene scripts (see include/acn_interp.h).
 *
 * The language has no written grammar; it is what the reference's evaluator does.  The rules this file follows
 * (paths relative to /root/reference):
 *   tokens, literals, `#parse`, `#source_file_name`, nested `{}` blocks        src/interpreter.c:207-511
 *   expression evaluation with a "front object" (operator binding, postfix ops)  src/interpreter.c:1412-1730
 *   statements, if / else / while / for                                          src/interpreter.c:1734-1850
 *   closures: signature * block, lexical frame, one local frame per block        src/interpreter.c:1880-1923
 *   arithmetic tables ( * + / - % comparison, logic, catenation )                src/interpreter.c:651-1231
 *   root frame: built-in functions and constants                                 src/interpreter.c:1943-2015
 *   members of scene / map / list / compound / object                            src/scene.c:293-331,
 *        src/container.c:156-231,423-518, src/compound.c:380-455, src/objects.c:1463-1725
 *   built-in functions                                                            src/closures.c:25-604
 * Value model: every script value is a reference-counted box; `def`, list.push, map member creation and
 * compound / scene push deep-copy, function arguments and `for` variables alias (as the reference's sr_s do).
 * Known deviations (documented in DESIGN.md): maps iterate in insertion order (beth: hash order), beth_object() makes the
 * texture maps / distance functions / obj_distance_s that set_texture_field / set_distance_function take and nothing else, beth's
 * generic object printing (`?`), read_from_file / write_to_file and string_fa() beyond integer padding are not provided.
 */
#include <ctype.h>
#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "acn_interp.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------------------------- */
/* data model                                                                                                    */

typedef struct val val;
typedef struct frame frame;
typedef struct block block;
typedef struct interp interp;

enum vtype
{
    V_BOOL = 1, V_INT, V_FLOAT, V_STR, V_VEC, V_COLOR, V_MAT, V_LIST, V_MAP, V_OBJ, V_SCENE,
    V_CLOSURE, V_SIG, V_BUILTIN, V_TYPE,
    V_BETH   /* a texture map or a distance function made by beth_object(): see beth_kinds_g */
};

/* argument types of a signature (interpreter.c:224-233) */
enum sigtype { S_ANY = 0, S_BOOL, S_INT, S_FLOAT, S_NUM, S_STRING, S_MAP, S_LIST, S_OBJECT, S_V3D, S_FUNC };

typedef struct sigarg { const char* name; int type; } sigarg;

struct val
{
    int type;
    int rc;
    union
    {
        int b;
        int64_t i;
        double f;
        acn_v3 v;
        acn_m3 m;
        char* s;
        struct { val** d; size_t n, cap; } list;                 /* entries may be NULL (unset index) */
        struct { const char** k; val** d; size_t n, cap; } map;  /* insertion order */
        acn_obj* obj;                                            /* any object or a compound */
        acn_scene* scene;
        struct { block* blk; val* sig; frame* lex; } clo;
        struct { sigarg* a; size_t n; } sig;
        int builtin;
        int mtype;
        struct { int kind; acn_v3 c1, c2; double scale, ex_radius; } beth;
    } u;
};

struct frame
{
    const char** k;
    val** v;
    size_t n, cap;
    frame* ext;
};

enum tk
{
    TK_END = 0, TK_DATA, TK_NAME, TK_COMMA, TK_SEMI, TK_LPAR, TK_RPAR, TK_LBRK, TK_RBRK, TK_DEF, TK_FSIG, TK_DYNARR,
    TK_OP_BEGIN,
    TK_DOT, TK_QUERY, TK_DQUERY, TK_MUL, TK_DIV, TK_MOD, TK_ADD, TK_SUB,
    TK_ASG_BEGIN, TK_ASSIGN, TK_MUL_ASG, TK_ADD_ASG, TK_SUB_ASG, TK_DIV_ASG, TK_MOD_ASG, TK_ASG_END,
    TK_EQ, TK_LT, TK_NE, TK_LE, TK_GT, TK_GE,
    TK_NOT, TK_AND, TK_OR, TK_XOR, TK_CAT, TK_ICPS, TK_OCPS, TK_CMPD, TK_ENV,
    TK_OP_END,
    TK_IF, TK_WHILE, TK_ELSE, TK_FOR, TK_IN
};

typedef struct token
{
    int kind;
    val* lit;            /* TK_DATA literal */
    block* blk;          /* TK_DATA block */
    const char* name;    /* TK_NAME (interned) */
    int file, line;
} token;

struct block
{
    token* t;
    size_t n, cap;
    frame local;         /* one local frame per block, cleared on every call (interpreter.c:1900-1907) */
};

struct interp
{
    jmp_buf jb;
    char err[ 1024 ];
    const acn_interp_opts* opts;
    char** names;  size_t n_names, cap_names;        /* interned identifiers */
    char** files;  size_t n_files, cap_files;
    clock_t start;
    int depth;
};

typedef struct ev { block* blk; size_t ix; frame* fr; interp* ip; } ev;

static char last_error_g[ 1024 ];
const char* acn_interp_last_error( void ) { return last_error_g; }

/* ------------------------------------------------------------------------------------------------------------- */
/* errors, memory                                                                                                */

static void fail_at( interp* ip, int file, int line, const char* fmt, ... )
{
    va_list a;
    int n = 0;
    if( file >= 0 && ( size_t )file < ip->n_files ) n = snprintf( ip->err, sizeof( ip->err ), "%s:%d: ", ip->files[ file ], line );
    va_start( a, fmt );
    vsnprintf( ip->err + n, sizeof( ip->err ) - n, fmt, a );
    va_end( a );
    longjmp( ip->jb, 1 );
}

static const token* cur_tok( const ev* e )
{
    static const token end_tok = { TK_END, NULL, NULL, NULL, -1, 0 };
    if( e->ix < e->blk->n ) return &e->blk->t[ e->ix ];
    if( e->blk->n ) return &e->blk->t[ e->blk->n - 1 ];
    return &end_tok;
}

#define FAIL( e, ... ) fail_at( ( e )->ip, cur_tok( e )->file, cur_tok( e )->line, __VA_ARGS__ )

static void* xalloc( size_t n ) { void* p = calloc( 1, n ? n : 1 ); if( !p ) { fprintf( stderr, "acn_interp: out of memory\n" ); abort(); } return p; }
static void* xrealloc( void* p, size_t n ) { p = realloc( p, n ? n : 1 ); if( !p ) { fprintf( stderr, "acn_interp: out of memory\n" ); abort(); } return p; }
static char* xstrdup( const char* s ) { size_t n = strlen( s ) + 1; char* r = xalloc( n ); memcpy( r, s, n ); return r; }

static const char* intern( interp* ip, const char* s )
{
    for( size_t i = 0; i < ip->n_names; i++ ) if( strcmp( ip->names[ i ], s ) == 0 ) return ip->names[ i ];
    if( ip->n_names == ip->cap_names ) { ip->cap_names = ip->cap_names ? ip->cap_names * 2 : 256; ip->names = xrealloc( ip->names, ip->cap_names * sizeof( char* ) ); }
    return ip->names[ ip->n_names++ ] = xstrdup( s );
}

static const char* type_name( int t )
{
    switch( t )
    {
        case V_BOOL: return "bool";   case V_INT: return "int";     case V_FLOAT: return "float"; case V_STR: return "string";
        case V_VEC: return "v3d";     case V_COLOR: return "color"; case V_MAT: return "m3d";     case V_LIST: return "list";
        case V_MAP: return "map";     case V_OBJ: return "object";  case V_SCENE: return "scene"; case V_CLOSURE: return "func";
        case V_SIG: return "signature"; case V_BUILTIN: return "func"; case V_TYPE: return "type";
        case V_BETH: return "beth object";
    }
    return "null";
}
/* What beth_object( name ) can make: the types the object members set_texture_field / set_distance_function take
 * (textures.c:82-88,130-138; distance.c:30-35,68-74) -- beth's registry of every reflected type is not available. */
enum { BETH_TXM_PLAIN = 0, BETH_TXM_CHESS, BETH_DISTANCE_SPHERE, BETH_DISTANCE_TORUS, BETH_KINDS };
static const char* const beth_kinds_g[ BETH_KINDS ] = { "txm_plain_s", "txm_chess_s", "distance_sphere_s", "distance_torus_s" };
static const char* vt( const val* v ) { return !v ? "null" : v->type == V_BETH ? beth_kinds_g[ v->u.beth.kind ] : type_name( v->type ); }

/* ------------------------------------------------------------------------------------------------------------- */
/* values                                                                                                        */

static val* v_new( int type ) { val* v = xalloc( sizeof( val ) ); v->type = type; v->rc = 1; return v; }
static val* v_ref( val* v ) { if( v ) v->rc++; return v; }
static void v_unref( val* v );

static void v_release_payload( val* v )
{
    switch( v->type )
    {
        case V_STR: free( v->u.s ); break;
        case V_LIST:
            for( size_t i = 0; i < v->u.list.n; i++ ) v_unref( v->u.list.d[ i ] );
            free( v->u.list.d );
            break;
        case V_MAP:
            for( size_t i = 0; i < v->u.map.n; i++ ) v_unref( v->u.map.d[ i ] );
            free( v->u.map.d ); free( ( void* )v->u.map.k );
            break;
        case V_OBJ: acn_obj_discard( v->u.obj ); break;
        case V_SCENE: acn_scene_s_discard( v->u.scene ); break;
        case V_CLOSURE: v_unref( v->u.clo.sig ); break;
        case V_SIG: free( v->u.sig.a ); break;
        default: break;
    }
    memset( &v->u, 0, sizeof( v->u ) );
}

static void v_unref( val* v )
{
    if( !v ) return;
    if( --v->rc > 0 ) return;
    v_release_payload( v );
    free( v );
}

static val* v_bool( int b ) { val* v = v_new( V_BOOL ); v->u.b = b ? 1 : 0; return v; }
static val* v_int( int64_t i ) { val* v = v_new( V_INT ); v->u.i = i; return v; }
static val* v_float( double f ) { val* v = v_new( V_FLOAT ); v->u.f = f; return v; }
static val* v_str( const char* s ) { val* v = v_new( V_STR ); v->u.s = xstrdup( s ); return v; }
static val* v_vec( acn_v3 a ) { val* v = v_new( V_VEC ); v->u.v = a; return v; }
static val* v_color( acn_v3 a ) { val* v = v_new( V_COLOR ); v->u.v = a; return v; }
static val* v_mat( acn_m3 m ) { val* v = v_new( V_MAT ); v->u.m = m; return v; }
static val* v_obj( acn_obj* o ) { val* v = v_new( V_OBJ ); v->u.obj = o; return v; }   /* takes ownership */

static int is_compound( const val* v ) { return v && v->type == V_OBJ && acn_obj_type( v->u.obj ) == ACN_COMPOUND; }
static int is_object( const val* v )   { return v && v->type == V_OBJ && acn_obj_type( v->u.obj ) != ACN_COMPOUND; }

acn_scene* acn_scene_s_clone( const acn_scene* o )
{
    acn_scene* s = acn_scene_s_create();
    acn_obj* l = s->light; acn_obj* m = s->matter;
    *s = *o;
    acn_obj_discard( l ); acn_obj_discard( m );
    s->light = acn_obj_clone( o->light );
    s->matter = acn_obj_clone( o->matter );
    return s;
}

static void list_push_owned( val* l, val* e )
{
    if( l->u.list.n == l->u.list.cap ) { l->u.list.cap = l->u.list.cap ? l->u.list.cap * 2 : 8; l->u.list.d = xrealloc( l->u.list.d, l->u.list.cap * sizeof( val* ) ); }
    l->u.list.d[ l->u.list.n++ ] = e;
}

static val** map_slot( val* m, const char* key )
{
    for( size_t i = 0; i < m->u.map.n; i++ ) if( m->u.map.k[ i ] == key ) return &m->u.map.d[ i ];
    return NULL;
}

static void map_set_owned( val* m, const char* key, val* e )
{
    val** s = map_slot( m, key );
    if( s ) { v_unref( *s ); *s = e; return; }
    if( m->u.map.n == m->u.map.cap )
    {
        m->u.map.cap = m->u.map.cap ? m->u.map.cap * 2 : 8;
        m->u.map.d = xrealloc( m->u.map.d, m->u.map.cap * sizeof( val* ) );
        m->u.map.k = xrealloc( ( void* )m->u.map.k, m->u.map.cap * sizeof( char* ) );
    }
    m->u.map.k[ m->u.map.n ] = key;
    m->u.map.d[ m->u.map.n++ ] = e;
}

/* deep copy (sr_clone) */
static val* v_clone( const val* s )
{
    if( !s ) return NULL;
    val* v = v_new( s->type );
    switch( s->type )
    {
        case V_STR: v->u.s = xstrdup( s->u.s ); break;
        case V_LIST: for( size_t i = 0; i < s->u.list.n; i++ ) list_push_owned( v, v_clone( s->u.list.d[ i ] ) ); break;
        case V_MAP:  for( size_t i = 0; i < s->u.map.n; i++ ) map_set_owned( v, s->u.map.k[ i ], v_clone( s->u.map.d[ i ] ) ); break;
        case V_OBJ: v->u.obj = acn_obj_clone( s->u.obj ); break;
        case V_SCENE: v->u.scene = acn_scene_s_clone( s->u.scene ); break;
        case V_CLOSURE: v->u.clo = s->u.clo; v_ref( v->u.clo.sig ); break;    /* keeps the lexical frame (interpreter.c:1871-1876) */
        case V_SIG:
            v->u.sig.n = s->u.sig.n;
            v->u.sig.a = xalloc( sizeof( sigarg ) * s->u.sig.n );
            memcpy( v->u.sig.a, s->u.sig.a, sizeof( sigarg ) * s->u.sig.n );
            break;
        default: v->u = s->u; break;
    }
    return v;
}

/* typed copy into existing storage: `x = expr` on a defined x (interpreter.c:1477).  Leaf numbers convert to
 * the destination's type, everything else takes the source's type and content. */
static void v_assign( val* dst, const val* src )
{
    if( dst == src ) return;
    if( dst->type == V_INT && src->type == V_FLOAT ) { dst->u.i = ( int64_t )src->u.f; return; }
    if( dst->type == V_INT && src->type == V_BOOL ) { dst->u.i = src->u.b; return; }
    if( dst->type == V_FLOAT && src->type == V_INT ) { dst->u.f = ( double )src->u.i; return; }
    if( dst->type == V_FLOAT && src->type == V_BOOL ) { dst->u.f = src->u.b; return; }
    if( dst->type == V_BOOL && src->type == V_INT ) { dst->u.b = src->u.i != 0; return; }
    if( ( dst->type == V_VEC || dst->type == V_COLOR ) && ( src->type == V_VEC || src->type == V_COLOR ) ) { dst->u.v = src->u.v; return; }
    val* c = v_clone( src );
    v_release_payload( dst );
    dst->type = c->type;
    dst->u = c->u;
    free( c );
}

static int is_num( const val* v ) { return v && ( v->type == V_INT || v->type == V_FLOAT || v->type == V_BOOL ); }
static double to_f3( const val* v ) { return v->type == V_INT ? ( double )v->u.i : v->type == V_FLOAT ? v->u.f : ( double )v->u.b; }

/* ------------------------------------------------------------------------------------------------------------- */
/* frames (bclos_frame_s)                                                                                        */

static val** frame_get_local( frame* f, const char* key )
{
    for( size_t i = 0; i < f->n; i++ ) if( f->k[ i ] == key ) return &f->v[ i ];
    return NULL;
}

static val** frame_get( frame* f, const char* key )
{
    for( ; f; f = f->ext ) { val** s = frame_get_local( f, key ); if( s ) return s; }
    return NULL;
}

static val** frame_set( frame* f, const char* key, val* owned )
{
    val** s = frame_get_local( f, key );
    if( s ) { v_unref( *s ); *s = owned; return s; }
    if( f->n == f->cap )
    {
        f->cap = f->cap ? f->cap * 2 : 16;
        f->k = xrealloc( ( void* )f->k, f->cap * sizeof( char* ) );
        f->v = xrealloc( f->v, f->cap * sizeof( val* ) );
    }
    f->k[ f->n ] = key;
    f->v[ f->n ] = owned;
    return &f->v[ f->n++ ];
}

static void frame_clear( frame* f )
{
    for( size_t i = 0; i < f->n; i++ ) v_unref( f->v[ i ] );
    f->n = 0;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* tokenizer (mcode_s_parse)                                                                                     */

typedef struct lexer { interp* ip; const char* s; size_t p, n; int file, line; } lexer;

static void lex_fail( lexer* lx, const char* msg ) { fail_at( lx->ip, lx->file, lx->line, "%s", msg ); }

static void skip_space( lexer* lx )
{
    for( ;; )
    {
        while( lx->p < lx->n && isspace( ( unsigned char )lx->s[ lx->p ] ) ) { if( lx->s[ lx->p ] == '\n' ) lx->line++; lx->p++; }
        if( lx->p + 1 < lx->n && lx->s[ lx->p ] == '/' && lx->s[ lx->p + 1 ] == '/' )
        {
            while( lx->p < lx->n && lx->s[ lx->p ] != '\n' ) lx->p++;
        }
        else if( lx->p + 1 < lx->n && lx->s[ lx->p ] == '/' && lx->s[ lx->p + 1 ] == '*' )
        {
            lx->p += 2;
            while( lx->p + 1 < lx->n && !( lx->s[ lx->p ] == '*' && lx->s[ lx->p + 1 ] == '/' ) ) { if( lx->s[ lx->p ] == '\n' ) lx->line++; lx->p++; }
            if( lx->p + 1 >= lx->n ) lex_fail( lx, "Stream ends in comment" );
            lx->p += 2;
        }
        else break;
    }
}

static token* push_tok( block* b, lexer* lx, int kind )
{
    if( b->n == b->cap ) { b->cap = b->cap ? b->cap * 2 : 64; b->t = xrealloc( b->t, b->cap * sizeof( token ) ); }
    token* t = &b->t[ b->n++ ];
    memset( t, 0, sizeof( *t ) );
    t->kind = kind; t->file = lx->file; t->line = lx->line;
    return t;
}

static int match( lexer* lx, const char* s )
{
    size_t l = strlen( s );
    if( lx->p + l <= lx->n && memcmp( lx->s + lx->p, s, l ) == 0 ) { lx->p += l; return 1; }
    return 0;
}

static char* read_file( const char* path, size_t* size )
{
    FILE* f = fopen( path, "rb" );
    if( !f ) return NULL;
    fseek( f, 0, SEEK_END ); long n = ftell( f ); fseek( f, 0, SEEK_SET );
    char* s = xalloc( ( size_t )n + 1 );
    if( fread( s, 1, ( size_t )n, f ) != ( size_t )n ) { fclose( f ); free( s ); return NULL; }
    fclose( f );
    s[ n ] = 0;
    *size = ( size_t )n;
    return s;
}

static int add_file( interp* ip, const char* name )
{
    if( ip->n_files == ip->cap_files ) { ip->cap_files = ip->cap_files ? ip->cap_files * 2 : 8; ip->files = xrealloc( ip->files, ip->cap_files * sizeof( char* ) ); }
    ip->files[ ip->n_files ] = xstrdup( name );
    return ( int )ip->n_files++;
}

static const struct { const char* name; int type; } sig_types_g[] =
{
    { "bool", S_BOOL }, { "int", S_INT }, { "float", S_FLOAT }, { "num", S_NUM }, { "string", S_STRING }, { "map", S_MAP },
    { "list", S_LIST }, { "object", S_OBJECT }, { "v3d", S_V3D }, { "func", S_FUNC }, { NULL, 0 }
};

static void parse_block( lexer* lx, block* b );

/* literal: integer part, fraction accumulated digit by digit with f *= 0.1, then * pow( 10, exponent )
 * (interpreter.c:247-281) -- NOT a correctly rounded decimal conversion, and deliberately reproduced. */
static void parse_number( lexer* lx, block* b )
{
    uint64_t vi = 0; double vf = 0; int64_t vx = 0; int is_int = 1;
    while( lx->p < lx->n && isdigit( ( unsigned char )lx->s[ lx->p ] ) ) vi = vi * 10 + ( uint64_t )( lx->s[ lx->p++ ] - '0' );
    if( lx->p < lx->n && lx->s[ lx->p ] == '.' )
    {
        lx->p++; is_int = 0;
        double f = 0.1;
        while( lx->p < lx->n && isdigit( ( unsigned char )lx->s[ lx->p ] ) ) { vf += f * ( lx->s[ lx->p++ ] - '0' ); f *= 0.1; }
    }
    if( lx->p < lx->n && ( lx->s[ lx->p ] == 'e' || lx->s[ lx->p ] == 'E' ) )
    {
        lx->p++; is_int = 0;
        int neg = 0;
        if( lx->p < lx->n && ( lx->s[ lx->p ] == '+' || lx->s[ lx->p ] == '-' ) ) neg = lx->s[ lx->p++ ] == '-';
        if( !( lx->p < lx->n && isdigit( ( unsigned char )lx->s[ lx->p ] ) ) ) lex_fail( lx, "Exponent expected." );
        while( lx->p < lx->n && isdigit( ( unsigned char )lx->s[ lx->p ] ) ) vx = vx * 10 + ( lx->s[ lx->p++ ] - '0' );
        if( neg ) vx = -vx;
    }
    token* t = push_tok( b, lx, TK_DATA );
    if( is_int ) t->lit = v_int( ( int64_t )vi );
    else { double v = ( double )vi + vf; v *= pow( 10.0, ( double )vx ); t->lit = v_float( v ); }
}

static void parse_string_literal( lexer* lx, char** out )
{
    size_t cap = 64, n = 0; char* s = xalloc( cap );
    for( ;; )
    {
        if( lx->p >= lx->n ) lex_fail( lx, "Stream ends in string literal" );
        char c = lx->s[ lx->p++ ];
        if( c == '"' ) break;
        if( c == '\n' ) lx->line++;
        if( c == '\\' && lx->p < lx->n )
        {
            char d = lx->s[ lx->p ];
            if( d == '"' ) { c = '"'; lx->p++; } else if( d == 'n' ) { c = '\n'; lx->p++; } else if( d == 'r' ) { c = '\r'; lx->p++; }
            else if( d == 't' ) { c = '\t'; lx->p++; } else if( d == '0' ) { c = 0; lx->p++; } else if( d == '\\' ) { c = '\\'; lx->p++; }
        }
        if( n + 2 > cap ) { cap *= 2; s = xrealloc( s, cap ); }
        s[ n++ ] = c;
    }
    s[ n ] = 0;
    *out = s;
}

static void parse_include( lexer* lx, block* b )
{
    skip_space( lx );
    if( !match( lx, "\"" ) ) lex_fail( lx, "File name expected." );
    char* file; parse_string_literal( lx, &file );
    if( !file[ 0 ] ) lex_fail( lx, "File name expected." );
    char* path = file;
    if( file[ 0 ] != '/' )   /* relative to the including file (interpreter.c:479-491) */
    {
        const char* cur = lx->ip->files[ lx->file ];
        const char* slash = strrchr( cur, '/' );
        if( slash )
        {
            size_t dl = ( size_t )( slash - cur );
            path = xalloc( dl + strlen( file ) + 2 );
            memcpy( path, cur, dl ); path[ dl ] = '/'; strcpy( path + dl + 1, file );
            free( file );
        }
    }
    size_t size; char* text = read_file( path, &size );
    if( !text ) { char msg[ 600 ]; snprintf( msg, sizeof( msg ), "Cannot open '%s'.", path ); lex_fail( lx, msg ); }
    lexer sub = { lx->ip, text, 0, size, add_file( lx->ip, path ), 1 };
    parse_block( &sub, b );
    if( sub.p < sub.n ) lex_fail( &sub, "Unexpected '}'." );
    free( text ); free( path );
}

static void parse_block( lexer* lx, block* b )
{
    skip_space( lx );
    while( lx->p < lx->n )
    {
        char c = lx->s[ lx->p ];
        if( isdigit( ( unsigned char )c ) ) parse_number( lx, b );
        else if( c == '"' )
        {
            lx->p++;
            token* t = push_tok( b, lx, TK_DATA );
            char* s; parse_string_literal( lx, &s );
            t = &b->t[ b->n - 1 ];
            t->lit = v_new( V_STR ); t->lit->u.s = s;
        }
        else if( isalpha( ( unsigned char )c ) || c == '_' )
        {
            char name[ 256 ]; size_t l = 0;
            while( lx->p < lx->n && ( isalnum( ( unsigned char )lx->s[ lx->p ] ) || lx->s[ lx->p ] == '_' ) ) { if( l < sizeof( name ) - 1 ) name[ l++ ] = lx->s[ lx->p ]; lx->p++; }
            name[ l ] = 0;
            if(      !strcmp( name, "true"  ) ) push_tok( b, lx, TK_DATA )->lit = v_bool( 1 );
            else if( !strcmp( name, "false" ) ) push_tok( b, lx, TK_DATA )->lit = v_bool( 0 );
            else if( !strcmp( name, "AND"   ) ) push_tok( b, lx, TK_AND );
            else if( !strcmp( name, "OR"    ) ) push_tok( b, lx, TK_OR );
            else if( !strcmp( name, "XOR"   ) ) push_tok( b, lx, TK_XOR );
            else if( !strcmp( name, "NOT"   ) ) push_tok( b, lx, TK_NOT );
            else if( !strcmp( name, "CAT"   ) ) push_tok( b, lx, TK_CAT );
            else if( !strcmp( name, "def"   ) ) push_tok( b, lx, TK_DEF );
            else if( !strcmp( name, "if"    ) ) push_tok( b, lx, TK_IF );
            else if( !strcmp( name, "while" ) ) push_tok( b, lx, TK_WHILE );
            else if( !strcmp( name, "for"   ) ) push_tok( b, lx, TK_FOR );
            else if( !strcmp( name, "in"    ) ) push_tok( b, lx, TK_IN );
            else if( !strcmp( name, "else"  ) ) push_tok( b, lx, TK_ELSE );
            else
            {
                int st = -1;
                for( int i = 0; sig_types_g[ i ].name; i++ ) if( !strcmp( name, sig_types_g[ i ].name ) ) st = sig_types_g[ i ].type;
                if( st >= 0 ) { val* v = v_new( V_TYPE ); v->u.mtype = st; push_tok( b, lx, TK_DATA )->lit = v; }
                else push_tok( b, lx, TK_NAME )->name = intern( lx->ip, name );
            }
        }
        else if( strchr( "!?.=+-*/%><&|:", c ) )
        {
            lx->p++;
            int k = 0;
            switch( c )
            {
                case '!': k = TK_NOT; break;
                case '?': k = match( lx, "?" ) ? TK_DQUERY : TK_QUERY; break;
                case '.': k = TK_DOT; break;
                case '=': k = match( lx, "=" ) ? TK_EQ : TK_ASSIGN; break;
                case '+': k = match( lx, "=" ) ? TK_ADD_ASG : TK_ADD; break;
                case '-': k = match( lx, "=" ) ? TK_SUB_ASG : TK_SUB; break;
                case '*': k = match( lx, "=" ) ? TK_MUL_ASG : TK_MUL; break;
                case '/': k = match( lx, "=" ) ? TK_DIV_ASG : TK_DIV; break;
                case '%': k = match( lx, "=" ) ? TK_MOD_ASG : TK_MOD; break;
                case '<': k = match( lx, "=" ) ? TK_LE : match( lx, ">" ) ? TK_NE : match( lx, "-" ) ? TK_FSIG : TK_LT; break;
                case '>': k = match( lx, "=" ) ? TK_GE : TK_GT; break;
                case '&': k = TK_AND; break;
                case '|': k = TK_OR; break;
                case ':': k = TK_CAT; break;
            }
            push_tok( b, lx, k );
        }
        else if( strchr( ";,()[]", c ) )
        {
            lx->p++;
            int k = 0;
            switch( c )
            {
                case ';': k = TK_SEMI; break;
                case ',': k = TK_COMMA; break;
                case '(': k = match( lx, "&)" ) ? TK_ICPS : match( lx, "|)" ) ? TK_OCPS : match( lx, ":)" ) ? TK_CMPD : match( lx, "@)" ) ? TK_ENV : TK_LPAR; break;
                case ')': k = TK_RPAR; break;
                case '[': k = match( lx, "]" ) ? TK_DYNARR : TK_LBRK; break;
                case ']': k = TK_RBRK; break;
            }
            push_tok( b, lx, k );
        }
        else if( c == '{' )
        {
            lx->p++;
            token* t = push_tok( b, lx, TK_DATA );
            block* nb = xalloc( sizeof( block ) );
            size_t ti = b->n - 1;
            parse_block( lx, nb );
            if( !match( lx, "}" ) ) lex_fail( lx, "'}' expected." );
            t = &b->t[ ti ];
            t->blk = nb;
        }
        else if( c == '}' ) break;   /* end of block, not consumed */
        else if( match( lx, "#parse" ) ) parse_include( lx, b );
        else if( match( lx, "#source_file_name" ) ) push_tok( b, lx, TK_DATA )->lit = v_str( lx->ip->files[ lx->file ] );
        else lex_fail( lx, "Syntax error." );
        skip_space( lx );
    }
}

/* ------------------------------------------------------------------------------------------------------------- */
/* transforms over containers (container.c:67-148,283-366)                                                       */

static void any_move( val* v, acn_v3 vec )
{
    if( !v ) return;
    if( v->type == V_OBJ ) acn_obj_move( v->u.obj, vec );
    else if( v->type == V_LIST ) for( size_t i = 0; i < v->u.list.n; i++ ) any_move( v->u.list.d[ i ], vec );
    else if( v->type == V_MAP ) for( size_t i = 0; i < v->u.map.n; i++ ) any_move( v->u.map.d[ i ], vec );
}
static void any_rotate( val* v, const acn_m3* m )
{
    if( !v ) return;
    if( v->type == V_OBJ ) acn_obj_rotate( v->u.obj, m );
    else if( v->type == V_LIST ) for( size_t i = 0; i < v->u.list.n; i++ ) any_rotate( v->u.list.d[ i ], m );
    else if( v->type == V_MAP ) for( size_t i = 0; i < v->u.map.n; i++ ) any_rotate( v->u.map.d[ i ], m );
}
static void any_scale( val* v, double f )
{
    if( !v ) return;
    if( v->type == V_OBJ ) acn_obj_scale( v->u.obj, f );
    else if( v->type == V_LIST ) for( size_t i = 0; i < v->u.list.n; i++ ) any_scale( v->u.list.d[ i ], f );
    else if( v->type == V_MAP ) for( size_t i = 0; i < v->u.map.n; i++ ) any_scale( v->u.map.d[ i ], f );
}

/* ------------------------------------------------------------------------------------------------------------- */
/* arithmetic tables.  Operands are consumed.                                                                    */

static acn_v3 vmlf( acn_v3 a, double f ) { acn_v3 r = { a.x * f, a.y * f, a.z * f }; return r; }
static acn_v3 mmlv( const acn_m3* o, acn_v3 v )
{
    acn_v3 r = { o->x.x * v.x + o->x.y * v.y + o->x.z * v.z, o->y.x * v.x + o->y.y * v.y + o->y.z * v.z, o->z.x * v.x + o->z.y * v.y + o->z.z * v.z };
    return r;
}

static int is_transformable( const val* v ) { return v && ( v->type == V_LIST || v->type == V_MAP || v->type == V_OBJ ); }

static val* op_mul( ev* e, val* a, val* b )   /* interpreter.c:651-785 */
{
    val* r = NULL;
    int ta = a ? a->type : 0, tb = b ? b->type : 0;
    if( ta == V_INT || ta == V_FLOAT || ta == V_BOOL )
    {
        if( tb == V_VEC ) r = v_vec( vmlf( b->u.v, to_f3( a ) ) );
        else if( ta == V_INT && tb == V_INT ) r = v_int( a->u.i * b->u.i );
        else if( ta == V_INT && tb == V_BOOL ) r = v_int( a->u.i * b->u.b );
        else if( ta == V_BOOL && tb == V_INT ) r = v_int( a->u.b * b->u.i );
        else if( ta == V_BOOL && tb == V_BOOL ) r = v_bool( a->u.b && b->u.b );
        else if( tb == V_INT || tb == V_FLOAT || tb == V_BOOL ) r = v_float( to_f3( a ) * to_f3( b ) );
    }
    else if( ta == V_VEC )
    {
        if( tb == V_INT || tb == V_FLOAT || tb == V_BOOL ) r = v_vec( vmlf( a->u.v, to_f3( b ) ) );
        else if( tb == V_VEC ) r = v_float( a->u.v.x * b->u.v.x + a->u.v.y * b->u.v.y + a->u.v.z * b->u.v.z );
    }
    else if( ta == V_MAT )
    {
        if( tb == V_INT || tb == V_FLOAT ) { double f = to_f3( b ); acn_m3 m = { vmlf( a->u.m.x, f ), vmlf( a->u.m.y, f ), vmlf( a->u.m.z, f ) }; r = v_mat( m ); }
        else if( tb == V_VEC ) r = v_vec( mmlv( &a->u.m, b->u.v ) );
        else if( tb == V_MAT ) { acn_m3 m = { mmlv( &a->u.m, b->u.m.x ), mmlv( &a->u.m, b->u.m.y ), mmlv( &a->u.m, b->u.m.z ) }; r = v_mat( m ); }
    }
    else if( ta == V_SIG )
    {
        if( tb == V_CLOSURE )   /* signature * block: a function (interpreter.c:742-757) */
        {
            r = v_new( V_CLOSURE );
            r->u.clo.blk = b->u.clo.blk; r->u.clo.lex = b->u.clo.lex; r->u.clo.sig = v_clone( a );
        }
    }
    else if( is_transformable( a ) )
    {
        if( tb == V_INT || tb == V_FLOAT ) { r = v_clone( a ); any_scale( r, to_f3( b ) ); }
        else if( tb == V_MAT ) { r = v_clone( a ); any_rotate( r, &b->u.m ); }
        else if( tb == V_VEC && is_object( a ) ) r = v_obj( acn_obj_scale_s_create_scale( a->u.obj, b->u.v ) );
    }
    if( !r ) FAIL( e, "Cannot evaluate '%s' * '%s'", vt( a ), vt( b ) );
    v_unref( a ); v_unref( b );
    return r;
}

static val* op_mod( ev* e, val* a, val* b )   /* interpreter.c:789-814 */
{
    if( !( a && b && a->type == V_INT && b->type == V_INT ) ) FAIL( e, "Cannot evaluate '%s' %% '%s'", vt( a ), vt( b ) );
    if( b->u.i == 0 ) FAIL( e, "Modulo by zero." );
    val* r = v_int( a->u.i % b->u.i );
    v_unref( a ); v_unref( b );
    return r;
}

static void fmt_f3( char* buf, size_t n, double f ) { snprintf( buf, n, "%g", f ); }

static val* op_add( ev* e, val* a, val* b )   /* interpreter.c:818-924 */
{
    val* r = NULL;
    int ta = a ? a->type : 0, tb = b ? b->type : 0;
    char buf[ 64 ];
    if( ta == V_INT || ta == V_FLOAT || ta == V_BOOL )
    {
        if( tb == V_STR && ta != V_BOOL )
        {
            if( ta == V_INT ) snprintf( buf, sizeof( buf ), "%lld", ( long long )a->u.i ); else fmt_f3( buf, sizeof( buf ), a->u.f );
            char* s = xalloc( strlen( buf ) + strlen( b->u.s ) + 1 ); strcpy( s, buf ); strcat( s, b->u.s );
            r = v_new( V_STR ); r->u.s = s;
        }
        else if( ta == V_FLOAT && is_num( b ) ) r = v_float( a->u.f + to_f3( b ) );
        else if( tb == V_FLOAT ) r = v_float( to_f3( a ) + b->u.f );
        else if( tb == V_INT || tb == V_BOOL ) r = v_int( ( ta == V_INT ? a->u.i : a->u.b ) + ( tb == V_INT ? b->u.i : b->u.b ) );
    }
    else if( ta == V_VEC ) { if( tb == V_VEC ) { acn_v3 s = { a->u.v.x + b->u.v.x, a->u.v.y + b->u.v.y, a->u.v.z + b->u.v.z }; r = v_vec( s ); } }
    else if( ta == V_STR )
    {
        r = v_clone( a );
        const char* add = "";
        if( tb == V_STR ) add = b->u.s;
        else if( tb == V_INT ) { snprintf( buf, sizeof( buf ), "%lld", ( long long )b->u.i ); add = buf; }
        r->u.s = xrealloc( r->u.s, strlen( r->u.s ) + strlen( add ) + 1 );
        strcat( r->u.s, add );
    }
    else if( is_transformable( a ) ) { if( tb == V_VEC ) { r = v_clone( a ); any_move( r, b->u.v ); } }
    if( !r ) FAIL( e, "Cannot evaluate '%s' + '%s'", vt( a ), vt( b ) );
    v_unref( a ); v_unref( b );
    return r;
}

/* 1: a < b, -1: a > b, 0 equal (interpreter.c:929-983) */
static int op_cmp( ev* e, val* a, val* b )
{
    if( !is_num( a ) || !is_num( b ) ) FAIL( e, "Cannot compare '%s' with '%s'", vt( a ), vt( b ) );
    int r;
    if( a->type != V_FLOAT && b->type != V_FLOAT )
    {
        int64_t x = a->type == V_INT ? a->u.i : a->u.b, y = b->type == V_INT ? b->u.i : b->u.b;
        r = x < y ? 1 : x > y ? -1 : 0;
    }
    else { double x = to_f3( a ), y = to_f3( b ); r = x < y ? 1 : x > y ? -1 : 0; }
    v_unref( a ); v_unref( b );
    return r;
}

static val* op_inverse( ev* e, val* a )   /* interpreter.c:987-1005 */
{
    if( !( a && ( a->type == V_INT || a->type == V_FLOAT ) ) ) FAIL( e, "Cannot invert '%s'", vt( a ) );
    double x = to_f3( a );
    val* r = v_float( x != 0 ? 1.0 / x : INFINITY );
    v_unref( a );
    return r;
}

static val* op_logic( ev* e, int op, val* a, val* b )   /* interpreter.c:1009-1080 */
{
    val* r = NULL;
    if( a && b && a->type == V_BOOL && b->type == V_BOOL )
    {
        int x = a->u.b, y = b->u.b;
        r = v_bool( op == TK_AND ? ( x && y ) : op == TK_OR ? ( x || y ) : ( ( x && !y ) || ( !x && y ) ) );
    }
    else if( op != TK_XOR && is_object( a ) && is_object( b ) )
    {
        r = v_obj( op == TK_AND ? acn_obj_pair_inside_s_create_pair( a->u.obj, b->u.obj ) : acn_obj_pair_outside_s_create_pair( a->u.obj, b->u.obj ) );
    }
    if( !r ) FAIL( e, "Cannot evaluate '%s' %s '%s'", vt( a ), op == TK_AND ? "AND" : op == TK_OR ? "OR" : "XOR", vt( b ) );
    v_unref( a ); v_unref( b );
    return r;
}

static val* op_not( ev* e, val* a )   /* interpreter.c:1084-1105 */
{
    val* r = NULL;
    if( a && a->type == V_BOOL ) r = v_bool( !a->u.b );
    else if( is_object( a ) ) r = v_obj( acn_obj_neg_s_create_neg( a->u.obj ) );
    if( !r ) FAIL( e, "Cannot evaluate NOT '%s'", vt( a ) );
    v_unref( a );
    return r;
}

static val* op_cat( val* a, val* b )   /* interpreter.c:1204-1231 */
{
    val* r;
    if( a->type == V_LIST )
    {
        r = v_clone( a );
        if( b->type == V_LIST ) for( size_t i = 0; i < b->u.list.n; i++ ) list_push_owned( r, v_clone( b->u.list.d[ i ] ) );
        else list_push_owned( r, v_clone( b ) );
    }
    else
    {
        r = v_new( V_LIST );
        list_push_owned( r, v_clone( a ) );
        list_push_owned( r, v_clone( b ) );
    }
    v_unref( a ); v_unref( b );
    return r;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* list -> composite / compound (container.c:368-421)                                                            */

static val* list_composite( ev* e, val* l, int inside )
{
    size_t n = l->u.list.n;
    if( n == 0 ) FAIL( e, "Cannot create a composite of an empty list." );
    for( size_t i = 0; i < n; i++ ) if( !is_object( l->u.list.d[ i ] ) ) FAIL( e, "Composite: list element %zu is '%s', not an object.", i, vt( l->u.list.d[ i ] ) );
    if( n == 1 ) return v_ref( l->u.list.d[ 0 ] );   /* the element itself, not a copy (container.c:371-374) */
    acn_obj** a = xalloc( n * sizeof( acn_obj* ) );
    for( size_t i = 0; i < n; i++ ) a[ i ] = l->u.list.d[ i ]->u.obj;
    acn_obj* o = inside ? acn_create_inside_composite( a, n ) : acn_create_outside_composite( a, n );
    free( a );
    return v_obj( o );
}

/* compound_s_push_q (compound.c:140-203): objects and compounds by the compound's own rules, maps and lists
 * element by element */
static void compound_push_any( ev* e, acn_obj* c, const val* x )
{
    if( !x ) return;
    if( x->type == V_OBJ ) acn_compound_s_push( c, x->u.obj );
    else if( x->type == V_LIST ) for( size_t i = 0; i < x->u.list.n; i++ ) compound_push_any( e, c, x->u.list.d[ i ] );
    else if( x->type == V_MAP ) for( size_t i = 0; i < x->u.map.n; i++ ) compound_push_any( e, c, x->u.map.d[ i ] );
    else FAIL( e, "Cannot push object %s to compound_s.", vt( x ) );
}

static val* list_compound( ev* e, val* l )
{
    val* c = v_obj( acn_compound_s_create() );
    for( size_t i = 0; i < l->u.list.n; i++ ) compound_push_any( e, c->u.obj, l->u.list.d[ i ] );
    return c;
}

static void auto_envelope( ev* e, acn_obj* o )
{
    if( e->ip->opts->auto_envelope == ACN_AUTOENV_SKIP ) return;
    if( acn_obj_set_auto_envelope( o ) != ACN_OK ) FAIL( e, "set_auto_envelope failed: %s", acn_last_error() );
}

/* ------------------------------------------------------------------------------------------------------------- */
/* evaluator                                                                                                     */

static val* eval( ev* e, val* front );
static val* execute( ev* e );

static int peek( const ev* e ) { return e->ix < e->blk->n ? e->blk->t[ e->ix ].kind : TK_END; }
static int try_tok( ev* e, int k ) { if( peek( e ) == k ) { e->ix++; return 1; } return 0; }

static const char* tok_symbol( int k )
{
    switch( k )
    {
        case TK_COMMA: return ","; case TK_SEMI: return ";"; case TK_LPAR: return "("; case TK_RPAR: return ")";
        case TK_LBRK: return "["; case TK_RBRK: return "]"; case TK_NAME: return "name"; case TK_IN: return "in";
        case TK_ASSIGN: return "=";
    }
    return "?";
}
static void expect( ev* e, int k ) { if( !try_tok( e, k ) ) FAIL( e, "'%s' expected.", tok_symbol( k ) ); }
static const char* expect_name( ev* e )
{
    if( peek( e ) != TK_NAME ) FAIL( e, "Name expected." );
    return e->blk->t[ e->ix++ ].name;
}

static val* eval_req( ev* e )
{
    val* v = eval( e, NULL );
    if( !v ) FAIL( e, "Expression yields no value." );
    return v;
}
static double eval_f3( ev* e )   /* interpreter.c:1318-1334 */
{
    val* v = eval( e, NULL );
    if( !v || !( v->type == V_INT || v->type == V_FLOAT ) ) FAIL( e, "Scalar expected." );
    double r = to_f3( v ); v_unref( v ); return r;
}
static acn_v3 eval_v3d( ev* e )   /* interpreter.c:1305-1314 */
{
    val* v = eval( e, NULL );
    if( !v || !( v->type == V_VEC || v->type == V_COLOR ) ) FAIL( e, "Vector expected." );
    acn_v3 r = v->u.v; v_unref( v ); return r;
}
static acn_m3 eval_rot( ev* e )
{
    val* v = eval( e, NULL );
    if( !v || v->type != V_MAT ) FAIL( e, "Rotation expected." );
    acn_m3 r = v->u.m; v_unref( v ); return r;
}
static val* eval_string( ev* e )
{
    val* v = eval( e, NULL );
    if( !v || v->type != V_STR ) FAIL( e, "String expected." );
    return v;
}

/* ---- built-in functions (closures.c) ---- */
enum bi
{
    BI_VEC, BI_VECX, BI_VECY, BI_VECZ, BI_ROTX, BI_ROTY, BI_ROTZ, BI_COLOR, BI_COLR, BI_COLG, BI_COLB,
    BI_SQRT, BI_SQR, BI_EXP, BI_LOG, BI_TO_DEG, BI_TO_RAD, BI_SIN, BI_COS, BI_TAN, BI_SIN_D, BI_COS_D, BI_TAN_D,
    BI_ASIN, BI_ACOS, BI_ATAN, BI_POW, BI_FLOOR, BI_CEILING,
    BI_FILE_EXISTS, BI_FILE_TOUCH, BI_FILE_DELETE, BI_FILE_RENAME,
    BI_PLANE, BI_SPHERE, BI_SQUAROID, BI_CYLINDER, BI_TORUS, BI_HYP1, BI_HYP2, BI_ELLIPSOID, BI_CONE,
    BI_STRING_FA, BI_STRING_TO_NUM, BI_BETH_OBJECT, BI_GET_TIME
};
/* args: n = num, s = string, a = anything */
static const struct { const char* name; int id; const char* args; } builtins_g[] =
{
    { "vec", BI_VEC, "nnn" }, { "vecx", BI_VECX, "n" }, { "vecy", BI_VECY, "n" }, { "vecz", BI_VECZ, "n" },
    { "rotx", BI_ROTX, "n" }, { "roty", BI_ROTY, "n" }, { "rotz", BI_ROTZ, "n" },
    { "color", BI_COLOR, "nnn" }, { "colr", BI_COLR, "n" }, { "colg", BI_COLG, "n" }, { "colb", BI_COLB, "n" },
    { "sqrt", BI_SQRT, "n" }, { "sqr", BI_SQR, "n" }, { "exp", BI_EXP, "n" }, { "log", BI_LOG, "n" },
    { "to_deg", BI_TO_DEG, "n" }, { "to_rad", BI_TO_RAD, "n" }, { "sin", BI_SIN, "n" }, { "cos", BI_COS, "n" }, { "tan", BI_TAN, "n" },
    { "sin_d", BI_SIN_D, "n" }, { "cos_d", BI_COS_D, "n" }, { "tan_d", BI_TAN_D, "n" },
    { "asin", BI_ASIN, "n" }, { "acos", BI_ACOS, "n" }, { "atan", BI_ATAN, "n" }, { "pow", BI_POW, "nn" },
    { "floor", BI_FLOOR, "n" }, { "ceiling", BI_CEILING, "n" },
    { "file_exists", BI_FILE_EXISTS, "s" }, { "file_touch", BI_FILE_TOUCH, "s" }, { "file_delete", BI_FILE_DELETE, "s" }, { "file_rename", BI_FILE_RENAME, "ss" },
    { "create_plane", BI_PLANE, "" }, { "create_sphere", BI_SPHERE, "n" }, { "create_squaroid", BI_SQUAROID, "nnnn" },
    { "create_cylinder", BI_CYLINDER, "nn" }, { "create_torus", BI_TORUS, "nn" }, { "create_hyperboloid1", BI_HYP1, "nnn" },
    { "create_hyperboloid2", BI_HYP2, "nnn" }, { "create_ellipsoid", BI_ELLIPSOID, "nnn" }, { "create_cone", BI_CONE, "nnn" },
    { "string_fa", BI_STRING_FA, "sa" }, { "string_to_num", BI_STRING_TO_NUM, "s" }, { "beth_object", BI_BETH_OBJECT, "s" },
    { "get_time", BI_GET_TIME, "" },
    { NULL, 0, NULL }
};

static acn_v3 V3( double x, double y, double z ) { acn_v3 v = { x, y, z }; return v; }

/* string_fa: beth's st_s_create_fa with ONE argument.  Provided: "#<s3_t*>", "#<f3_t*>" and left padding
 * "#pl<N>'<c>'{ ... }" around them -- what the shipped scripts use (diamond_video.acn:197). */
static val* string_fa( ev* e, const char* fmt, const val* arg )
{
    char out[ 1024 ]; size_t n = 0;
    for( const char* p = fmt; *p; )
    {
        if( n >= sizeof( out ) - 320 ) FAIL( e, "string_fa: result longer than %d characters.", ( int )sizeof( out ) - 320 );
        if( *p != '#' ) { out[ n++ ] = *p++; continue; }
        int pad = 0; char padc = ' ';
        const char* q = p + 1;
        int braced = 0;
        if( q[ 0 ] == 'p' && q[ 1 ] == 'l' )
        {
            q += 2;
            while( isdigit( ( unsigned char )*q ) ) pad = pad * 10 + ( *q++ - '0' );
            if( q[ 0 ] == '\'' && q[ 1 ] && q[ 2 ] == '\'' ) { padc = q[ 1 ]; q += 3; }
            if( *q == '{' ) { braced = 1; q++; }
            if( *q != '#' ) FAIL( e, "string_fa: unsupported format '%s'.", fmt );
            q++;
        }
        char item[ 256 ];
        if( !strncmp( q, "<s3_t*>", 7 ) ) { if( !is_num( arg ) ) FAIL( e, "string_fa: number expected." ); snprintf( item, sizeof( item ), "%lld", ( long long )( arg->type == V_FLOAT ? ( int64_t )arg->u.f : arg->type == V_INT ? arg->u.i : arg->u.b ) ); q += 7; }
        else if( !strncmp( q, "<f3_t*>", 7 ) ) { if( !is_num( arg ) ) FAIL( e, "string_fa: number expected." ); fmt_f3( item, sizeof( item ), to_f3( arg ) ); q += 7; }
        else if( !strncmp( q, "<sc_t>", 6 ) ) { if( arg->type != V_STR ) FAIL( e, "string_fa: string expected." ); if( strlen( arg->u.s ) >= sizeof( item ) ) FAIL( e, "string_fa: string argument longer than %d characters.", ( int )sizeof( item ) - 1 ); snprintf( item, sizeof( item ), "%s", arg->u.s ); q += 6; }
        else FAIL( e, "string_fa: unsupported format '%s'.", fmt );
        if( braced ) { if( *q != '}' ) FAIL( e, "string_fa: unsupported format '%s'.", fmt ); q++; }
        if( pad > 255 ) FAIL( e, "string_fa: padding wider than 255." );
        for( int l = ( int )strlen( item ); l < pad; l++ ) out[ n++ ] = padc;
        n += ( size_t )snprintf( out + n, sizeof( out ) - n, "%s", item );
        p = q;
    }
    out[ n ] = 0;
    return v_str( out );
}

static val* call_builtin( ev* e, int id, val** a )
{
    #define N( i ) to_f3( a[ i ] )
    switch( id )
    {
        case BI_VEC:   return v_vec( V3( N( 0 ), N( 1 ), N( 2 ) ) );
        case BI_VECX:  return v_vec( V3( N( 0 ), 0, 0 ) );
        case BI_VECY:  return v_vec( V3( 0, N( 0 ), 0 ) );
        case BI_VECZ:  return v_vec( V3( 0, 0, N( 0 ) ) );
        case BI_ROTX:  return v_mat( acn_rotx( N( 0 ) ) );
        case BI_ROTY:  return v_mat( acn_roty( N( 0 ) ) );
        case BI_ROTZ:  return v_mat( acn_rotz( N( 0 ) ) );
        case BI_COLOR: return v_color( V3( N( 0 ), N( 1 ), N( 2 ) ) );
        case BI_COLR:  return v_color( V3( N( 0 ), 0, 0 ) );
        case BI_COLG:  return v_color( V3( 0, N( 0 ), 0 ) );
        case BI_COLB:  return v_color( V3( 0, 0, N( 0 ) ) );
        case BI_SQRT:  return v_float( sqrt( N( 0 ) ) );
        case BI_SQR:   { double x = N( 0 ); return v_float( x * x ); }
        case BI_EXP:   return v_float( exp( N( 0 ) ) );
        case BI_LOG:   return v_float( log( N( 0 ) ) );
        case BI_TO_DEG: return v_float( N( 0 ) * 180.0 / M_PI );
        case BI_TO_RAD: return v_float( N( 0 ) * M_PI / 180.0 );
        case BI_SIN:   return v_float( sin( N( 0 ) ) );
        case BI_COS:   return v_float( cos( N( 0 ) ) );
        case BI_TAN:   return v_float( tan( N( 0 ) ) );
        case BI_SIN_D: return v_float( sin( M_PI * N( 0 ) / 180.0 ) );
        case BI_COS_D: return v_float( cos( M_PI * N( 0 ) / 180.0 ) );
        case BI_TAN_D: return v_float( tan( M_PI * N( 0 ) / 180.0 ) );
        case BI_ASIN:  return v_float( asin( N( 0 ) ) );
        case BI_ACOS:  return v_float( acos( N( 0 ) ) );
        case BI_ATAN:  return v_float( atan( N( 0 ) ) );
        case BI_POW:   return v_float( pow( N( 0 ), N( 1 ) ) );
        case BI_FLOOR: return v_float( floor( N( 0 ) ) );
        case BI_CEILING: return v_float( ceil( N( 0 ) ) );
        case BI_FILE_EXISTS: return v_bool( access( a[ 0 ]->u.s, F_OK ) == 0 );
        case BI_FILE_TOUCH: case BI_FILE_DELETE: case BI_FILE_RENAME:
            if( e->ip->opts->readonly_fs ) return v_bool( 0 );
            if( id == BI_FILE_DELETE ) return v_bool( remove( a[ 0 ]->u.s ) == 0 );
            if( id == BI_FILE_RENAME ) return v_bool( rename( a[ 0 ]->u.s, a[ 1 ]->u.s ) == 0 );
            { FILE* f = fopen( a[ 0 ]->u.s, "ab" ); if( f ) fclose( f ); return v_bool( f != NULL ); }
        case BI_PLANE:     return v_obj( acn_obj_plane_s_create() );
        case BI_SPHERE:    return v_obj( acn_obj_sphere_s_create( N( 0 ) ) );
        case BI_SQUAROID:  return v_obj( acn_obj_squaroid_s_create_squaroid( N( 0 ), N( 1 ), N( 2 ), N( 3 ) ) );
        case BI_CYLINDER:  return v_obj( acn_obj_squaroid_s_create_cylinder( N( 0 ), N( 1 ) ) );
        case BI_TORUS:     return v_obj( acn_obj_torus_create( N( 0 ), N( 1 ) ) );
        case BI_HYP1:      return v_obj( acn_obj_squaroid_s_create_hyperboloid1( N( 0 ), N( 1 ), N( 2 ) ) );
        case BI_HYP2:      return v_obj( acn_obj_squaroid_s_create_hyperboloid2( N( 0 ), N( 1 ), N( 2 ) ) );
        case BI_ELLIPSOID: return v_obj( acn_obj_squaroid_s_create_ellipsoid( N( 0 ), N( 1 ), N( 2 ) ) );
        case BI_CONE:      return v_obj( acn_obj_squaroid_s_create_cone( N( 0 ), N( 1 ), N( 2 ) ) );
        case BI_STRING_FA: return string_fa( e, a[ 0 ]->u.s, a[ 1 ] );
        case BI_STRING_TO_NUM:   /* closures.c:155-182 */
        {
            const char* s = a[ 0 ]->u.s;
            while( *s == ' ' || *s == '\t' || *s == '\n' ) s++;
            size_t l = strspn( s, "+-0123456789eE." );
            int is_float = 0;
            for( size_t i = 0; i < l; i++ ) if( s[ i ] == '.' || s[ i ] == 'e' || s[ i ] == 'E' ) is_float = 1;
            return is_float ? v_float( strtod( s, NULL ) ) : v_int( strtoll( s, NULL, 10 ) );
        }
        case BI_BETH_OBJECT:   /* closures.c:446-456 */
        {
            const char* name = a[ 0 ]->u.s;
            if( !strcmp( name, "obj_distance_s" ) ) return v_obj( acn_obj_distance_s_create() );
            for( int k = 0; k < BETH_KINDS; k++ )
            {
                if( strcmp( name, beth_kinds_g[ k ] ) ) continue;
                val* v = v_new( V_BETH );   /* zeroed: colors 0 */
                v->u.beth.kind = k;
                v->u.beth.scale = 1.0;      /* textures.c:137 */
                v->u.beth.ex_radius = 0.5;  /* distance.c:73 */
                return v;
            }
            FAIL( e, "beth_object( \"%s\" ): beth's object registry is not available; txm_plain_s, txm_chess_s, distance_sphere_s, distance_torus_s and obj_distance_s are.", name );
            return NULL;
        }
        case BI_GET_TIME: return v_float( ( double )( clock() - e->ip->start ) / CLOCKS_PER_SEC );
    }
    #undef N
    FAIL( e, "Unknown built-in." );
    return NULL;
}

static int sig_accepts( int st, const val* v )   /* interpreter.c:1389-1399 */
{
    if( st == S_ANY ) return 1;
    if( !v ) return 0;
    switch( st )
    {
        case S_BOOL: return v->type == V_BOOL;
        case S_INT: return v->type == V_INT;
        case S_FLOAT: return v->type == V_FLOAT;
        case S_NUM: return is_num( v );
        case S_STRING: return v->type == V_STR;
        case S_MAP: return v->type == V_MAP;
        case S_LIST: return v->type == V_LIST;
        case S_OBJECT: return v->type == V_OBJ;
        case S_V3D: return v->type == V_VEC;
        case S_FUNC: return v->type == V_CLOSURE || v->type == V_BUILTIN;
    }
    return 0;
}

static const char* sig_type_name( int st )
{
    for( int i = 0; sig_types_g[ i ].name; i++ ) if( sig_types_g[ i ].type == st ) return sig_types_g[ i ].name;
    return "any";
}

#define ACN_MAX_ARGS 16
#define ACN_MAX_DEPTH 200

/* `f( a, b )` (interpreter.c:1374-1407, 1896-1923) */
static val* eval_call( ev* e, val* fn )
{
    if( !( fn->type == V_CLOSURE || fn->type == V_BUILTIN ) ) FAIL( e, "'%s' is no function.", vt( fn ) );
    expect( e, TK_LPAR );
    val* args[ ACN_MAX_ARGS ] = { 0 };
    size_t nargs;
    const char* bargs = NULL;
    if( fn->type == V_BUILTIN ) { bargs = builtins_g[ fn->u.builtin ].args; nargs = strlen( bargs ); }
    else nargs = fn->u.clo.sig ? fn->u.clo.sig->u.sig.n : 0;
    if( nargs > ACN_MAX_ARGS ) FAIL( e, "Too many arguments." );
    for( size_t i = 0; i < nargs; i++ )
    {
        if( i > 0 ) expect( e, TK_COMMA );
        args[ i ] = eval( e, NULL );
        int ok;
        if( bargs ) ok = args[ i ] && ( bargs[ i ] == 'n' ? is_num( args[ i ] ) : bargs[ i ] == 's' ? args[ i ]->type == V_STR : 1 );
        else ok = sig_accepts( fn->u.clo.sig->u.sig.a[ i ].type, args[ i ] );
        if( !ok ) FAIL( e, "Function argument %zu is '%s' and not of '%s'.", i + 1, vt( args[ i ] ),
                        bargs ? ( bargs[ i ] == 'n' ? "num" : bargs[ i ] == 's' ? "string" : "any" ) : sig_type_name( fn->u.clo.sig->u.sig.a[ i ].type ) );
    }
    val* ret;
    if( fn->type == V_BUILTIN )
    {
        ret = call_builtin( e, builtins_g[ fn->u.builtin ].id, args );
        for( size_t i = 0; i < nargs; i++ ) v_unref( args[ i ] );
    }
    else
    {
        block* blk = fn->u.clo.blk;
        frame* local = &blk->local;
        local->ext = fn->u.clo.lex ? fn->u.clo.lex : e->fr;
        frame_clear( local );
        for( size_t i = 0; i < nargs; i++ ) frame_set( local, fn->u.clo.sig->u.sig.a[ i ].name, args[ i ] );   /* by reference */
        if( ++e->ip->depth > ACN_MAX_DEPTH ) FAIL( e, "Call depth exceeds %d.", ACN_MAX_DEPTH );
        ev sub = { blk, 0, local, e->ip };
        ret = execute( &sub );
        e->ip->depth--;
    }
    expect( e, TK_RPAR );
    return ret;
}

/* ---- members ---- */

enum { F_U64, F_I64, F_F64, F_VEC, F_COLOR };
typedef struct scene_field { const char* name; int kind; size_t off; } scene_field;
#define SF( name, kind, member ) { name, kind, offsetof( acn_scene, member ) }
static const scene_field scene_fields_g[] =
{
    SF( "threads", F_U64, threads ), SF( "image_width", F_U64, prm.image_width ), SF( "image_height", F_U64, prm.image_height ),
    SF( "gamma", F_F64, prm.gamma ), SF( "gradient_threshold", F_F64, gradient_threshold ),
    SF( "gradient_samples", F_U64, gradient_samples ), SF( "gradient_cycles", F_U64, gradient_cycles ),
    SF( "background_color", F_COLOR, prm.background_color ), SF( "camera_position", F_VEC, prm.camera_position ),
    SF( "camera_view_direction", F_VEC, prm.camera_view_direction ), SF( "camera_top_direction", F_VEC, prm.camera_top_direction ),
    SF( "camera_focal_length", F_F64, prm.camera_focal_length ), SF( "trace_depth", F_U64, prm.trace_depth ),
    SF( "trace_min_intensity", F_F64, prm.trace_min_intensity ), SF( "direct_samples", F_U64, prm.direct_samples ),
    SF( "path_samples", F_U64, prm.path_samples ), SF( "max_path_length", F_F64, prm.max_path_length ),
    SF( "experimental_level", F_I64, prm.experimental_level ),
    { NULL, 0, 0 }
};

static void scene_push_any( ev* e, acn_scene* s, const val* v )   /* scene.c:238-279 */
{
    if( !v ) return;
    if( v->type == V_OBJ ) acn_scene_s_push( s, v->u.obj );
    else if( v->type == V_MAP ) for( size_t i = 0; i < v->u.map.n; i++ ) scene_push_any( e, s, v->u.map.d[ i ] );
    else if( v->type == V_LIST ) for( size_t i = 0; i < v->u.list.n; i++ ) scene_push_any( e, s, v->u.list.d[ i ] );
}

static val* scene_member( ev* e, val* front, const char* key )
{
    acn_scene* s = front->u.scene;
    for( const scene_field* f = scene_fields_g; f->name; f++ )
    {
        if( strcmp( f->name, key ) ) continue;
        char* p = ( char* )s + f->off;
        if( try_tok( e, TK_ASSIGN ) )   /* bcore_via nset with conversion (interpreter.c:1488-1492) */
        {
            val* v = eval_req( e );
            switch( f->kind )
            {
                case F_U64: if( !is_num( v ) ) FAIL( e, "scene.%s: number expected.", key ); *( uint64_t* )p = v->type == V_FLOAT ? ( uint64_t )v->u.f : ( uint64_t )( v->type == V_INT ? v->u.i : v->u.b ); break;
                case F_I64: if( !is_num( v ) ) FAIL( e, "scene.%s: number expected.", key ); *( int64_t* )p = v->type == V_FLOAT ? ( int64_t )v->u.f : ( v->type == V_INT ? v->u.i : v->u.b ); break;
                case F_F64: if( !is_num( v ) ) FAIL( e, "scene.%s: number expected.", key ); *( double* )p = to_f3( v ); break;
                default:
                    if( !( v->type == V_VEC || v->type == V_COLOR ) ) FAIL( e, "scene.%s: vector expected.", key );
                    ( ( double* )p )[ 0 ] = v->u.v.x; ( ( double* )p )[ 1 ] = v->u.v.y; ( ( double* )p )[ 2 ] = v->u.v.z;
                    break;
            }
            v_unref( v );
            return v_ref( front );
        }
        switch( f->kind )
        {
            case F_U64: return v_int( ( int64_t )*( uint64_t* )p );
            case F_I64: return v_int( *( int64_t* )p );
            case F_F64: return v_float( *( double* )p );
            case F_VEC: return v_vec( V3( ( ( double* )p )[ 0 ], ( ( double* )p )[ 1 ], ( ( double* )p )[ 2 ] ) );
            default:    return v_color( V3( ( ( double* )p )[ 0 ], ( ( double* )p )[ 1 ], ( ( double* )p )[ 2 ] ) );
        }
    }
    if( !strcmp( key, "clear" ) ) { expect( e, TK_LPAR ); expect( e, TK_RPAR ); acn_scene_s_clear( s ); return NULL; }
    if( !strcmp( key, "push" ) )
    {
        expect( e, TK_LPAR );
        val* v = eval( e, NULL );
        scene_push_any( e, s, v );
        v_unref( v );
        expect( e, TK_RPAR );
        return NULL;
    }
    if( !strcmp( key, "create_image" ) )
    {
        expect( e, TK_LPAR );
        val* f = eval_string( e );
        int st = e->ip->opts->on_create_image ? e->ip->opts->on_create_image( e->ip->opts->ctx, s, f->u.s )
                                              : acn_scene_s_create_image_file( s, f->u.s );
        if( st != ACN_OK ) FAIL( e, "create_image( \"%s\" ) failed with status %d: %s", f->u.s, st, acn_last_error() );
        v_unref( f );
        expect( e, TK_RPAR );
        return NULL;
    }
    FAIL( e, "scene_s has no member '%s'.", key );
    return NULL;
}

static val* vec_member( ev* e, val* front, const char* key )
{
    double* c = !strcmp( key, "x" ) ? &front->u.v.x : !strcmp( key, "y" ) ? &front->u.v.y : !strcmp( key, "z" ) ? &front->u.v.z : NULL;
    if( !c ) FAIL( e, "Object '%s' has no element named '%s'.", vt( front ), key );
    if( try_tok( e, TK_ASSIGN ) ) { val* v = eval_req( e ); if( !is_num( v ) ) FAIL( e, "Scalar expected." ); *c = to_f3( v ); v_unref( v ); return v_ref( front ); }
    return v_float( *c );
}

/* reflected members of a beth_object() value (interpreter.c:1486-1497: bcore_via get / set by name) */
static val* beth_member( ev* e, val* front, const char* key )
{
    const int kind = front->u.beth.kind;
    acn_v3* c = NULL; double* f = NULL;
    if( kind == BETH_TXM_PLAIN && !strcmp( key, "color" ) ) c = &front->u.beth.c1;
    else if( kind == BETH_TXM_CHESS && !strcmp( key, "color1" ) ) c = &front->u.beth.c1;
    else if( kind == BETH_TXM_CHESS && !strcmp( key, "color2" ) ) c = &front->u.beth.c2;
    else if( kind == BETH_TXM_CHESS && !strcmp( key, "scale" ) ) f = &front->u.beth.scale;
    else if( kind == BETH_DISTANCE_TORUS && !strcmp( key, "ex_radius" ) ) f = &front->u.beth.ex_radius;
    else FAIL( e, "Object '%s' has no element named '%s'.", vt( front ), key );
    if( try_tok( e, TK_ASSIGN ) )
    {
        val* v = eval_req( e );
        if( c ) { if( v->type != V_COLOR ) FAIL( e, "Color expected." ); *c = v->u.v; }
        else    { if( !is_num( v ) ) FAIL( e, "Scalar expected." ); *f = to_f3( v ); }
        v_unref( v );
        return v_ref( front );
    }
    return c ? v_color( *c ) : v_float( *f );
}

/* move / rotate / scale shared by map, list, compound and objects */
static int transform_member( ev* e, val* front, const char* key )
{
    if( !strcmp( key, "move" ) ) { expect( e, TK_LPAR ); acn_v3 v = eval_v3d( e ); any_move( front, v ); expect( e, TK_RPAR ); return 1; }
    if( !strcmp( key, "rotate" ) ) { expect( e, TK_LPAR ); acn_m3 m = eval_rot( e ); any_rotate( front, &m ); expect( e, TK_RPAR ); return 1; }
    if( !strcmp( key, "scale" ) ) { expect( e, TK_LPAR ); double f = eval_f3( e ); any_scale( front, f ); expect( e, TK_RPAR ); return 1; }
    return 0;
}

static val* map_member( ev* e, val* front, const char* key )   /* container.c:156-231 */
{
    val** s = map_slot( front, key );
    if( s ) return v_ref( *s );
    if( try_tok( e, TK_ASSIGN ) )
    {
        val* v = eval( e, NULL );
        val* c = v_clone( v );
        v_unref( v );
        map_set_owned( front, key, c );
        return v_ref( c );
    }
    if( transform_member( e, front, key ) ) return NULL;
    if( !strcmp( key, "has" ) )
    {
        expect( e, TK_LPAR );
        const char* k = expect_name( e );
        expect( e, TK_RPAR );
        return v_bool( map_slot( front, k ) != NULL );
    }
    FAIL( e, "Map has no element of name %s.", key );
    return NULL;
}

static val* list_member( ev* e, val* front, const char* key )   /* container.c:423-518 */
{
    if( !strcmp( key, "push" ) )
    {
        expect( e, TK_LPAR );
        val* v = eval( e, NULL );
        val* c = v_clone( v );
        v_unref( v );
        list_push_owned( front, v_clone( c ) );
        expect( e, TK_RPAR );
        return c;
    }
    if( transform_member( e, front, key ) ) return NULL;
    if( !strcmp( key, "size" ) ) { expect( e, TK_LPAR ); expect( e, TK_RPAR ); return v_int( ( int64_t )front->u.list.n ); }
    if( !strcmp( key, "clear" ) )
    {
        expect( e, TK_LPAR ); expect( e, TK_RPAR );
        for( size_t i = 0; i < front->u.list.n; i++ ) v_unref( front->u.list.d[ i ] );
        front->u.list.n = 0;
        return NULL;
    }
    if( !strcmp( key, "create_inside_composite" ) ) { expect( e, TK_LPAR ); expect( e, TK_RPAR ); return list_composite( e, front, 1 ); }
    if( !strcmp( key, "create_outside_composite" ) ) { expect( e, TK_LPAR ); expect( e, TK_RPAR ); return list_composite( e, front, 0 ); }
    if( !strcmp( key, "create_compound" ) ) { expect( e, TK_LPAR ); expect( e, TK_RPAR ); return list_compound( e, front ); }
    FAIL( e, "arr_s has no element of name %s.", key );
    return NULL;
}

static void member_set_envelope( ev* e, val* front )   /* objects.c:1518-1540, compound.c:420-442 */
{
    expect( e, TK_LPAR );
    val* v = eval( e, NULL );
    if( !( is_object( v ) && acn_obj_type( v->u.obj ) == ACN_SPHERE ) ) FAIL( e, "Object '%s' cannot be used as envelope (use a sphere).", vt( v ) );
    double p[ 3 ];
    acn_obj_get_pos( v->u.obj, p );
    acn_obj_set_envelope( front->u.obj, V3( p[ 0 ], p[ 1 ], p[ 2 ] ), acn_obj_sphere_s_get_radius( v->u.obj ) );
    v_unref( v );
    expect( e, TK_RPAR );
}

static val* obj_member( ev* e, val* front, const char* key )   /* objects.c:1463-1725, compound.c:380-455 */
{
    acn_obj* o = front->u.obj;
    if( transform_member( e, front, key ) ) return NULL;
    if( !strcmp( key, "set_envelope" ) ) { member_set_envelope( e, front ); return NULL; }
    if( !strcmp( key, "set_auto_envelope" ) ) { expect( e, TK_LPAR ); expect( e, TK_RPAR ); auto_envelope( e, o ); return NULL; }
    if( is_compound( front ) )
    {
        if( !strcmp( key, "push" ) )
        {
            expect( e, TK_LPAR );
            val* v = eval( e, NULL );
            if( !v || v->type != V_OBJ ) FAIL( e, "Cannot push '%s' to compound_s.", vt( v ) );
            acn_compound_s_push( o, v->u.obj );
            v_unref( v );
            expect( e, TK_RPAR );
            return NULL;
        }
        FAIL( e, "Compound has no element of name %s.", key );
    }
    static const struct { const char* name; int kind; void ( *fv )( acn_obj*, acn_v3 ); void ( *ff )( acn_obj*, double ); } setters[] =
    {
        { "set_color", 0, acn_obj_set_color, NULL }, { "set_transparency", 0, acn_obj_set_transparency, NULL },
        { "set_refractive_index", 1, NULL, acn_obj_set_refractive_index }, { "set_radiance", 1, NULL, acn_obj_set_radiance },
        { "set_fresnel_reflectivity", 1, NULL, acn_obj_set_fresnel_reflectivity },
        { "set_chromatic_reflectivity", 1, NULL, acn_obj_set_chromatic_reflectivity },
        { "set_diffuse_reflectivity", 1, NULL, acn_obj_set_diffuse_reflectivity }, { "set_sigma", 1, NULL, acn_obj_set_sigma },
        { "set_surface_roughness", 1, NULL, acn_obj_set_surface_roughness }, { NULL, 0, NULL, NULL }
    };
    for( int i = 0; setters[ i ].name; i++ )
    {
        if( strcmp( key, setters[ i ].name ) ) continue;
        expect( e, TK_LPAR );
        if( setters[ i ].kind == 0 ) setters[ i ].fv( o, eval_v3d( e ) ); else setters[ i ].ff( o, eval_f3( e ) );
        expect( e, TK_RPAR );
        return NULL;
    }
    if( !strcmp( key, "set_material" ) )
    {
        expect( e, TK_LPAR );
        val* s = eval( e, NULL );
        if( !s || s->type != V_STR ) FAIL( e, "set_surface: string-argument expected." );
        if( acn_obj_set_material( o, s->u.s ) != ACN_OK ) FAIL( e, "set_surface: Unknown material specification '%s.", s->u.s );
        v_unref( s );
        expect( e, TK_RPAR );
        return NULL;
    }
    double fv;
    if( acn_obj_get_field( o, key, &fv ) )   /* reflected members: sphere.radius, squaroid.a .b .c .r */
    {
        if( try_tok( e, TK_ASSIGN ) )
        {
            val* v = eval_req( e );
            if( !is_num( v ) ) FAIL( e, "Scalar expected." );
            acn_obj_set_field( o, key, to_f3( v ) );
            v_unref( v );
            return v_ref( front );
        }
        return v_float( fv );
    }
    if( !strcmp( key, "set_texture_field" ) )   /* objects.c:1510-1517, interpreter.c:1338-1343 */
    {
        expect( e, TK_LPAR );
        val* v = eval( e, NULL );
        if( !v || v->type != V_BETH || v->u.beth.kind > BETH_TXM_CHESS ) FAIL( e, "Texture map expected." );
        if( v->u.beth.kind == BETH_TXM_PLAIN ) acn_obj_set_texture_field_plain( o, v->u.beth.c1 );
        else acn_obj_set_texture_field_chess( o, v->u.beth.c1, v->u.beth.c2, v->u.beth.scale );
        v_unref( v );
        expect( e, TK_RPAR );
        return NULL;
    }
    if( !strcmp( key, "set_distance_function" ) )   /* objects.c:1691-1710 */
    {
        expect( e, TK_LPAR );
        if( acn_obj_type( o ) != ACN_DISTANCE ) FAIL( e, "Object '%s' must be 'obj_distance_s'.", vt( front ) );
        val* v = eval( e, NULL );
        if( !v || v->type != V_BETH || v->u.beth.kind < BETH_DISTANCE_SPHERE ) FAIL( e, "Object '%s' cannot be used as distance function.", vt( v ) );
        acn_obj_set_distance_function( o, v->u.beth.kind == BETH_DISTANCE_TORUS ? ACN_SDF_TORUS : ACN_SDF_SPHERE, v->u.beth.ex_radius );
        v_unref( v );
        expect( e, TK_RPAR );
        return NULL;
    }
    FAIL( e, "Object has no element of name %s.", key );
    return NULL;
}

static val* eval_member( ev* e, val* front )   /* interpreter.c:1481-1523; front is borrowed */
{
    const char* key = expect_name( e );
    switch( front->type )
    {
        case V_SCENE: return scene_member( e, front, key );
        case V_MAP:   return map_member( e, front, key );
        case V_LIST:  return list_member( e, front, key );
        case V_OBJ:   return obj_member( e, front, key );
        case V_VEC: case V_COLOR: return vec_member( e, front, key );
        case V_BETH:  return beth_member( e, front, key );
        default: break;
    }
    FAIL( e, "Object '%s' has no element named '%s'.", vt( front ), key );
    return NULL;
}

static val* eval_index( ev* e, val* front )   /* interpreter.c:1430-1456 */
{
    if( front->type != V_LIST ) FAIL( e, "Cannot index '%s'.", vt( front ) );
    val* iv = eval( e, NULL );
    expect( e, TK_RBRK );
    if( !is_num( iv ) ) FAIL( e, "Numeric index expected." );
    int64_t index = iv->type == V_FLOAT ? ( int64_t )iv->u.f : iv->type == V_INT ? iv->u.i : iv->u.b;
    v_unref( iv );
    if( index < 0 ) FAIL( e, "Index is negative." );
    if( ( size_t )index >= front->u.list.n )
    {
        if( index > 1000000000 ) FAIL( e, "Attempting to allocate an array of %lld elements seems unintended.", ( long long )index );
        while( front->u.list.n <= ( size_t )index ) list_push_owned( front, NULL );
    }
    if( !front->u.list.d[ index ] && peek( e ) == TK_ASSIGN )
    {
        e->ix++;
        val* v = eval( e, NULL );
        val* c = v_clone( v );
        v_unref( v );
        front->u.list.d[ index ] = c;     /* list may have been reallocated by the evaluation: index again */
    }
    return v_ref( front->u.list.d[ index ] );
}

static val* make_signature( ev* e )   /* interpreter.c:1619-1646 */
{
    val* sig = v_new( V_SIG );
    size_t cap = 0;
    expect( e, TK_LPAR );
    while( !try_tok( e, TK_RPAR ) )
    {
        int type = S_ANY;
        if( peek( e ) == TK_DATA )
        {
            const token* t = &e->blk->t[ e->ix ];
            if( !t->lit || t->lit->type != V_TYPE ) FAIL( e, "Unhandled data element in argument list." );
            type = t->lit->u.mtype;
            e->ix++;
        }
        const char* name = expect_name( e );
        if( sig->u.sig.n == cap ) { cap = cap ? cap * 2 : 8; sig->u.sig.a = xrealloc( sig->u.sig.a, cap * sizeof( sigarg ) ); }
        sig->u.sig.a[ sig->u.sig.n ].name = name;
        sig->u.sig.a[ sig->u.sig.n++ ].type = type;
        if( peek( e ) != TK_RPAR ) expect( e, TK_COMMA );
    }
    return sig;
}

static void print_val( const val* v, int indent )
{
    if( !v ) { printf( "null" ); return; }
    switch( v->type )
    {
        case V_BOOL: printf( "%s", v->u.b ? "true" : "false" ); break;
        case V_INT: printf( "%lld", ( long long )v->u.i ); break;
        case V_FLOAT: printf( "%.17g", v->u.f ); break;
        case V_STR: printf( "%s", v->u.s ); break;
        case V_VEC: case V_COLOR: printf( "<%s> %.17g %.17g %.17g </>", vt( v ), v->u.v.x, v->u.v.y, v->u.v.z ); break;
        case V_MAT: printf( "<m3d> %g %g %g  %g %g %g  %g %g %g </>", v->u.m.x.x, v->u.m.x.y, v->u.m.x.z, v->u.m.y.x, v->u.m.y.y, v->u.m.y.z, v->u.m.z.x, v->u.m.z.y, v->u.m.z.z ); break;
        case V_LIST:
            printf( "<list>\n" );
            for( size_t i = 0; i < v->u.list.n; i++ ) { printf( "%*s", indent + 4, "" ); print_val( v->u.list.d[ i ], indent + 4 ); printf( "\n" ); }
            printf( "%*s</>", indent, "" );
            break;
        case V_MAP:
            printf( "<map>\n" );
            for( size_t i = 0; i < v->u.map.n; i++ ) { printf( "%*s%s: ", indent + 4, "", v->u.map.k[ i ] ); print_val( v->u.map.d[ i ], indent + 4 ); printf( "\n" ); }
            printf( "%*s</>", indent, "" );
            break;
        case V_OBJ: printf( "<object type=%d/>", acn_obj_type( v->u.obj ) ); break;
        case V_BETH: printf( "<%s/>", beth_kinds_g[ v->u.beth.kind ] ); break;
        default: printf( "<%s/>", vt( v ) ); break;
    }
}

/* Evaluates one expression.  `front` (consumed) is the value to the left, if any. */
static val* eval( ev* e, val* front )
{
    int opr = 0;
    if( front )
    {
        int code = peek( e );
        if( code > TK_OP_BEGIN && code < TK_OP_END ) { opr = code; e->ix++; }
        else if( code == TK_LPAR )      /* call; evaluation does not continue behind it (interpreter.c:1423-1429) */
        {
            val* r = eval_call( e, front );
            v_unref( front );
            return r;
        }
        else if( code == TK_LBRK )
        {
            e->ix++;
            val* r = eval_index( e, front );
            v_unref( front );
            return r;
        }
        else return front;

        if( opr > TK_ASG_BEGIN && opr < TK_ASG_END )   /* interpreter.c:1463-1480 */
        {
            val* rhs = eval( e, NULL );
            if( !rhs ) FAIL( e, "Assignment from empty object." );
            switch( opr )
            {
                case TK_ADD_ASG: rhs = op_add( e, v_ref( front ), rhs ); break;
                case TK_SUB_ASG: rhs = op_add( e, v_ref( front ), op_mul( e, v_float( -1 ), rhs ) ); break;
                case TK_MUL_ASG: rhs = op_mul( e, v_ref( front ), rhs ); break;
                case TK_DIV_ASG: rhs = op_mul( e, v_ref( front ), op_inverse( e, rhs ) ); break;
                case TK_MOD_ASG: rhs = op_mod( e, v_ref( front ), rhs ); break;
                default: break;
            }
            v_assign( front, rhs );
            v_unref( rhs );
            return front;
        }
        if( opr == TK_DOT )
        {
            val* r = eval_member( e, front );
            v_unref( front );
            return r;
        }
    }
    else
    {
        int code = peek( e );
        if( code == TK_QUERY || code == TK_DQUERY )
        {
            e->ix++;
            val* v = eval( e, NULL );
            print_val( v, 0 ); printf( "\n" ); fflush( stdout );
            v_unref( v );
            return NULL;
        }
    }

    int unary = 0;
    switch( peek( e ) )
    {
        case TK_ADD: case TK_SUB: case TK_NOT: case TK_ICPS: case TK_OCPS: case TK_CMPD: case TK_ENV: unary = peek( e ); e->ix++; break;
        default: break;
    }

    val* obj = NULL;
    if( peek( e ) == TK_DATA )
    {
        const token* t = &e->blk->t[ e->ix++ ];
        if( t->blk )   /* a block is a closure over the current frame (interpreter.c:1574-1580) */
        {
            obj = v_new( V_CLOSURE );
            obj->u.clo.blk = t->blk; obj->u.clo.lex = e->fr; obj->u.clo.sig = NULL;
        }
        else obj = v_clone( t->lit );
    }
    else if( peek( e ) == TK_NAME )
    {
        const char* key = e->blk->t[ e->ix++ ].name;
        val** slot = frame_get( e->fr, key );
        int pk = peek( e );
        if( pk > TK_ASG_BEGIN && pk < TK_ASG_END )
        {
            if( !slot ) FAIL( e, "'%s' was not defined. Use 'def %s' to define it.", key, key );
            if( !*slot )
            {
                expect( e, TK_ASSIGN );
                val* v = eval( e, NULL );
                val* c = v_clone( v );
                v_unref( v );
                slot = frame_get( e->fr, key );
                if( slot ) { v_unref( *slot ); *slot = c; } else v_unref( c );
            }
            else obj = eval( e, v_ref( *slot ) );
        }
        else if( !slot ) FAIL( e, "Unknown name '%s'", key );
        else obj = v_ref( *slot );
    }
    else if( try_tok( e, TK_DYNARR ) ) obj = v_new( V_LIST );
    else if( try_tok( e, TK_FSIG ) ) obj = make_signature( e );
    else if( try_tok( e, TK_LPAR ) ) { obj = eval( e, NULL ); expect( e, TK_RPAR ); }
    else if( try_tok( e, TK_DEF ) )
    {
        const char* key = expect_name( e );
        if( frame_get_local( e->fr, key ) ) FAIL( e, "'%s' is already defined.", key );
        if( try_tok( e, TK_ASSIGN ) )
        {
            val* v = eval( e, NULL );
            val* c = v_clone( v );
            v_unref( v );
            frame_set( e->fr, key, c );
            obj = v_ref( c );
        }
        else frame_set( e->fr, key, NULL );
    }

    /* postfix operations bind tighter than anything else (interpreter.c:1668-1677) */
    while( obj && ( peek( e ) == TK_LPAR || peek( e ) == TK_LBRK || peek( e ) == TK_DOT ) ) obj = eval( e, obj );

    if( obj )
    {
        switch( unary )
        {
            case TK_SUB:  obj = op_mul( e, v_int( -1 ), obj ); break;
            case TK_NOT:  obj = op_not( e, obj ); break;
            case TK_ICPS: case TK_OCPS:
            {
                if( obj->type != V_LIST ) FAIL( e, "Cannot create %s-composite of '%s'", unary == TK_ICPS ? "inside" : "outside", vt( obj ) );
                val* r = list_composite( e, obj, unary == TK_ICPS ); v_unref( obj ); obj = r;
            }
            break;
            case TK_CMPD:
            {
                if( obj->type != V_LIST ) FAIL( e, "Cannot create compound of '%s'", vt( obj ) );
                val* r = list_compound( e, obj ); v_unref( obj ); obj = r;
            }
            break;
            case TK_ENV:   /* interpreter.c:1172-1200 */
            {
                val* r;
                if( obj->type == V_LIST ) r = list_compound( e, obj );
                else if( obj->type == V_OBJ ) r = v_clone( obj );
                else { FAIL( e, "Cannot compute envelope for of '%s'", vt( obj ) ); r = NULL; }
                v_unref( obj ); obj = r;
                auto_envelope( e, obj->u.obj );
            }
            break;
            default: break;
        }

        if( opr )
        {
            switch( opr )
            {
                /* left to right, continuing with the result as front */
                case TK_MUL: return eval( e, op_mul( e, front, obj ) );
                case TK_DIV: return eval( e, op_mul( e, front, op_inverse( e, obj ) ) );
                case TK_MOD: return eval( e, op_mod( e, front, obj ) );
                case TK_EQ:  return eval( e, v_bool( op_cmp( e, front, obj ) == 0 ) );
                case TK_LT:  return eval( e, v_bool( op_cmp( e, front, obj ) >  0 ) );
                case TK_LE:  return eval( e, v_bool( op_cmp( e, front, obj ) >= 0 ) );
                case TK_GT:  return eval( e, v_bool( op_cmp( e, front, obj ) <  0 ) );
                case TK_GE:  return eval( e, v_bool( op_cmp( e, front, obj ) <= 0 ) );
                /* the right side takes the rest of the expression first */
                case TK_ADD: return op_add( e, front, eval( e, obj ) );
                case TK_SUB: return op_add( e, front, eval( e, op_mul( e, v_int( -1 ), obj ) ) );
                case TK_AND: case TK_OR: case TK_XOR: return op_logic( e, opr, front, eval( e, obj ) );
                case TK_CAT: return eval( e, op_cat( front, obj ) );
                default: FAIL( e, "Invalid operator." );
            }
        }
        else obj = eval( e, obj );   /* operators behind the operand */
    }
    else if( opr ) FAIL( e, "Expression does not yield an operand for the operator." );
    v_unref( front );
    return obj;
}

/* first `else` or `;` behind ix: the jump target the reference's tokenizer stores with if / while / for */
static size_t jump_target( ev* e, size_t ix, int stop_at_else )
{
    for( ; ix < e->blk->n; ix++ )
    {
        int k = e->blk->t[ ix ].kind;
        if( k == TK_SEMI || ( stop_at_else && k == TK_ELSE ) ) return ix;
    }
    FAIL( e, "';' expected." );
    return ix;
}

static int eval_condition( ev* e )
{
    expect( e, TK_LPAR );
    val* c = eval( e, NULL );
    expect( e, TK_RPAR );
    if( !c || c->type != V_BOOL ) FAIL( e, "Expression does not evaluate to boolean." );
    int flag = c->u.b;
    v_unref( c );
    return flag;
}

static val* execute( ev* e )   /* interpreter.c:1734-1850 */
{
    val* ret = NULL;
    while( e->ix < e->blk->n )
    {
        val* obj = NULL;
        int code = peek( e );
        if( code == TK_IF )
        {
            e->ix++;
            size_t target = jump_target( e, e->ix, 1 );
            int flag = eval_condition( e );
            if( flag ) obj = eval( e, NULL ); else e->ix = target;
            if( peek( e ) == TK_ELSE )
            {
                e->ix++;
                if( flag ) e->ix = jump_target( e, e->ix, 0 ); else obj = eval( e, NULL );
            }
        }
        else if( code == TK_WHILE )
        {
            e->ix++;
            size_t end = jump_target( e, e->ix, 0 );
            size_t begin = e->ix;
            for( ;; )
            {
                if( eval_condition( e ) ) { v_unref( obj ); obj = eval( e, NULL ); e->ix = begin; }
                else { e->ix = end; break; }
            }
        }
        else if( code == TK_FOR )
        {
            e->ix++;
            size_t end = jump_target( e, e->ix, 0 );
            frame* ff = xalloc( sizeof( frame ) );   /* kept alive: blocks evaluated in the loop may have captured it */
            ff->ext = e->fr;
            e->fr = ff;
            const char* key = expect_name( e );
            frame_set( ff, key, NULL );
            expect( e, TK_LPAR );
            expect( e, TK_IN );
            val* arr = eval( e, NULL );
            if( !arr || arr->type != V_LIST ) FAIL( e, "Expected: for '%s' in 'list-expression'.", key );
            expect( e, TK_RPAR );
            size_t begin = e->ix;
            for( size_t i = 0; i < arr->u.list.n; i++ )
            {
                if( !arr->u.list.d[ i ] ) continue;
                frame_set( ff, key, v_ref( arr->u.list.d[ i ] ) );   /* the element itself */
                v_unref( eval( e, NULL ) );
                e->ix = begin;
            }
            v_unref( arr );
            e->ix = end;
            e->fr = ff->ext;
            frame_clear( ff );
        }
        else obj = eval( e, NULL );
        expect( e, TK_SEMI );
        v_unref( ret );
        ret = obj;
    }
    return ret;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* entry points                                                                                                  */

static void root_frame( interp* ip, frame* f )   /* interpreter.c:1943-2015 */
{
    for( int i = 0; builtins_g[ i ].name; i++ ) { val* v = v_new( V_BUILTIN ); v->u.builtin = i; frame_set( f, intern( ip, builtins_g[ i ].name ), v ); }
    val* s = v_new( V_SCENE ); s->u.scene = acn_scene_s_create();
    frame_set( f, intern( ip, "scene_s" ), s );
    frame_set( f, intern( ip, "obj_sphere_s" ), v_obj( acn_obj_sphere_s_create( 1.0 ) ) );
    frame_set( f, intern( ip, "obj_plane_s" ), v_obj( acn_obj_plane_s_create() ) );
    frame_set( f, intern( ip, "arr_s" ), v_new( V_LIST ) );
    frame_set( f, intern( ip, "map_s" ), v_new( V_MAP ) );
    val* args = v_new( V_LIST );
    for( int i = 0; i < ip->opts->argc; i++ ) list_push_owned( args, v_str( ip->opts->argv[ i ] ) );
    frame_set( f, intern( ip, "program_args" ), args );
}

static int run( const char* text, size_t size, const char* name, const acn_interp_opts* opts )
{
    static const acn_interp_opts default_opts = { 0 };
    interp* ip = xalloc( sizeof( interp ) );
    ip->opts = opts ? opts : &default_opts;
    ip->start = clock();
    block* top = xalloc( sizeof( block ) );
    frame* root = xalloc( sizeof( frame ) );
    int status = ACN_OK;
    last_error_g[ 0 ] = 0;
    if( setjmp( ip->jb ) == 0 )
    {
        lexer lx = { ip, text, 0, size, add_file( ip, name ), 1 };
        if( lx.n >= 2 && lx.s[ 0 ] == '#' && lx.s[ 1 ] == '!' ) while( lx.p < lx.n && lx.s[ lx.p ] != '\n' ) lx.p++;   /* "#! /path/to/actinon" */
        skip_space( &lx );
        if( match( &lx, "<mclosure_s>" ) )   /* beth's object header of the script files */
        {
            skip_space( &lx );
            if( !match( &lx, "</>" ) ) lex_fail( &lx, "'</>' expected." );
        }
        parse_block( &lx, top );
        if( lx.p < lx.n ) lex_fail( &lx, "Unexpected '}'." );
        root_frame( ip, root );
        top->local.ext = root;
        ev e = { top, 0, &top->local, ip };
        v_unref( execute( &e ) );
    }
    else
    {
        status = ACN_ERR_ARG;
        snprintf( last_error_g, sizeof( last_error_g ), "%s", ip->err );
    }
    /* Blocks, frames and interned names live as long as closures may refer to them; a script runs once per
     * process invocation, so they are released with the process.  Frame contents are dropped here. */
    frame_clear( &top->local );
    frame_clear( root );
    return status;
}

int acn_interpret_string( const char* text, const char* name, const acn_interp_opts* opts )
{
    if( !text ) { snprintf( last_error_g, sizeof( last_error_g ), "no script text" ); return ACN_ERR_ARG; }
    return run( text, strlen( text ), name ? name : "<string>", opts );
}

int acn_interpret_file( const char* path, const acn_interp_opts* opts )
{
    size_t size;
    char* text = path ? read_file( path, &size ) : NULL;
    if( !text ) { snprintf( last_error_g, sizeof( last_error_g ), "Cannot open '%s'.", path ? path : "(null)" ); return ACN_ERR_ARG; }
    int st = run( text, size, path, opts );
    free( text );
    return st;
}

static int capture_scene( void* ctx, acn_scene* scene, const char* file )
{
    acn_scene** out = ctx;
    ( void )file;
    if( !*out ) *out = acn_scene_s_clone( scene );
    return ACN_OK;
}

acn_scene* acn_scene_from_script( const char* path, int auto_env )
{
    acn_scene* out = NULL;
    acn_interp_opts opts = { capture_scene, &out, auto_env, 1, 0, NULL };
    if( acn_interpret_file( path, &opts ) != ACN_OK ) { acn_scene_s_discard( out ); return NULL; }
    if( !out ) snprintf( last_error_g, sizeof( last_error_g ), "%s: script never calls create_image.", path );
    return out;
}
