/* acn_unimachine.h -- the two CSG machines in LOCK-STEP form (included by acn_device.h).
 *
 * obj_ray_hit / obj_side (objects.c:261-284, 366-370) of a CSG tree whose ROOT is the same in every lane of the wave --
 * which is how the root-compound loops call them (element_hit).  The per-lane machines above let every lane keep its
 * own ( node, pc ): lanes drift apart after the first data-dependent branch, and from then on the wave executes, step
 * by step, the union of what its lanes are doing -- plane here, squaroid there, frame logic elsewhere (31 % VALU lane
 * utilisation in k_walk, profiles/r02/NOTES.md) -- with every node field read per lane from LDS.
 *
 * Here the wave walks the tree ONCE, in the reference's order, with ONE ( node, phase ) for all lanes; what differs per
 * lane is only whether the lane takes part in an evaluation (`act`) and the data it carries.  A lane's own sequence of
 * arithmetic is exactly what the reference's recursion does for that ray, so results are bit-identical; a lane that
 * needs no evaluation of a subtree (envelope missed, pair already decided) sits it out.  Node records are read with
 * wave-uniform addresses (scalar loads from the constant address space), type dispatch is a scalar branch, and the
 * control state lives in SGPRs.
 *
 * Frame of a pair (the reference's locals across its child calls), per lane: one word + `a` + the parked normal --
 * the same 12 (+24) bytes as in the per-lane machine, in the same LDS planes -- with the word now holding
 *   uniform   node << 8 | inherit << 7 | phase << 4
 *   per lane  swapped << 3 | mode << 1 | in        mode: 0 undecided, 1 in the alternating walk, 2 decided
 * A decided lane keeps its result in ( a, parked normal ) until the frame completes for the wave. */
#ifndef ACN_UNIMACHINE_H
#define ACN_UNIMACHINE_H

#ifndef ACN_UNI_MACHINE
#define ACN_UNI_MACHINE 1
#endif
/* which pairs the lock-step machines evaluate in line (pair_hit / pair_side above) instead of through a frame:
 * 0 none, 1 leaf pairs, 2 leaf pairs and level-2 pairs */
#ifndef ACN_UNI_PAIR_LEVEL
#define ACN_UNI_PAIR_LEVEL 2
#endif

#define UF_IN         0x1u
#define UF_MODE( w )  ( ( ( w ) >> 1 ) & 3u )
#define UF_WALKING    ( 1u << 1 )
#define UF_DECIDED    ( 2u << 1 )
#define UF_SWAP       0x8u
#define UF_LANE_BITS  0xFu
#define UF_PHASE( w ) ( ( ( w ) >> 4 ) & 7u )
#define UF_INHERIT    0x80u
#define UF_NODE( w )  ( ( int )( ( w ) >> 8 ) )
#define UF_PACK( node, phase, inherit ) ( ( ( uint32_t )( node ) << 8 ) | ( ( inherit ) ? UF_INHERIT : 0u ) | ( ( uint32_t )( phase ) << 4 ) )
#define UF_SET_PHASE( w, phase ) ( ( ( w ) & ~( 7u << 4 ) ) | ( ( uint32_t )( phase ) << 4 ) )

DEV bool wave_any( bool x ) { return __ballot( x ) != 0ull; }
DEV uint32_t wave_uniform( uint32_t x ) { return ( uint32_t )__builtin_amdgcn_readfirstlane( ( int )x ); }

/* the global-memory view of the scene for the in-line pair evaluators (their node pointer is then wave-uniform too) */
template< class SR > DEV SceneRefT< NodeP > uni_view( const SR& sc )
{
    SceneRefT< NodeP > g;
    g.nodes = sc.gnodes; g.gnodes = sc.gnodes; g.elems = sc.elems; g.flags = sc.flags; g.n_elems = sc.n_elems; g.lds_stack = sc.lds_stack;
    return g;
}

/* obj_side for the lanes with `act`, the others get 0.  Side frame word: node << 4 | phase << 2 | early << 1 | in
 * (early: the lane's first child already decided the pair, objects.c:1096-1099 / 1253-1256). */
template< class SR, class CT >
DEV_SIDE int obj_side_uni( SR sc, int root, bool act, V3 pos, CT* cnt )
{
    const SceneRefT< NodeP > g = uni_view( sc );
    uint32_t st[ ACN_CSG_MAX_DEPTH ];
    V3 aux[ ACN_CSG_MAX_DEPTH ];
    const bool lds = sc.lds_stack != ACN_NO_LDS_STACK;
    LdsU32P ls = ( LdsU32P )( ( char ACN_LDS* )acn_lds_raw + sc.lds_stack ) + ACN_LDS_DEPTH * ACN_LDS_LANES * 9 + threadIdx.x;
    uint32_t cur = 0;
    int depth = 0, na = 0;
    int node = root;
    int r = 0;
    ACN_LAP( PH_M_FRAME );
    for( ;; )
    {
        /* EVAL( node ) for the lanes with act */
        ACN_NODE_UNIFORM( n, &g.nodes[ node ] )
        const int type = n->type;
        const uint32_t nflags = n->flags;
        bool have = true;
        bool in = act;
        if( act ) { cnt->inc( CNT_SIDE ); r = 1; }
        if( nflags & ACN_NODE_HAS_ENVELOPE ) { if( act ) in = env_side( n, pos ) != 1; }
        if( !wave_any( in ) )
        {
            /* every lane is outside the envelope: r = 1 */
        }
        else if( type <= ACN_DISTANCE )
        {
            if( in )
            {
                if( type == ACN_PLANE )         { cnt->cost( ACN_F_SIDE_PLANE ); r = v_sub_mlv( pos, ld3( n->pos ), ld3( n->rax + 6 ) ) > 0 ? 1 : -1; }   /* gmath.h:52-55 */
                else if( type == ACN_SPHERE )   r = sphere_observer_side( ld3( n->pos ), n->prm[ 0 ], pos );
                else if( type == ACN_SQUAROID ) r = squaroid_side( n, pos );
                else                            { r = distance_side( n, pos ); cnt->inc( CNT_SDF_EVAL ); }
            }
        }
#if ACN_UNI_PAIR_LEVEL >= 2
        else if( nflags & ( ACN_GFLAG_LEAF_PAIR | ACN_GFLAG_PAIR2 ) )
        {
            if( in ) r = pair_side< 2 >( g, n, pos, cnt );   /* (leaf pairs too: one expansion of the pair code) */
        }
#elif ACN_UNI_PAIR_LEVEL >= 1
        else if( nflags & ACN_GFLAG_LEAF_PAIR )
        {
            if( in ) r = pair_side< 1 >( g, n, pos, cnt );
        }
#endif
        else if( depth >= ACN_CSG_MAX_DEPTH )
        {
            atomicOr( sc.flags, ACN_FLAG_STACK_OVERFLOW );
        }
        else
        {
            if( depth > 0 )
            {
                if( lds && depth <= ACN_LDS_DEPTH ) ls[ ( depth - 1 ) * ACN_LDS_LANES ] = cur; else st[ depth - 1 ] = cur;
            }
            depth++;
            cur = ( ( uint32_t )node << 4 ) | ( 1u << 2 ) | ( in ? 1u : 0u );
            if( type == ACN_SCALE )   /* objects.c:1439-1443 */
            {
                if( in ) cnt->cost( ACN_F_SIDE_SCALE );
                aux[ na++ ] = pos;    /* na <= depth <= ACN_CSG_MAX_DEPTH */
                M3 rax = node_rax( n );
                V3 p = m_mlv( rax, v_sub( pos, ld3( n->pos ) ) );
                pos = v_mld( p, mk( n->prm[ 0 ], n->prm[ 1 ], n->prm[ 2 ] ) );
            }
            node = n->child0;
            act = in;
            have = false;
        }
        /* RETURN( r ) into the enclosing composites */
        while( have )
        {
            if( depth == 0 ) { ACN_LAP( PH_M_SIDE ); return r; }
            const uint32_t uw = wave_uniform( cur );
            const int cnode = ( int )( uw >> 4 );
            const uint32_t phase = ( uw >> 2 ) & 3u;
            NodeP fn = &g.nodes[ cnode ];
            const int ftype = fn->type;
            const bool fin = ( cur & 1u ) != 0;
            bool done = true;
            if( ftype == ACN_NEG ) { if( fin ) r = -r; }                  /* objects.c:1341-1344 */
            else if( ftype == ACN_SCALE ) pos = aux[ --na ];
            else
            {
                const int want = ( ftype == ACN_PAIR_INSIDE ) ? -1 : 1;     /* objects.c:1096-1099, 1253-1256 */
                if( phase == 1u )
                {
                    const bool go = fin && r == want;
                    if( fin && !go ) cur |= 2u;
                    if( wave_any( go ) )
                    {
                        cur = ( cur & ~( 3u << 2 ) ) | ( 2u << 2 );
                        node = fn->child1;
                        act = go;
                        done = false; have = false;
                    }
                    else if( fin ) r = -want;
                }
                else if( fin )
                {
                    r = ( cur & 2u ) ? -want : ( ( r == want ) ? want : -want );
                }
            }
            if( done )
            {
                depth--;
                if( depth > 0 ) cur = ( lds && depth <= ACN_LDS_DEPTH ) ? ls[ ( depth - 1 ) * ACN_LDS_LANES ] : st[ depth - 1 ];
            }
        }
    }
}

/* obj_ray_hit of the tree under `root` for every lane that calls (same root in all of them).
 * PARK: the origin of the ray the machine is evaluating lives in the lane's LDS slot (OrgLds, acn_device.h) instead of a register
 * that the allocator would spill: written where the reference's recursion passes another origin down (scale wrappers, the walk
 * steps of a pair) or returns to the frame's own, read where a leaf, an envelope or an in-line pair uses it. */
template< bool NOR, bool PARK = false, class SR, class CT >
DEV_HIT double obj_ray_hit_uni( SR sc, int root, V3 rp_in, V3 rd, V3* out_nor, CT* cnt )
{
    const SceneRefT< NodeP > g = uni_view( sc );
    OrgLds org;
    org.p = nullptr;
    if constexpr( PARK ) org.p = ( volatile double ACN_LDS* )( ( char ACN_LDS* )acn_lds_raw + sc.lds_stack + ACN_LDS_STACK_BYTES + ACN_LDS_POOL_BYTES ) + threadIdx.x;
    V3 rp = rp_in;
    auto RP = [ & ]() -> V3 { if constexpr( PARK ) return org.get(); else return rp; };
    auto SET_RP = [ & ]( V3 v ) { if constexpr( PARK ) org.set( v ); else rp = v; };
    if constexpr( PARK ) org.set( rp_in );
    uint32_t st_w[ ACN_CSG_MAX_DEPTH ];
    double   st_a[ ACN_CSG_MAX_DEPTH ];
    V3       st_n[ ACN_CSG_MAX_DEPTH ];     /* touched only when NOR */
    V3       aux[ 2 * ACN_CSG_MAX_DEPTH ];  /* parked origins / directions */
    const bool lds = sc.lds_stack != ACN_NO_LDS_STACK;
    LdsF64P la = ( LdsF64P )( ( char ACN_LDS* )acn_lds_raw + sc.lds_stack ) + threadIdx.x;
    LdsF64P ln = la + ACN_LDS_DEPTH * ACN_LDS_LANES;   /* x, y, z planes of the parked normals */
    LdsU32P lw = ( LdsU32P )( ( char ACN_LDS* )acn_lds_raw + sc.lds_stack ) + ACN_LDS_DEPTH * ACN_LDS_LANES * 8 + threadIdx.x;
    uint32_t cur_w = 0;
    double cur_a = 0;                       /* pair: a1 (phase 2), walk offset (walking), result (decided) | scale: d_factor */
    V3 cur_n1 = mk( 0, 0, 0 );
    V3 cur_rp = rp_in;                      /* origin of the ray the current frame received */
    bool derived = false;                   /* wave-uniform: the rp of the next EVAL differs from cur_rp (walk steps, scale wrappers) */
    int depth = 0, na = 0;
    int node = root;
    bool act = true;
    double ret_a = F3_INF;
    V3 ret_n = mk( 0, 0, 0 );
    ACN_TALLY( 12, true );
    for( ;; )
    {
        /* ---- EVAL( node, rp, rd ) for the lanes with act ---- */
        ACN_LAP( PH_M_FRAME );
        ACN_NODE_UNIFORM( n, &g.nodes[ node ] )
        const int type = n->type;
        const uint32_t nflags = n->flags;
        bool have = true;
        bool in = act;
        if( act ) { cnt->inc( CNT_OBJ_HIT ); ret_a = F3_INF; }
        if( nflags & ACN_NODE_HAS_ENVELOPE ) { if( act ) in = env_ray_hits( n, RP(), rd ); }
        if( !wave_any( in ) )
        {
            /* no lane gets inside the envelope: f3_inf for all */
        }
        else if( type <= ACN_DISTANCE )
        {
            ACN_TALLY( 14, in );
            if( in )
            {
                if( type == ACN_PLANE )         ret_a = plane_ray_hit( ld3( n->pos ), ld3( n->rax + 6 ), RP(), rd, NOR, &ret_n );
                else if( type == ACN_SPHERE )   ret_a = sphere_ray_hit( ld3( n->pos ), n->prm[ 0 ], RP(), rd, NOR, &ret_n );
                else if( type == ACN_SQUAROID ) ret_a = squaroid_ray_hit( n, RP(), rd, NOR, &ret_n );
                else                            { V3 dn = mk( 0, 0, 0 ); ret_a = distance_ray_hit( n, RP(), rd, NOR, &dn, cnt ); if( NOR && ret_a < F3_INF ) ret_n = dn; }   /* (a real call: only dn's address escapes) */
                if( NOR && ret_a < F3_INF && n->surface_roughness > 0 ) ret_n = roughness_normal( n, ret_n, ray_pos( RP(), rd, ret_a ) );
            }
            ACN_LAP( PH_M_LEAF );
        }
#if ACN_UNI_PAIR_LEVEL >= 1
        else if( nflags & ( ACN_UNI_PAIR_LEVEL >= 2 ? ( ACN_GFLAG_LEAF_PAIR | ACN_GFLAG_PAIR2 ) : ACN_GFLAG_LEAF_PAIR ) )
        {
            ACN_TALLY( 14, in );
            if( in )
            {
#if ACN_UNI_PAIR_LEVEL >= 2
                /* (a leaf pair through the level-2 code as well: its operands are simple, so every step is the level-1 step,
                 * and the kernel carries one expansion of the pair code instead of two) */
                if constexpr( PARK ) ret_a = pair_hit< 2, true >( g, n, org, rd, NOR, &ret_n, cnt );
                else                 ret_a = pair_hit< 2, true >( g, n, rp, rd, NOR, &ret_n, cnt );
#else
                if constexpr( PARK ) ret_a = pair_hit< 1, true >( g, n, org, rd, NOR, &ret_n, cnt );
                else                 ret_a = pair_hit< 1, true >( g, n, rp, rd, NOR, &ret_n, cnt );
#endif
                if( NOR && ret_a < F3_INF && n->surface_roughness > 0 ) ret_n = roughness_normal( n, ret_n, ray_pos( RP(), rd, ret_a ) );
            }
            ACN_LAP( PH_M_PAIR );
        }
#endif
        else if( depth >= ACN_CSG_MAX_DEPTH )
        {
            atomicOr( sc.flags, ACN_FLAG_STACK_OVERFLOW );
        }
        else
        {
            if( depth > 0 )
            {
                if( lds && depth <= ACN_LDS_DEPTH )
                {
                    const int o = ( depth - 1 ) * ACN_LDS_LANES;
                    lw[ o ] = cur_w; la[ o ] = cur_a;
                    if( NOR ) { ln[ o ] = cur_n1.x; ln[ o + ACN_LDS_DEPTH * ACN_LDS_LANES ] = cur_n1.y; ln[ o + 2 * ACN_LDS_DEPTH * ACN_LDS_LANES ] = cur_n1.z; }
                }
                else
                {
                    st_w[ depth - 1 ] = cur_w; st_a[ depth - 1 ] = cur_a;
                    if( NOR ) st_n[ depth - 1 ] = cur_n1;
                }
            }
            depth++;
            if( derived ) aux[ na++ ] = cur_rp;      /* na <= 2 * depth */
            cur_w = UF_PACK( node, 1, !derived ) | ( in ? UF_IN : 0u );
            cur_rp = RP();
            derived = false;
            if( type == ACN_SCALE )   /* objects.c:1418-1428 */
            {
                if( in ) cnt->cost( ACN_F_SCALE_WRAP );
                M3 rax = node_rax( n );
                V3 inv_scale = mk( n->prm[ 0 ], n->prm[ 1 ], n->prm[ 2 ] );
                V3 p2 = v_mld( m_mlv( rax, v_sub( cur_rp, ld3( n->pos ) ) ), inv_scale );
                V3 d2 = v_mld( m_mlv( rax, rd ), inv_scale );
                double d_length = acn_sqrt( v_sqr( d2 ) );
                double d_factor = ( d_length > 0 ) ? ( 1.0 / d_length ) : 0;
                d2 = v_mlf( d2, d_factor );
                aux[ na++ ] = rd; cur_a = d_factor;
                SET_RP( p2 ); rd = d2;
                derived = true;
            }
            node = n->child0;
            act = in;
            have = false;
        }

        /* ---- RETURN( ret_a, ret_n ) into the enclosing composites ---- */
        while( have )
        {
            if( depth == 0 )
            {
                /* like the reference, the caller's normal is only written on a hit (obj_ray_exit relies on it) */
                if( NOR && ret_a < F3_INF ) *out_nor = ret_n;
                ACN_LAP( PH_M_FRAME );
                return ret_a;
            }
            const uint32_t uw = wave_uniform( cur_w );
            const int cnode = UF_NODE( uw );
            const uint32_t phase = UF_PHASE( uw );
            NodeP fn = &g.nodes[ cnode ];
            const int ftype = fn->type;
            const bool fin = ( cur_w & UF_IN ) != 0;
            bool done = true;
            if( ftype == ACN_NEG )   /* objects.c:1329-1339 */
            {
                if( NOR && fin && ret_a < F3_INF ) ret_n = v_neg( ret_n );
            }
            else if( ftype == ACN_SCALE )   /* objects.c:1430-1437 */
            {
                rd = aux[ --na ];
                if( fin )
                {
                    double a1 = ret_a + F3_EPS;
                    if( a1 < F3_INF )
                    {
                        if( NOR )
                        {
                            V3 n1 = v_mld( ret_n, mk( fn->prm[ 0 ], fn->prm[ 1 ], fn->prm[ 2 ] ) );
                            ret_n = v_of_length( m_tmlv( node_rax( fn ), n1 ), 1.0 );
                        }
                        ret_a = a1 * cur_a - F3_EPS;
                    }
                    else
                    {
                        ret_a = F3_INF;
                    }
                }
            }
            else   /* pair: objects.c:1052-1094 / 1209-1251 */
            {
                const int want = ( ftype == ACN_PAIR_INSIDE ) ? -1 : 1;
                const int c0 = fn->child0, c1 = fn->child1;
                bool resolve = false;
                if( phase == 1u )
                {
                    if( fin ) { cur_a = ret_a; cur_n1 = ret_n; }
                    cur_w = UF_SET_PHASE( cur_w, 2 );
                    node = c1; SET_RP( cur_rp ); derived = false; act = fin;
                    done = false; have = false;
                }
                else if( phase == 3u && wave_any( fin && UF_MODE( cur_w ) == 1u && ( cur_w & UF_SWAP ) ) )
                {
                    /* the walking lanes whose turn was child 0 have their hit; now those whose turn is child 1 */
                    cur_w = UF_SET_PHASE( cur_w, 4 );
                    node = c1; SET_RP( ray_pos( cur_rp, rd, cur_a ) ); derived = true; act = fin && UF_MODE( cur_w ) == 1u && ( cur_w & UF_SWAP );
                    done = false; have = false;
                }
                else
                {
                    /* Both children are evaluated (phase 2), or a round of the alternating walk is (phases 3 / 4).  Either
                     * way up to two side tests follow -- child 1, then child 0 -- through ONE in-line copy of the side machine:
                     *   phase 2   side( child 1, p( a1 ) ) for the lanes with a1 < a2; then side( child 0, p( a2 ) ) for the
                     *             lanes the first test did not decide and whose a2 is finite;
                     *   walk      side of the OTHER operand at the new hit: child 1 for the lanes that evaluated child 0 ... */
                    const bool p2 = phase == 2u;
                    const bool walking = !p2 && fin && UF_MODE( cur_w ) == 1u;
                    const double a = ret_a;                             /* phase 2: a2 | walk: the step */
                    if( p2 ? fin : walking ) cnt->cost( p2 ? 2 * ACN_F_PAIR_STEP : ACN_F_PAIR_STEP );
                    const bool wq = walking && a < F3_INF;
                    V3 pos0, pos1;
                    bool m0, m1;
                    if( p2 )
                    {
                        pos0 = ray_pos( cur_rp, rd, cur_a ); pos1 = ray_pos( cur_rp, rd, a );
                        m0 = fin && cur_a < a; m1 = false;
                    }
                    else
                    {
                        pos0 = pos1 = ray_pos( ray_pos( cur_rp, rd, cur_a ), rd, a );
                        m0 = wq && !( cur_w & UF_SWAP ); m1 = wq && ( cur_w & UF_SWAP );
                    }
                    int s0 = 0, s1 = 0;
                    for( int k = 0; k < 2; k++ )
                    {
                        if( k == 1 && p2 ) m1 = fin && !( m0 && s0 == want ) && a < F3_INF;
                        const bool m = k ? m1 : m0;
                        if( wave_any( m ) )
                        {
                            int s = obj_side_uni( sc, k ? c0 : c1, m, k ? pos1 : pos0, cnt );
                            if( k ) s1 = s; else s0 = s;
                        }
                    }
                    if( p2 )
                    {
                        /* a1 = cur_a, n1 = cur_n1, a2 = a, n2 = ret_n */
                        const bool dec1 = m0 && s0 == want;             /* ( a1, n1 ): already in place */
                        const bool miss = fin && !dec1 && a >= F3_INF;
                        if( miss ) cur_a = F3_INF;
                        const bool dec2 = m1 && s1 == want;
                        if( dec2 ) cur_n1 = ret_n;
                        if( m1 ) cur_a = a;                             /* the result ( a2, n2 ), or the walk offset */
                        const bool walk = m1 && !dec2;
                        cur_w = ( cur_w & ~( UF_SWAP | ( 3u << 1 ) ) ) | ( walk ? UF_WALKING : UF_DECIDED );
                    }
                    else
                    {
                        const bool whit = wq && ( ( cur_w & UF_SWAP ) ? s1 : s0 ) == want;
                        if( whit ) { cur_a = cur_a + a; cur_n1 = ret_n; }
                        bool cont = wq && !whit;
                        if( cont )
                        {
                            cur_a += a + 2 * F3_EPS;
                            if( !( cur_a < F3_INF ) ) cont = false;
                        }
                        if( walking && !whit && !cont ) cur_a = F3_INF;
                        if( walking )
                        {
                            if( cont ) cur_w ^= UF_SWAP;
                            else cur_w = ( cur_w & ~( 3u << 1 ) ) | UF_DECIDED;
                        }
                    }
                    resolve = true;
                }
                if( resolve )
                {
                    /* next round of the alternating walk for the lanes still in it, else the frame is complete */
                    const bool walking = fin && UF_MODE( cur_w ) == 1u;
                    const bool w0 = walking && !( cur_w & UF_SWAP );
                    if( wave_any( walking ) )
                    {
                        const bool first0 = wave_any( w0 );
                        cur_w = UF_SET_PHASE( cur_w, first0 ? 3 : 4 );
                        node = first0 ? c0 : c1;
                        SET_RP( ray_pos( cur_rp, rd, cur_a ) ); derived = true;
                        act = first0 ? w0 : walking;
                        done = false; have = false;
                    }
                    else if( fin ) { ret_a = cur_a; ret_n = cur_n1; }
                }
            }
            if( done )
            {
                /* POST of the composite itself (objects.c:266-282), then hand its result to its parent */
                SET_RP( cur_rp );
                if( NOR && fn->surface_roughness > 0 ) { if( fin && ret_a < F3_INF ) ret_n = roughness_normal( fn, ret_n, ray_pos( cur_rp, rd, ret_a ) ); }
                if( !( uw & UF_INHERIT ) ) cur_rp = aux[ --na ];
                derived = false;
                depth--;
                if( depth > 0 )
                {
                    if( lds && depth <= ACN_LDS_DEPTH )
                    {
                        const int o = ( depth - 1 ) * ACN_LDS_LANES;
                        cur_w = lw[ o ]; cur_a = la[ o ];
                        if( NOR ) cur_n1 = mk( ln[ o ], ln[ o + ACN_LDS_DEPTH * ACN_LDS_LANES ], ln[ o + 2 * ACN_LDS_DEPTH * ACN_LDS_LANES ] );
                    }
                    else
                    {
                        cur_w = st_w[ depth - 1 ]; cur_a = st_a[ depth - 1 ];
                        if( NOR ) cur_n1 = st_n[ depth - 1 ];
                    }
                }
            }
        }
    }
}

#endif /* ACN_UNIMACHINE_H */
