/* acn_chunkplan.h -- the chunk controller of launch_render (actinon_hip.hip) as plain host C, so that it can be unit-tested
 * without a GPU (tests/csrc/chunkplan_cpu.c, tests/test_chunkplan.py).
 *
 * A call is worked off in chunks of sample positions.  The size of a chunk is planned from the learned demand of every work
 * queue per position (`rate`) so that the fullest queue reaches `fill_target` of its capacity; a chunk whose records do not
 * fit anyway (a queue overflowed) is thrown away and redone smaller.  Rendering is deterministic: the same positions
 * overflow the same queues again, so the one property the controller MUST have is that a retry is strictly smaller than the
 * chunk that overflowed -- whatever fill_target, the rates and the "take all that is left" rule say (round 3's controller
 * could replay the same chunk for ever once fill_target had decayed below 0.425: ADVICE r03).  The rules:
 *   next     min( remaining, planned chunk ); if no retry is pending and the WHOLE rest is predicted to fill no queue beyond
 *            85 % ( rate * remaining <= 0.85 * capacity for every queue ), the rest is taken in one chunk -- a second chunk
 *            would be another whole chain of launches for a few positions; while a retry is pending the chunk is at most
 *            half of the one that overflowed;
 *   overflow fill_target decays by 0.85 (floor 0.3), the retry bound halves;
 *   fit      the retry bound is lifted; every fourth fitting chunk in a row raises fill_target by 5 % (ceiling 0.7). */
#ifndef ACN_CHUNKPLAN_H
#define ACN_CHUNKPLAN_H

#include <stddef.h>
#include <stdint.h>

#define ACN_PLAN_QUEUES 5

typedef struct
{
    double   fill_target;      /* persists with the handle (and is inherited by its lanes) */
    uint32_t fits_in_a_row;
    uint32_t retry_bound;      /* 0: no retry pending; else the next chunk may have at most this many positions */
} acn_chunk_ctl;

static inline void acn_ctl_init( acn_chunk_ctl* c ) { c->fill_target = 0.7; c->fits_in_a_row = 0; c->retry_bound = 0; }

/* positions of the next chunk.  remaining > 0; chunk: the planned size; fixed: the caller pinned the chunk size (ACN_CHUNK):
 * no take-all; rates_known / rate / cap: learned records per position and capacity of every queue */
static inline uint32_t acn_ctl_next( const acn_chunk_ctl* c, size_t remaining, size_t chunk, int fixed, int rates_known,
                                     const double* rate, const uint32_t* cap )
{
    size_t cnt = remaining < chunk ? remaining : chunk;
    if( !fixed && c->retry_bound == 0 )
    {
        int all = 1;
        if( rates_known )
        {
            for( int q = 0; q < ACN_PLAN_QUEUES; q++ ) if( rate[ q ] * ( double )remaining > 0.85 * ( double )cap[ q ] ) all = 0;
        }
        else all = ( double )remaining <= 1.2 * ( double )chunk;   /* nothing learned yet: the first guess errs on the safe side by factors */
        if( all ) cnt = remaining;
    }
    if( c->retry_bound != 0 && cnt > c->retry_bound ) cnt = c->retry_bound;
    if( cnt < 1 ) cnt = 1;
    return ( uint32_t )cnt;
}

/* a chunk of cnt positions overflowed; returns the planned size of the retry (the caller fails the call when cnt <= 1) */
static inline size_t acn_ctl_overflow( acn_chunk_ctl* c, uint32_t cnt )
{
    c->fill_target = c->fill_target * 0.85 < 0.3 ? 0.3 : c->fill_target * 0.85;
    c->fits_in_a_row = 0;
    c->retry_bound = cnt / 2 ? cnt / 2 : 1;
    return c->retry_bound;
}

static inline void acn_ctl_fit( acn_chunk_ctl* c )
{
    c->retry_bound = 0;
    if( ++c->fits_in_a_row >= 4 ) { c->fits_in_a_row = 0; c->fill_target = c->fill_target * 1.05 > 0.7 ? 0.7 : c->fill_target * 1.05; }
}

#endif /* ACN_CHUNKPLAN_H */
