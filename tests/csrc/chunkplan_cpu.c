/* test shim: actinon_amd/csrc/acn_chunkplan.h (the chunk controller of launch_render) behind plain exported functions */
#include "acn_chunkplan.h"

void     plan_init( acn_chunk_ctl* c, double fill_target ) { acn_ctl_init( c ); c->fill_target = fill_target; }
uint32_t plan_next( const acn_chunk_ctl* c, size_t remaining, size_t chunk, int fixed, int rates_known, const double* rate, const uint32_t* cap )
{
    return acn_ctl_next( c, remaining, chunk, fixed, rates_known, rate, cap );
}
size_t   plan_overflow( acn_chunk_ctl* c, uint32_t cnt ) { return acn_ctl_overflow( c, cnt ); }
void     plan_fit( acn_chunk_ctl* c ) { acn_ctl_fit( c ); }
