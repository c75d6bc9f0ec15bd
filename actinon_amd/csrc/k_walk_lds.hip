/* k_walk< false, true, * > (node array staged in LDS) and the launch dispatcher of k_walk: see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"

void acn_launch_walk_glb( KernelFlags f, uint32_t pass, bool last, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                          const double* pos_xy, size_t first_pixel, uint32_t base, uint32_t n_cam, TileOrder order,
                          unsigned long long* accum, unsigned long long* counters );     /* k_walk_glb.hip */
void acn_launch_walk_count( KernelFlags f, uint32_t pass, bool last, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                            const double* pos_xy, size_t first_pixel, uint32_t base, uint32_t n_cam, TileOrder order,
                            unsigned long long* accum, unsigned long long* counters );   /* k_walk_count.hip */

void acn_launch_walk( KernelFlags f, uint32_t pass, bool last, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                      const double* pos_xy, size_t first_pixel, uint32_t base, uint32_t n_cam, TileOrder order,
                      unsigned long long* accum, unsigned long long* counters )
{
    if( f.count )           acn_launch_walk_count( f, pass, last, q, lds_bytes, stream, s, pos_xy, first_pixel, base, n_cam, order, accum, counters );
    else if( !f.lds_nodes ) acn_launch_walk_glb( f, pass, last, q, lds_bytes, stream, s, pos_xy, first_pixel, base, n_cam, order, accum, counters );
    else if( f.prune )      ACN_LW_( false, true, true );
    else                    ACN_LW_( false, true, false );
}
