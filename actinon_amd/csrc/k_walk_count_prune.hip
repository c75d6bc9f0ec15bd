/* k_walk< true, *, true > (instrumented, with prune programs / in-line simple compounds): see acn_launch.h */
#include <hip/hip_runtime.h>
#include "acn_launch.h"

void acn_launch_walk_count_prune( KernelFlags f, uint32_t pass, bool last, const LevelQ& q, size_t lds_bytes, hipStream_t stream, const SceneArgs& s,
                                  const double* pos_xy, size_t first_pixel, uint32_t base, uint32_t n_cam, TileOrder order,
                                  unsigned long long* accum, unsigned long long* counters )
{
    if( f.lds_nodes ) ACN_LW_( true, true, true );
    else              ACN_LW_( true, false, true );
}
