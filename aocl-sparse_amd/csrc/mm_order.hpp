// mm_order.hpp -- block order of the csrmm kernels (csrmm_kernels.hip, csrmm_window_kernels.hip, csrmm_bell_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "internal.hpp"

namespace mi355
{
// Block index of a launch.  `word` = XCD chunk (0: launch order; > 0: every XCD a contiguous eighth of the blocks) | MM_DESCENDING:
// the blocks in descending order.  Every second csrmm product of a handle asks for descending (SpmvPlan::mm_products, set by
// csrmm_api.cpp through mm_direction_word): the end of one product's sweep over B and C is still in the 256 MB Infinity Cache
// when the next one starts there.  Each block's work does not depend on when it runs: same bits.  The launchers pass the bit
// (mmw) to the ROW-MAJOR kernels only -- the 32-column slab 0.179 -> 0.165 ms (C read), 0.136 -> 0.128 (overwritten), 256 columns
// overwritten 0.86 -> 0.84; the column-window kernel measured the same both ways and the blocked-ELL MFMA kernels 5 % SLOWER
// descending (1.09 -> 1.14 ms), so those always run ascending (profiles/r5/sell_placement.txt).
__device__ __forceinline__ unsigned mm_linear_index(int word)
{
    return (word & MM_DESCENDING) ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
}
__device__ __forceinline__ int mm_block_index(int word)
{
    const int      chunk = word & ~(MM_DESCENDING | MM_DEAL);
    const unsigned bi    = mm_linear_index(word);
    if(word & MM_DEAL) // chunks of `chunk` consecutive blocks dealt to the XCDs in turn (gridDim.x is a multiple of 8 * chunk)
    {
        const unsigned j = bi >> 3;
        return (int)(((j / chunk) * 8u + (bi & 7u)) * chunk + j % chunk);
    }
    return chunk > 0 ? (int)(bi & 7) * chunk + (int)(bi >> 3) : (int)bi;
}

// host side: the chunk word of a launch = its XCD chunk | the calling thread's direction for this product
inline int mmw(int chunk)
{
    return chunk | mm_direction_word();
}

} // namespace mi355
