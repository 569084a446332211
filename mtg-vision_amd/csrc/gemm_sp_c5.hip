// LDS-DMA split GEMM, tile configuration 5: 4 x 2 waves, wave tile 32 x 96, block 128 x 192 - the tile of configuration 1
// on eight waves of half the accumulators each (<= 128 VGPRs), so that two blocks per CU put four waves on every SIMD.
#define SP_CFG_ID 5
#define SP_WM 4
#define SP_WN 2
#define SP_TM 1
#define SP_TN 3
#include "gemm_sp_inst.h"
