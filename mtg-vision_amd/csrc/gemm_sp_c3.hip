// LDS-DMA split GEMM, tile configuration 3: 4 x 1 waves, wave tile 32 x 64, block 128 x 64.
#define SP_CFG_ID 3
#define SP_WM 4
#define SP_WN 1
#define SP_TM 1
#define SP_TN 2
#include "gemm_sp_inst.h"
