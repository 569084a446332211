// LDS-DMA split GEMM, tile configuration 2: 4 x 1 waves, wave tile 32 x 96, block 128 x 96.
#define SP_CFG_ID 2
#define SP_WM 4
#define SP_WN 1
#define SP_TM 1
#define SP_TN 3
#include "gemm_sp_inst.h"
