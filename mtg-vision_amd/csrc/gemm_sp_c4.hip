// LDS-DMA split GEMM, tile configuration 4: 4 x 1 waves, wave tile 32 x 32, block 128 x 32.
#define SP_CFG_ID 4
#define SP_WM 4
#define SP_WN 1
#define SP_TM 1
#define SP_TN 1
#include "gemm_sp_inst.h"
