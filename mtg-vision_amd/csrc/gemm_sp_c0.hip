// LDS-DMA split GEMM, tile configuration 0: 2 x 2 waves, wave tile 64 x 64, block 128 x 128.
#define SP_CFG_ID 0
#define SP_WM 2
#define SP_WN 2
#define SP_TM 2
#define SP_TN 2
#include "gemm_sp_inst.h"
