// LDS-DMA split GEMM, tile configuration 1: 2 x 2 waves, wave tile 64 x 96, block 128 x 192.
#define SP_CFG_ID 1
#define SP_WM 2
#define SP_WN 2
#define SP_TM 2
#define SP_TN 3
#include "gemm_sp_inst.h"
