// LDS-DMA split GEMM, tile configuration 6: 4 x 1 waves, wave tile 32 x 32, block 128 x 32, 16-k stages (KS = 1) - the window
// conv (A mode 5) for 3x3 / stride-1 convs whose channels come in slices of 16 (the detector's 16-channel bottlenecks at
// 160 x 160, which otherwise fall back to nine per-lane tap gathers with a K tail).
#define SP_CFG_ID 6
#define SP_WM 4
#define SP_WN 1
#define SP_TM 1
#define SP_TN 1
#define SP_KS_VALUE 1
#include "gemm_sp_inst.h"
