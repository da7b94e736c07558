// The cooperative K > 1 backward of sweep_mfma.hip as a translation unit of its own: same source,
// compiled with -fno-slp-vectorize (see the Makefile and dispatch_bwd_coop in sweep_mfma.hip).
#define MDMM_COOP_TU 1
#include "sweep_mfma.hip"
