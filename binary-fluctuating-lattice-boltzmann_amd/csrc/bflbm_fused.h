// bflbm_fused.h -- fused plane-marching kernel (schedule 1).  Placeholder until implemented.
#ifndef BFLBM_FUSED_H_
#define BFLBM_FUSED_H_
#include "bflbm_kernels.h"
static inline int fused_launch(const double*, double*, const double*, const double*, const Geo&, const DevParams&,
                               int, int, uint32_t, int, hipStream_t) { return 1; }
#endif
