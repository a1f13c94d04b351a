/* Entry points that exist only in probe builds of libmia_hip (hipcc -DMIA_EXPERIMENTS; tools/r5_store_hazard.sh): the column-reduce
 * epilogue of round 4 -- the input-gradient conv of the CONSUMING block also adds up the producing block's norm-backward sums (built,
 * correct, +-0 in the step: DESIGN section 4, round 4) -- kept because the store-data hazard was found in it and is re-measured through it
 * (profiles/r05_store_hazard.txt).  Not part of the drop-in boundary (include/mia_hip.h). */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int mia_conv_cr_supported(int mode, int dtype, int c1, int nout, int hout, int wout);
int mia_conv_mma_cr(int mode, int dtype, const void* in1, int c1, const void* wpack, int npad, int kpad, int flip_taps,
                    void* out, int nout, const void* y_prod, const float* scale, const float* shift, const float* xa,
                    const float* xb, float slope, float* partials, int n, int hin, int win, int hout, int wout, void* stream);
int mia_norm_act_bwd_pre(const void* dz, const void* y, void* dy, int dtype, const float* scale, const float* shift,
                         const float* xa, const float* xb, const float* ysum, int n, int64_t hw, int c, int mode,
                         int fixed_stats, float slope, int parts, const float* partials, float* c1, float* c2,
                         float* dgamma, float* dbeta, float* dbias, int accumulate, void* amax_out, void* stream);
#ifdef __cplusplus
}
#endif
