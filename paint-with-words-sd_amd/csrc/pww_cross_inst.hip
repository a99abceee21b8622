// One (storage type, workgroup width) slice of the general cross-attention kernel's instantiations (pww_cross_kernel.h): compiled with
// -DPWW_INST_F16 or -DPWW_INST_BF16 and -DPWW_INST_NW=2 or 4.
#include "pww_cross_kernel.h"

namespace pww {

#if defined(PWW_INST_F16) && PWW_INST_NW == 2
int cross_dispatch_f16_nw2(const CrossParams &cp, hipStream_t s, bool *launched) { return dispatch_cross_form<f16, 2>(cp, s, launched); }
#elif defined(PWW_INST_F16) && PWW_INST_NW == 4
int cross_dispatch_f16_nw4(const CrossParams &cp, hipStream_t s, bool *launched) { return dispatch_cross_form<f16, 4>(cp, s, launched); }
#elif defined(PWW_INST_BF16) && PWW_INST_NW == 2
int cross_dispatch_bf16_nw2(const CrossParams &cp, hipStream_t s, bool *launched) { return dispatch_cross_form<bf16, 2>(cp, s, launched); }
#elif defined(PWW_INST_BF16) && PWW_INST_NW == 4
int cross_dispatch_bf16_nw4(const CrossParams &cp, hipStream_t s, bool *launched) { return dispatch_cross_form<bf16, 4>(cp, s, launched); }
#else
#error "compile with -DPWW_INST_F16 | -DPWW_INST_BF16 and -DPWW_INST_NW=2 | 4"
#endif

}  // namespace pww
