// One storage type's instantiations of the attention kernels (pww_attn_kernel.h): compiled with -DPWW_INST_F16 or -DPWW_INST_BF16.
#include "pww_attn_kernel.h"

namespace pww {

#if defined(PWW_INST_F16)
int attn_dispatch_f16(const AttnParams &p, hipStream_t s) { return dispatch_nw<f16>(p, s); }
bool attn_wide_groups_rule(int B, int H, int N, int D) { return wide_groups(B, H, N, D); }
#elif defined(PWW_INST_BF16)
int attn_dispatch_bf16(const AttnParams &p, hipStream_t s) { return dispatch_nw<bf16>(p, s); }
#else
#error "compile with -DPWW_INST_F16 or -DPWW_INST_BF16"
#endif

}  // namespace pww
