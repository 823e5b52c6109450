// placeholder -- filled in by the tuned correlation kernels
#include "d2t_tuned.hpp"
namespace d2t { namespace tuned {
bool   corr_fwd_supported(int, int, int, int, int, int) { return false; }
size_t corr_fwd_ws_bytes(int, int, int, int, int, int) { return 0; }
int    corr_fwd_f32(const float*, const float*, float*, int, int, int, int, int, int, void*, hipStream_t) { return D2T_EINVAL; }
bool   corr_bwd_supported(int, int, int, int, int, int) { return false; }
size_t corr_bwd_ws_bytes(int, int, int, int, int, int) { return 0; }
int    corr_bwd_f32(const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, void*, hipStream_t) { return D2T_EINVAL; }
}}
