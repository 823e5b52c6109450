// placeholder -- filled in by the tuned pooling kernels
#include "d2t_tuned.hpp"
namespace d2t { namespace tuned {
bool   roipool_fwd_supported(int, int, int, int, int) { return false; }
size_t roipool_fwd_ws_bytes(int, int, int, int, int) { return 0; }
int    roipool_fwd_f32(const float*, const float*, float*, int, int, int, int, int, void*, hipStream_t) { return D2T_EINVAL; }
bool   roipool_bwd_supported(int, int, int, int, int) { return false; }
size_t roipool_bwd_ws_bytes(int, int, int, int, int) { return 0; }
int    roipool_bwd_f32(const float*, const float*, float*, int, int, int, int, int, void*, hipStream_t) { return D2T_EINVAL; }
bool   psroipool_bwd_supported(int, int, int, int, int) { return false; }
size_t psroipool_bwd_ws_bytes(int, int, int, int, int) { return 0; }
int    psroipool_bwd_f32(const float*, const float*, float*, int, int, int, int, int, void*, hipStream_t) { return D2T_EINVAL; }
}}
