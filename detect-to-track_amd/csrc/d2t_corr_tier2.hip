// d2t_corr_tier2.hip -- ONE translation unit for the correlation's second tier (everything outside the tuned envelope, bit-identical to
// the anchor kernels of d2t_generic.hip): the blocked / tiled kernels and the matrix-pipe forward for d_max <= 8.  The two parts keep
// their own files (they share nothing but d2t_kernels.hpp); this unit only compiles them together -- the library is ten units.
#include "d2t_corr_blocked.hip"
#include "d2t_corr_fwd_mfma.hip"
