// d2t_pool_backward.hip -- ONE translation unit for the tuned pooling backward kernels: the GEMM / row / plane forms (d2t_pool_bwd.hip)
// and the sorted-corner-list form for more than 32 targets (d2t_pool_sorted.hip).  The parts keep their own files; this unit only
// compiles them together -- the library is ten units.
#include "d2t_pool_bwd.hip"
#include "d2t_pool_sorted.hip"
