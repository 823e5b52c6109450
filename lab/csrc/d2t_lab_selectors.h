/* lab/d2t_lab_selectors.h -- implementation selectors of the LAB build of libd2t_ops.so (make -C csrc lab: -DD2T_LAB_KERNELS),
 * on top of the four of include/d2t_ops.h.  They demand one specific correlation-backward kernel for same-process A/B measurements
 * and for the tests of those kernels (tests/test_lab_kernels.py, skipped unless D2T_OPS_LIBRARY points at a lab build).  The product
 * library rejects these values with D2T_EINVAL. */
#ifndef D2T_LAB_SELECTORS_H
#define D2T_LAB_SELECTORS_H
enum {
    D2T_LAB_IMPL_STRIP16 = 3, /* the round-2 16-wave strip kernel (lab/d2t_corr_bwd16.inc) */
    D2T_LAB_IMPL_BF16X3  = 4, /* bf16 matrix pipe, every f32 operand split in three bf16 pieces (lab/d2t_corr_bwd8bf.hip) */
    D2T_LAB_IMPL_WIDE8   = 6, /* strips 8 pixels wide x 128 channels (lab/d2t_corr_bwd8w.hip): 81 against 70 us, lost */
    D2T_LAB_IMPL_STRIP4  = 7  /* the product's 8-wave kernel on strips 4 pixels wide, demanded */
};
#endif
