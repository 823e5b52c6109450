/*
 * torch_compat.h -- the ONLY adaptation applied when building the reference's own
 * sources for oracle/_ref/ (force-included with `-include`; no reference file is
 * copied, edited or shadowed).
 *
 * The reference pins torch==1.1.0 (/root/reference/requirements.txt:8) and its launchers
 * spell the dtype dispatch as
 *     AT_DISPATCH_FLOATING_TYPES(tensor.type(), "name", lambda)
 * (e.g. pointwise_correlation_cuda.cu:196).  torch 2.10's macro of the same name takes an
 * at::ScalarType and no longer converts from DeprecatedTypeProperties, so the call does
 * not compile.  This header re-expresses the same macro in terms of torch 2.10's own
 * AT_DISPATCH_SWITCH / AT_DISPATCH_CASE_FLOATING_TYPES, reading the scalar type off the
 * object the reference passes.  It does not touch kernels, launch geometry or arithmetic:
 * every __global__ function, index map and launch in oracle/_ref/ is compiled verbatim
 * by hipcc (HIP accepts the CUDA kernel dialect natively; no hipify pass is run).
 */
#pragma once
#include <ATen/ATen.h>
#include <ATen/Dispatch.h>

#undef AT_DISPATCH_FLOATING_TYPES
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) \
    AT_DISPATCH_SWITCH((TYPE).scalarType(), NAME, AT_DISPATCH_CASE_FLOATING_TYPES(__VA_ARGS__))
