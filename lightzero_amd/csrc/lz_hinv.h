// lz_hinv.h -- the inverse scalar transform h^-1 of the categorical value / value-prefix / reward heads, ONE definition for every
// kernel that ends a head (k_heads, k_heads_mm, the split heads in k_chain_w's prologue, k_rowfinal).
//
// Reference: InverseScalarTransform.__call__, lzero/policy/scaling_transform.py:82-92, evaluated by torch in fp32:
//     tmp = (torch.sqrt(1 + 4 * epsilon * (torch.abs(value) + 1 + epsilon)) - 1) / (2 * epsilon)
//     output = torch.sign(value) * (tmp * tmp - 1)
// i.e. eight individually rounded binary32 operations in this order: |v| + 1, + 0.001f, * 0.004f, 1 + ., sqrt, - 1, / 0.002f (a true
// division on torch's CPU path), t * t, - 1, * sign.  The formula subtracts 1 from a square root ~1.004 and divides by 0.002, so its
// output moves in steps of ~1.3e-4 (1 + |x|): a fused multiply-add anywhere in it lands on a neighbouring step for some inputs.  The
// translation units that include this header are compiled with FMA contraction ON (the matrix kernels want it), so contraction is
// switched off for this function by pragma -- left to the default the compiler emitted v_fmamk_f32 for `1 + 0.004 t` and
// v_fma_f32 for `t t - 1`, which differs from torch's result on a third of the inputs in [-2, 2] by up to 3.6e-4 (found in round 4 by
// disassembling; tests/test_hinv_gpu.py now holds every compiled copy BIT-EQUAL to
// torch's evaluation of the reference formula over the whole support range, lz_debug_inverse_scalar_transform).
// `/` and sqrtf are the correctly rounded HIP defaults (-fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float lz_inverse_scalar_transform(float value)
{
#pragma clang fp contract(off)
    const float eps = 0.001f;
    float t = fabsf(value) + 1.0f;
    t = t + eps;
    t = 0.004f * t;
    t = 1.0f + t;
    t = sqrtf(t);
    t = t - 1.0f;
    t = t / 0.002f;
    const float sgn = (value > 0.0f) ? 1.0f : (value < 0.0f ? -1.0f : 0.0f);   // torch.sign (sign(+-0) = 0; a NaN value leaves NaN in r)
    const float tt = t * t;
    const float r = tt - 1.0f;
    return sgn * r;
}
