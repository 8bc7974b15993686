// torch.optim.Adam's update of ONE element (L2 weight decay folded into the gradient, lerp form of the first moment,
// bias corrections from a step counter that already holds t): the one statement of the rule, shared by the multi-tensor
// launch (gda_optim.hip) and by kernels that apply the update where they produce the gradient (gda_critic.hip) -- the
// same operations in the same order, so the two give the same bits (-ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

struct GdaAdamCoef { float step_size, bc2_sqrt; };

__device__ __forceinline__ GdaAdamCoef gda_adam_coef(float step, float lr, float beta1, float beta2) {
    const float bc1 = 1.0f - powf(beta1, step);
    const float bc2 = 1.0f - powf(beta2, step);
    return GdaAdamCoef{lr / bc1, sqrtf(bc2)};
}

// the rule on values already in registers: returns the new parameter, updates the moments
__device__ __forceinline__ float gda_adam_value(float pi, float gi, float& mi, float& vi, const GdaAdamCoef c, float beta1,
                                                float beta2, float eps, float weight_decay) {
    if (weight_decay != 0.f) gi = gi + weight_decay * pi;
    mi = mi + (gi - mi) * (1.0f - beta1);                // torch: exp_avg.lerp_(grad, 1 - beta1)
    vi = vi * beta2 + (1.0f - beta2) * gi * gi;          // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    const float denom = sqrtf(vi) / c.bc2_sqrt + eps;
    return pi - c.step_size * (mi / denom);              // param.addcdiv_(exp_avg, denom, -step_size)
}

__device__ __forceinline__ void gda_adam_element(float* __restrict__ p, float gi, float* __restrict__ m, float* __restrict__ v,
                                                 const GdaAdamCoef c, float beta1, float beta2, float eps, float weight_decay) {
    float mi = *m, vi = *v;
    *p = gda_adam_value(*p, gi, mi, vi, c, beta1, beta2, eps, weight_decay);
    *m = mi;
    *v = vi;
}
