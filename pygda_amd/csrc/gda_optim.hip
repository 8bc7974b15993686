// The optimiser step, an HBM-bound kernel the framework's generic version runs far from the roof at
// these shapes:
//
//  gda_adam_multi_f32  torch.optim.Adam's update (the optimiser every trainer of the reference builds,
//                      e.g. pygda/models/a2gnn.py:290-294) over all parameter tensors in ONE launch with
//                      2048-element work items, so a step with one 867k-element weight and a few small
//                      ones still fills the chip.  Same arithmetic as torch's (L2 weight decay folded
//                      into the gradient, lerp form of the first moment, bias corrections from
//                      device-resident per-tensor step counters: hipGraph-capturable).
//
// (A one-launch column sum for the bias gradients -- per-block partials folded by the last block to
// finish -- was measured and dropped: the device-scope release it needs writes back the XCD's whole
// dirty L2 on gfx950, 64-77 us per call against 5-9 us for the generic two-kernel reduction.)
#include "gda_common.h"
#include "gda_adam_rule.h"

namespace {

constexpr int TB = 256;
struct AdamTable {
    float* p[GDA_ADAM_MAX_TENSORS];
    const float* g[GDA_ADAM_MAX_TENSORS];
    const float* g2[GDA_ADAM_MAX_TENSORS];                     // second contribution to the gradient, or NULL (gda_adam_multi_sum_f32)
    float* m[GDA_ADAM_MAX_TENSORS];
    float* v[GDA_ADAM_MAX_TENSORS];
    float* step[GDA_ADAM_MAX_TENSORS];
    int64_t first_item[GDA_ADAM_MAX_TENSORS + 1];             // prefix sums of work items
    int64_t numel[GDA_ADAM_MAX_TENSORS];
    int n;
};
constexpr int ITEM = 2048;                                    // elements per work item (8 per thread)

__global__ void __launch_bounds__(TB)
k_adam(AdamTable t, float lr, float beta1, float beta2, float eps, float weight_decay) {
    const int64_t item = blockIdx.x;
    int k = 0;
    while (k + 1 < t.n && item >= t.first_item[k + 1]) ++k;
    const int64_t base = (item - t.first_item[k]) * ITEM;
    const GdaAdamCoef coef = gda_adam_coef(*t.step[k], lr, beta1, beta2);      // the counter is already incremented for this update
    float* __restrict__ p = t.p[k];
    const float* __restrict__ g = t.g[k];
    const float* __restrict__ g2 = t.g2[k];
    float* __restrict__ m = t.m[k];
    float* __restrict__ v = t.v[k];
    const int64_t n = t.numel[k];
#pragma unroll
    for (int u = 0; u < ITEM / TB; ++u) {
        const int64_t i = base + u * TB + threadIdx.x;
        if (i >= n) break;
        float gi = g[i];
        if (g2) gi = gi + g2[i];                               // what autograd's accumulation would have stored (one rounding)
        gda_adam_element(p + i, gi, m + i, v + i, coef, beta1, beta2, eps, weight_decay);      // gda_adam_rule.h
    }
}

__global__ void k_step_inc(AdamTable t) {
    if ((int)threadIdx.x < t.n) *t.step[threadIdx.x] += 1.0f;
}

struct BumpTable {
    float* step[GDA_ADAM_MAX_TENSORS];
    int64_t* counter;
    int n;
};
__global__ void k_step_bump(BumpTable t) {
    if ((int)threadIdx.x < t.n) *t.step[threadIdx.x] += 1.0f;
    if (threadIdx.x == 63 && t.counter) *t.counter += 1;
}

}  // namespace

extern "C" int gda_step_bump(int64_t* counter, float* const* steps, int n_steps, gda_stream_t stream_) {
    if (n_steps < 0 || n_steps > GDA_ADAM_MAX_TENSORS) return GDA_E_SIZE;
    if (n_steps > 0 && !steps) return GDA_E_NULL;
    if (n_steps == 0 && !counter) return GDA_OK;
    BumpTable t;
    t.n = n_steps;
    t.counter = counter;
    for (int k = 0; k < n_steps; ++k) {
        if (!steps[k]) return GDA_E_NULL;
        for (int j = 0; j < k; ++j)
            if (steps[j] == steps[k]) return GDA_E_UNSUPPORTED;       // one increment per counter and launch
        t.step[k] = steps[k];
    }
    k_step_bump<<<1, 64, 0, (hipStream_t)stream_>>>(t);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_adam_multi_f32(const gda_adam_tensor* tensors, int n_tensors, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, gda_stream_t stream_) {
    return gda_adam_multi_ex_f32(tensors, n_tensors, lr, beta1, beta2, eps, weight_decay, 0, stream_);
}

extern "C" int gda_adam_multi_ex_f32(const gda_adam_tensor* tensors, int n_tensors, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, int flags, gda_stream_t stream_) {
    return gda_adam_multi_sum_f32(tensors, nullptr, n_tensors, lr, beta1, beta2, eps, weight_decay, flags, stream_);
}

extern "C" int gda_adam_multi_sum_f32(const gda_adam_tensor* tensors, const float* const* grad2, int n_tensors, float lr,
                                      float beta1, float beta2, float eps, float weight_decay, int flags,
                                      gda_stream_t stream_) {
    if (flags & ~GDA_ADAM_STEPS_BUMPED) return GDA_E_UNSUPPORTED;
    if (n_tensors < 0 || n_tensors > GDA_ADAM_MAX_TENSORS) return GDA_E_SIZE;
    if (n_tensors == 0) return GDA_OK;
    if (!tensors) return GDA_E_NULL;
    AdamTable t;
    t.n = 0;
    int64_t items = 0;
    for (int k = 0; k < n_tensors; ++k) {
        const gda_adam_tensor& e = tensors[k];
        if (e.numel < 0) return GDA_E_SIZE;
        if (e.numel == 0) continue;
        if (!e.param || !e.grad || !e.exp_avg || !e.exp_avg_sq || !e.step) return GDA_E_NULL;
        if (grad2 && grad2[k] && (grad2[k] == e.grad || (const float*)e.param == grad2[k])) return GDA_E_ALIAS;
        t.p[t.n] = e.param; t.g[t.n] = e.grad; t.m[t.n] = e.exp_avg; t.v[t.n] = e.exp_avg_sq; t.step[t.n] = e.step;
        t.g2[t.n] = grad2 ? grad2[k] : nullptr;
        t.numel[t.n] = e.numel;
        t.first_item[t.n] = items;
        items += gda_cdiv(e.numel, ITEM);
        ++t.n;
    }
    t.first_item[t.n] = items;
    if (items >= INT32_MAX) return GDA_E_SIZE;
    hipStream_t stream = (hipStream_t)stream_;
    if (items == 0) return GDA_OK;
    if (!(flags & GDA_ADAM_STEPS_BUMPED)) {
        k_step_inc<<<1, 64, 0, stream>>>(t);
        GDA_LAUNCH_CHECK();
    }
    k_adam<<<(unsigned)items, TB, 0, stream>>>(t, lr, beta1, beta2, eps, weight_decay);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
