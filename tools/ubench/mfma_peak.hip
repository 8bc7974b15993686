// Stand-alone probe (NOT part of libgda_hip.so): what the fp32 matrix cores sustain with NO memory traffic at all --
// every wave runs independent chains of v_mfma_f32_32x32x2_f32 on registers -- for 1, 2 and 4 waves per SIMD and 4 or
// 8 accumulators per wave.  The nominal figure (MI355X_MICROARCH.md) is 157.3 TF = 256 CUs x 4 SIMDs x 64 flop/clk x
// 2.4 GHz; the number measured here is the ceiling a GEMM kernel on this part can approach (clock under matrix load
// included).   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
using f32x16 = __attribute__((ext_vector_type(16))) float;

// BURST: each accumulator's MFMAs issued four in a row (dependent back to back), as a compiler may order them
template <int NACC, bool BURST = false>
__global__ void __launch_bounds__(256) k_mfma(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int i = 0; i < iters; ++i) {
        if constexpr (BURST) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < NACC; ++t)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC, bool BURST = false>
void run(int wg_per_cu, int iters) {
    float* out;
    HIP_OK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu;
    k_mfma<NACC, BURST><<<grid, 256>>>(out, 10, 1.f, 1.f);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipEventRecord(e0));
    k_mfma<NACC, BURST><<<grid, 256>>>(out, iters, 1.f, 1.f);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)grid * 4 /* waves */ * iters * 8.0 * NACC * 4096.0;
    std::printf("{\"acc_per_wave\": %d, \"burst4\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f, \"frac_of_157.3\": %.3f}\n", NACC, (int)BURST,
                wg_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
    HIP_OK(hipFree(out));
}

int main() {
    for (int w : {1, 2, 4}) { run<4>(w, 4000 / w); run<8>(w, 2000 / w); }
    // few accumulators per wave (the MMD backward holds two 32 x 32 tiles per wave, four waves per SIMD)
    for (int w : {2, 4}) { run<1>(w, 16000 / w); run<2>(w, 8000 / w); run<2, true>(w, 8000 / w); run<4, true>(w, 4000 / w); }
    return 0;
}
