// Stand-alone micro-benchmark: what bounds a streaming exp-sum over an 80 MB matrix?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ L, long m, long n, float bw0, double* out) {
    float nib[5]; float f = 1.f;
    for (int q = 0; q < 5; ++q) { nib[q] = -1.f / (bw0 * f); f *= 2.f; }
    const int t = blockIdx.y;
    const float* P = L + (long)t * m * m;
    float local = 0.f;
    const long per_row = m / 4;
    for (long r0 = (long)blockIdx.x * 8; r0 < m; r0 += (long)gridDim.x * 8) {
        const long r1 = r0 + 8 < m ? r0 + 8 : m;
        for (long g = threadIdx.x; g < (r1 - r0) * per_row; g += 256) {
            const long i = r0 + g / per_row, j = (g % per_row) * 4;
            const float4 dv = *reinterpret_cast<const float4*>(P + i * m + j);
            const float dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float kv = 0.f;
                if (MODE == 0) { for (int q = 0; q < 5; ++q) kv += __expf(dd[c] * nib[q]); }
                if (MODE == 1) { for (int q = 0; q < 5; ++q) kv += dd[c] * nib[q]; }
                if (MODE == 2) { for (int q = 0; q < 5; ++q) kv += expf(dd[c] * nib[q]); }
                local += ((i < n) == (j + c < n)) ? kv : -kv;
            }
        }
    }
    if (MODE == 3) local = P[blockIdx.x];
    __shared__ float red[256];
    red[threadIdx.x] = local; __syncthreads();
    if (threadIdx.x == 0) { double s = 0; for (int k2 = 0; k2 < 256; ++k2) s += red[k2]; out[t * gridDim.x + blockIdx.x] = s; }
}

template <int MODE> float run(const float* L, long m, double* out, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) k<MODE><<<dim3(grid, 5), 256>>>(L, m, m / 2, 3.0f, out);
    hipEventRecord(a);
    for (int it = 0; it < 20; ++it) k<MODE><<<dim3(grid, 5), 256>>>(L, m, m / 2, 3.0f, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 20 * 1000;
}

int main() {
    const long m = 2000; float* L; double* out;
    CK(hipMalloc(&L, 5 * m * m * 4)); CK(hipMalloc(&out, 8 * 5 * 4096));
    std::vector<float> h(5 * m * m); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 97) * 0.1f;
    CK(hipMemcpy(L, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int grid : {64, 256, 1024}) {
        printf("grid %4d x5: __expf %.1f us | no-exp %.1f us | expf %.1f us | empty %.1f us\n", grid,
               run<0>(L, m, out, grid), run<1>(L, m, out, grid), run<2>(L, m, out, grid), run<3>(L, m, out, grid));
    }
    return 0;
}
