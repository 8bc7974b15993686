// Stand-alone micro-benchmark of the pair-distance tile kernel: which phase bounds it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int TB = 256, TILE = 64, DK = 32, LDT = TILE + 4;

template <int MODE>   // 0 full, 1 skip compute, 2 skip LDS staging (compute on stale LDS), 3 skip global loads
__global__ void __launch_bounds__(TB)
k(const float* __restrict__ rows, long d, long m, float* __restrict__ l2) {
    __shared__ __attribute__((aligned(16))) float As[DK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[DK][LDT];
    const int t = blockIdx.z;
    if (blockIdx.x < blockIdx.y) return;
    const long i0 = (long)blockIdx.y * TILE, j0 = (long)blockIdx.x * TILE;
    const int tid = threadIdx.x, ty = tid / 16, tx = tid % 16;
    const int lr = tid / 8, kq = (tid % 8) * 4;
    const float* base = rows + (long)t * m * d;
    float acc[4][4] = {};
    for (long k0 = 0; k0 < d; k0 += DK) {
        float4 va[2], vb[2];
        for (int q = 0; q < 2; ++q) {
            if (MODE != 3) {
                va[q] = *reinterpret_cast<const float4*>(base + (i0 + lr + 32 * q) * d + k0 + kq);
                vb[q] = *reinterpret_cast<const float4*>(base + (j0 + lr + 32 * q) * d + k0 + kq);
            } else { va[q] = make_float4(1, 2, 3, 4); vb[q] = make_float4(2, 3, 4, 5); }
        }
        __syncthreads();
        if (MODE != 2) {
            for (int q = 0; q < 2; ++q) {
                const int r = lr + 32 * q;
                As[kq + 0][r] = va[q].x; As[kq + 1][r] = va[q].y; As[kq + 2][r] = va[q].z; As[kq + 3][r] = va[q].w;
                Bs[kq + 0][r] = vb[q].x; Bs[kq + 1][r] = vb[q].y; Bs[kq + 2][r] = vb[q].z; Bs[kq + 3][r] = vb[q].w;
            }
        } else { acc[0][0] += va[0].x + vb[0].x + va[1].y + vb[1].y; }
        __syncthreads();
        if (MODE != 1) {
#pragma unroll 8
            for (int kk = 0; kk < DK; ++kk) {
                const float4 a4 = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
                const float4 b4 = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) { const float df = bv[b] - av[a]; acc[a][b] = fmaf(df, df, acc[a][b]); }
            }
        } else { acc[0][0] += As[tid % DK][tid % 64] + Bs[tid % DK][tid % 64]; }
    }
    float* out = l2 + (long)t * m * m;
    for (int a = 0; a < 4; ++a) {
        const long i = i0 + ty * 4 + a, j = j0 + tx * 4;
        if (i < m && j + 3 < m) *reinterpret_cast<float4*>(out + i * m + j) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
    }
    if (blockIdx.x != blockIdx.y)
        for (int b = 0; b < 4; ++b) {
            const long j = j0 + tx * 4 + b, i = i0 + ty * 4;
            if (j < m && i + 3 < m) *reinterpret_cast<float4*>(out + j * m + i) = make_float4(acc[0][b], acc[1][b], acc[2][b], acc[3][b]);
        }
}

template <int MODE> float run(const float* rows, long d, long m, float* l2) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const unsigned nt = (unsigned)((m + TILE - 1) / TILE);
    for (int w = 0; w < 3; ++w) k<MODE><<<dim3(nt, nt, 5), TB>>>(rows, d, m, l2);
    (void)hipEventRecord(a);
    for (int it = 0; it < 20; ++it) k<MODE><<<dim3(nt, nt, 5), TB>>>(rows, d, m, l2);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 20 * 1000;
}

int main() {
    const long m = 2048, d = 128; float *rows, *l2;     // m padded to a tile multiple for the probe
    (void)hipMalloc(&rows, 5 * m * d * 4); (void)hipMalloc(&l2, 5 * m * m * 4);
    std::vector<float> h(5 * m * d); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 7) % 101) * 0.01f;
    (void)hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("full %.1f us | no-compute %.1f us | no-LDS-stage %.1f us | no-global-load %.1f us\n",
           run<0>(rows, d, m, l2), run<1>(rows, d, m, l2), run<2>(rows, d, m, l2), run<3>(rows, d, m, l2));
    return 0;
}
