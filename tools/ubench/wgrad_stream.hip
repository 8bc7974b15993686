// Where does the tall weight gradient's time go?  (round 6: the fp32-MFMA kernel, the split-fp16 kernel, dealt chunks, two
// chunks in flight and two workgroups per CU all take ~60 us for 158,720 rows x (128 + 128) floats = 2.7 TB/s.)
// The same two operands streamed by kernels that share one property of the real kernel at a time:
//   v0  256 workgroups x 512 threads, one contiguous row slab each, 16-byte loads in the real kernel's mapping, values summed
//       in registers: the launch shape alone
//   v1  v0 + the chunk written to LDS and a barrier per 32 rows (no arithmetic on it)
//   v2  a flat grid-stride stream over both arrays (2048 x 256 threads): what the memory system gives this footprint
//   v3  v0 with 1024 workgroups (four per CU)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/wgrad_stream.hip -o /tmp/wgrad_stream && /tmp/wgrad_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <bool LDS>
__global__ void __launch_bounds__(512) k_slab(const float* __restrict__ G, const float* __restrict__ X, long M, long rows_per_slab,
                                              float* __restrict__ out) {
    __shared__ float4 img[LDS ? 2 * 32 * 64 : 1];          // one 32-row chunk of both operands (32 KB)
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long m0 = r0; m0 < r1; m0 += 32) {
        float4 vx[2], vg[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + 512 * q, row = idx / 32, c4 = idx % 32;
            long r = m0 + row; r = r < r1 ? r : r1 - 1;
            vx[q] = *reinterpret_cast<const float4*>(X + r * 128 + 4 * c4);
            vg[q] = *reinterpret_cast<const float4*>(G + r * 128 + 4 * c4);
        }
        if (LDS) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { img[tid + 512 * q] = vx[q]; img[2048 + tid + 512 * q] = vg[q]; }
            __syncthreads();
            const float4 a = img[(tid * 7) & 2047], b = img[2048 + ((tid * 13) & 2047)];
            acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
            __syncthreads();
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                acc.x += vx[q].x + vg[q].x; acc.y += vx[q].y + vg[q].y; acc.z += vx[q].z + vg[q].z; acc.w += vx[q].w + vg[q].w;
            }
        }
    }
    out[(long)blockIdx.x * 512 + tid] = acc.x + acc.y + acc.z + acc.w;
}

__global__ void __launch_bounds__(256) k_flat(const float4* __restrict__ G, const float4* __restrict__ X, long n4, float* __restrict__ out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 a = G[i], b = X[i];
        acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
    }
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    const long M = 158720, d = 128;
    float *G, *X, *out;
    CK(hipMalloc(&G, M * d * 4)); CK(hipMalloc(&X, M * d * 4)); CK(hipMalloc(&out, 2048 * 512 * 4));
    CK(hipMemset(G, 0, M * d * 4)); CK(hipMemset(X, 0, M * d * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double mb = 2.0 * M * d * 4 / 1e6;
    auto report = [&](const char* name, float ms, int reps) { printf("%s  %.1f us  %.2f TB/s\n", name, ms * 1e3 / reps, mb / (ms * 1e3 / reps)); };
    const int reps = 40;
    for (int variant = 0; variant < 5; ++variant) {
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) CK(hipEventRecord(a));
            for (int r = 0; r < (pass ? reps : 5); ++r) {
                if (variant == 0) k_slab<false><<<256, 512>>>(G, X, M, 620, out);
                if (variant == 1) k_slab<true><<<256, 512>>>(G, X, M, 620, out);
                if (variant == 2) k_flat<<<2048, 256>>>((const float4*)G, (const float4*)X, M * d / 4, out);
                if (variant == 3) k_slab<false><<<1024, 512>>>(G, X, M, 155, out);
                if (variant == 4) k_slab<true><<<1024, 512>>>(G, X, M, 155, out);
            }
            if (pass) { CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b));
                        const char* names[] = {"v0 slabs 256x512        ", "v1 + LDS + barrier      ", "v2 flat 2048x256        ", "v3 slabs 1024x512       ", "v4 1024x512 + LDS       "};
                        report(names[variant], ms, reps); }
        }
    }
    return 0;
}
