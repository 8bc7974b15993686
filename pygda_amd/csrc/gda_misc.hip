// ABI version, status strings, and the feature-row gather used for mini-batch assembly.
#include "gda_common.h"

namespace {

constexpr int TB = 256;

// out[r, :] = x[idx[r], :]  -- the x[n_id] slice PyG's NeighborLoader performs
// (pygda/models/a2gnn.py:260-277).  A lane group of G lanes moves one row with 16-byte
// accesses: fully coalesced reads of whole feature rows, HBM bound (2*n_out*d*4 bytes).
template <int VEC>
__global__ void __launch_bounds__(TB)
k_gather(const float* __restrict__ x, int64_t ldx, int d, const int64_t* __restrict__ idx,
         int64_t n_out, float* __restrict__ out, int64_t ldo) {
    const int per_row = (d + VEC - 1) / VEC;                 // vector slots per row
    const int64_t total = n_out * per_row;
    for (int64_t s = (int64_t)blockIdx.x * TB + threadIdx.x; s < total; s += (int64_t)gridDim.x * TB) {
        const int64_t r = s / per_row;
        const int c = (int)(s % per_row) * VEC;
        const float* p = x + idx[r] * ldx + c;
        float* q = out + r * ldo + c;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(q) = *reinterpret_cast<const float4*>(p);
        else *q = *p;
    }
}

// Graph-level readout  out[g, :] = (sum of x[seg[g] .. seg[g+1], :]) / max(seg[g+1] - seg[g], 1)  -- PyG's
// global_mean_pool(x, batch) for the sorted `batch` vector a DataLoader produces (pygda/nn/a2gnn_base.py:140-141):
// the rows of a graph are added in node order (the CPU scatter's order), then divided by the node count.
// One thread per (graph, column): a graph of the TU datasets has tens of nodes; reads are coalesced across columns.
__global__ void __launch_bounds__(TB)
k_segment_mean_fwd(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ seg, int64_t G, int d,
                   float* __restrict__ out, int64_t ldo) {
    const int64_t s = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= G * d) return;
    const int64_t g = s / d;
    const int c = (int)(s % d);
    const int64_t b = seg[g], e = seg[g + 1];
    float acc = 0.f;
    for (int64_t r = b; r < e; ++r) acc = __fadd_rn(acc, x[r * ldx + c]);
    const float cnt = (float)(e - b > 1 ? e - b : 1);
    out[g * ldo + c] = __fdiv_rn(acc, cnt);
}

// gx[i, :] = gout[batch[i], :] / max(count(batch[i]), 1)
__global__ void __launch_bounds__(TB)
k_segment_mean_bwd(const float* __restrict__ gout, int64_t ldg, const int64_t* __restrict__ seg,
                   const int64_t* __restrict__ batch, int64_t n, int d, float* __restrict__ gx, int64_t ldx) {
    const int64_t s = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= n * d) return;
    const int64_t i = s / d;
    const int c = (int)(s % d);
    const int64_t g = batch[i];
    const int64_t m = seg[g + 1] - seg[g];
    gx[i * ldx + c] = __fdiv_rn(gout[g * ldg + c], (float)(m > 1 ? m : 1));
}

// dst[i] = src[i] for n 16-byte pieces + tail bytes: a copy whose SOURCE may be pinned host memory read over the bus
__global__ void __launch_bounds__(TB)
k_copy_pieces(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t pieces, const unsigned char* __restrict__ tail_src,
              unsigned char* __restrict__ tail_dst, int tail) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * TB) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) tail_dst[threadIdx.x] = tail_src[threadIdx.x];
}

}  // namespace

// Host-to-device copy of a PINNED (hipHostMalloc / hipHostRegister) block as a KERNEL on `stream` that reads the host
// memory over the bus -- for the small per-step blocks a captured training step is fed with (row samples, interpolation
// weights: 0.1 - 1 MB).  hipMemcpyAsync hands such a copy to a DMA engine; behind a running graph the hand-over from the
// compute queue to the engine and back costs 200 - 370 us of idle device time per replay (profiles/r6_experiments.txt, 10),
// a kernel in the stream costs its own 20 - 30 us.  GDA_E_UNSUPPORTED when the host block is not device-mapped.
extern "C" int gda_copy_from_pinned(void* dst, const void* src_host, size_t bytes, gda_stream_t stream) {
    if (bytes == 0) return GDA_OK;
    if (!dst || !src_host) return GDA_E_NULL;
    void* dsrc = nullptr;
    if (hipHostGetDevicePointer(&dsrc, const_cast<void*>(src_host), 0) != hipSuccess || !dsrc) {
        (void)hipGetLastError();
        return GDA_E_UNSUPPORTED;
    }
    if (((uintptr_t)dst | (uintptr_t)dsrc) % 16 != 0) return GDA_E_UNSUPPORTED;
    const int64_t pieces = (int64_t)(bytes / 16);
    const int tail = (int)(bytes % 16);
    int64_t blocks = gda_cdiv(pieces > 0 ? pieces : 1, TB);
    if (blocks > 1024) blocks = 1024;
    k_copy_pieces<<<(unsigned)blocks, TB, 0, (hipStream_t)stream>>>(
        static_cast<const uint4*>(dsrc), static_cast<uint4*>(dst), pieces,
        static_cast<const unsigned char*>(dsrc) + pieces * 16, static_cast<unsigned char*>(dst) + pieces * 16, tail);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_abi_version(void) { return 1; }

extern "C" const char* gda_status_string(int status) {
    switch (status) {
        case GDA_OK: return "ok";
        case GDA_E_NULL: return "gda: required pointer is NULL";
        case GDA_E_SIZE: return "gda: negative, overflowing or inconsistent size";
        case GDA_E_WORKSPACE: return "gda: workspace too small";
        case GDA_E_UNSUPPORTED: return "gda: unsupported configuration";
        case GDA_E_ALIAS: return "gda: output aliases an input";
        default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    if (status <= GDA_E_RCCL) return "gda: RCCL call failed (ncclResult_t = GDA_E_RCCL - status)";
    return "gda: unknown status";
}

extern "C" int gda_gather_rows_f32(const float* x, int64_t ldx, int64_t d, const int64_t* idx,
                                   int64_t n_out, float* out, int64_t ldo, gda_stream_t stream_) {
    if (n_out < 0 || d < 0 || d >= INT32_MAX || ldx < d || ldo < d) return GDA_E_SIZE;
    if (n_out == 0 || d == 0) return GDA_OK;
    if (!x || !idx || !out) return GDA_E_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    const bool v4 = (d % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) &&
                    ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const int64_t slots = n_out * (v4 ? d / 4 : d);
    int64_t grid = gda_cdiv(slots, TB);
    if (grid > 256 * 16) grid = 256 * 16;                     // grid-stride beyond 16 workgroups per CU
    if (v4) k_gather<4><<<(unsigned)grid, TB, 0, stream>>>(x, ldx, (int)d, idx, n_out, out, ldo);
    else k_gather<1><<<(unsigned)grid, TB, 0, stream>>>(x, ldx, (int)d, idx, n_out, out, ldo);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_segment_mean_fwd_f32(const float* x, int64_t ldx, const int64_t* seg_ptr, int64_t G, int64_t d,
                                        float* out, int64_t ldo, gda_stream_t stream_) {
    if (G < 0 || d < 0 || d >= INT32_MAX || ldx < d || ldo < d) return GDA_E_SIZE;
    if (G == 0 || d == 0) return GDA_OK;
    if (!x || !seg_ptr || !out) return GDA_E_NULL;
    k_segment_mean_fwd<<<(unsigned)gda_cdiv(G * d, TB), TB, 0, (hipStream_t)stream_>>>(x, ldx, seg_ptr, G, (int)d, out, ldo);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}

extern "C" int gda_segment_mean_bwd_f32(const float* gout, int64_t ldg, const int64_t* seg_ptr, const int64_t* batch,
                                        int64_t n, int64_t d, float* gx, int64_t ldx, gda_stream_t stream_) {
    if (n < 0 || d < 0 || d >= INT32_MAX || ldx < d || ldg < d) return GDA_E_SIZE;
    if (n == 0 || d == 0) return GDA_OK;
    if (!gout || !seg_ptr || !batch || !gx) return GDA_E_NULL;
    k_segment_mean_bwd<<<(unsigned)gda_cdiv(n * d, TB), TB, 0, (hipStream_t)stream_>>>(gout, ldg, seg_ptr, batch, n, (int)d, gx, ldx);
    GDA_LAUNCH_CHECK();
    return GDA_OK;
}
