// HBM-bound helpers around the two hot kernels: node mean + broadcast (global_attr block of
// lib/nn/encoders/sgp_spatial_encoder.py:32-34), strided slot copy (the torch.cat of
// sgp_spatial_encoder.py:35 when a block is produced elsewhere), and row gathers (halo packing
// for the multi-GPU hop exchange; IID sampling of lib/datasets/iid_dataset.py:57-99).
#include "common.h"

using sgp::f32x4;

namespace {

// one workgroup per (feature chunk of 64*V floats, batch b): 4 waves stride the rows, lanes
// own V consecutive features; partial sums meet in LDS.
template <int V>
__global__ __launch_bounds__(256) void col_sum_kernel(
        const float* __restrict__ X, long long xrs, long long xbs,
        float* __restrict__ sums /* [batch, feat] */, int n_rows, int feat, int row_split) {
    __shared__ float red[4][64 * V];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int f = blockIdx.x * 64 * V + lane * V;
    const int nsplit = gridDim.z, split = blockIdx.z;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    if (f < feat) {
        const float* base = X + (long long)b * xbs + f;
        for (int r = split * 4 + wave; r < n_rows; r += 4 * nsplit) {
            if constexpr (V == 4) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(base + (long long)r * xrs);
                acc[0] += x.x; acc[1] += x.y; acc[2] += x.z; acc[3] += x.w;
            } else {
                acc[0] += base[(long long)r * xrs];
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) red[wave][lane * V + v] = acc[v];
    __syncthreads();
    if (wave == 0 && f < feat) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float s = red[0][lane * V + v] + red[1][lane * V + v] + red[2][lane * V + v] +
                            red[3][lane * V + v];
            if (f + v < feat) {
                if (nsplit == 1) sums[(long long)b * feat + f + v] = s;
                else atomicAdd(&sums[(long long)b * feat + f + v], s);
            }
        }
    }
    (void)row_split;
}

template <int V>
__global__ __launch_bounds__(256) void bcast_rows_kernel(
        const float* __restrict__ src, float scale, float* __restrict__ Y, long long yrs, long long ybs,
        int n_rows, int feat) {
    const int b = blockIdx.y;
    const int per_row = (feat + V - 1) / V;
    const long long total = (long long)n_rows * per_row;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per_row);
        const int f = (int)(i % per_row) * V;
        float* yp = Y + (long long)b * ybs + (long long)r * yrs + f;
        const float* sp = src + (long long)b * feat + f;
        if constexpr (V == 4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(sp);
            *reinterpret_cast<f32x4*>(yp) = v * scale;
        } else {
            yp[0] = sp[0] * scale;
        }
    }
}

// feat % 4 == 0 and feat <= 1024: a thread keeps its 16 bytes of the (scaled) source row in registers and
// writes them to every row of its workgroup's row range (streamed stores: the block is written once
// and read by nobody on this GPU before it leaves) -- no index arithmetic or source load per element
__global__ __launch_bounds__(256) void bcast_rows_wide_kernel(
        const float* __restrict__ src, float scale, float* __restrict__ Y, long long yrs, long long ybs,
        int n_rows, int feat, int rows_per_wg) {
    const int b = blockIdx.y;
    const int per_row = feat >> 2;                       // float4 pieces per row (<= 256)
    const int rows_in_flight = 256 / per_row;
    const int piece = threadIdx.x % per_row, rsub = threadIdx.x / per_row;
    if (rsub >= rows_in_flight) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long long)b * feat + piece * 4) * scale;
    const int r_end = min(n_rows, (int)(blockIdx.x + 1) * rows_per_wg);
    float* yp = Y + (long long)b * ybs + piece * 4;
    for (int r = blockIdx.x * rows_per_wg + rsub; r < r_end; r += rows_in_flight)
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(yp + (long long)r * yrs));
}

template <int V>
__global__ __launch_bounds__(256) void copy_rows_kernel(
        const float* __restrict__ X, long long xrs, long long xbs,
        float* __restrict__ Y, long long yrs, long long ybs, int n_rows, int feat) {
    const int b = blockIdx.y;
    const int per_row = (feat + V - 1) / V;
    const long long total = (long long)n_rows * per_row;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per_row);
        const int f = (int)(i % per_row) * V;
        const float* xp = X + (long long)b * xbs + (long long)r * xrs + f;
        float* yp = Y + (long long)b * ybs + (long long)r * yrs + f;
        if constexpr (V == 4) *reinterpret_cast<f32x4*>(yp) = *reinterpret_cast<const f32x4*>(xp);
        else yp[0] = xp[0];
    }
}

// out[b, k, :] = X[step ? step[k] : b, node[k], :]
template <int V>
__global__ __launch_bounds__(256) void gather_rows_kernel(
        const float* __restrict__ X, long long xrs, long long xbs,
        const int* __restrict__ step, const int* __restrict__ node, int n_index,
        float* __restrict__ out, long long ors, long long obs, int feat) {
    const int b = blockIdx.y;
    const int per_row = (feat + V - 1) / V;
    const long long total = (long long)n_index * per_row;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i / per_row);
        const int f = (int)(i % per_row) * V;
        const int sb = step ? step[k] : b;
        const float* xp = X + (long long)sb * xbs + (long long)node[k] * xrs + f;
        float* op = out + (long long)b * obs + (long long)k * ors + f;
        if constexpr (V == 4) *reinterpret_cast<f32x4*>(op) = *reinterpret_cast<const f32x4*>(xp);
        else op[0] = xp[0];
    }
}

inline bool vec_ok(const void* p, long long s0, long long s1, int feat) {
    return feat % 4 == 0 && s0 % 4 == 0 && s1 % 4 == 0 && sgp::aligned16(p);
}

inline unsigned grid_for(long long work_items) {
    long long g = (work_items + 255) / 256;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    return (unsigned)g;
}

}  // namespace

// max |x| over a strided [batch, n_rows, feat] view: one atomicMax on the bit pattern of the non-negative
// maximum per workgroup (IEEE order = integer order for non-negative floats; NaN / inf bit patterns win, so a
// non-finite input shows as a non-finite bound).  feat % 4 == 0 takes 16-byte loads.
__global__ __launch_bounds__(256) void abs_max_kernel(const float* x, long long xrs, long long xbs, int n_rows,
                                                      int batch, int feat, unsigned* out) {
    const int b = blockIdx.y;
    const float* xb = x + (long long)b * xbs;
    unsigned m = 0;
    if ((feat & 3) == 0) {
        const int q = feat >> 2;
        const long long total = (long long)n_rows * q;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long r = i / q; const int c = (int)(i - r * q);
            const float4 v = *(const float4*)(xb + r * xrs + 4 * c);
            m = max(m, __float_as_uint(fabsf(v.x))); m = max(m, __float_as_uint(fabsf(v.y)));
            m = max(m, __float_as_uint(fabsf(v.z))); m = max(m, __float_as_uint(fabsf(v.w)));
        }
    } else {
        const long long total = (long long)n_rows * feat;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long r = i / feat; const int c = (int)(i - r * feat);
            m = max(m, __float_as_uint(fabsf(xb[r * xrs + c])));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(part[0], part[1]), max(part[2], part[3])));
}


extern "C" {

int sgp_bcast_rows_f32(const float* src, float scale, float* Y, int64_t yrs, int64_t ybs,
                       int32_t n_rows, int32_t batch, int32_t feat, sgp_stream_t stream) {
    SGP_REQUIRE(src && Y, "sgp_bcast_rows_f32: null pointer");
    SGP_REQUIRE(n_rows >= 0 && batch >= 0 && feat >= 0 && batch <= 65535, "sgp_bcast_rows_f32: bad size");
    if (!n_rows || !batch || !feat) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (vec_ok(Y, yrs, ybs, feat) && sgp::aligned16(src) && feat <= 1024) {
        // enough workgroups to fill the chip, each with a few hundred rows to stream
        long long want = 8192 / (batch < 1 ? 1 : batch);
        if (want < 1) want = 1;
        int rows_per_wg = (int)((n_rows + want - 1) / want);
        if (rows_per_wg < 64) rows_per_wg = 64;
        hipLaunchKernelGGL(bcast_rows_wide_kernel, dim3((n_rows + rows_per_wg - 1) / rows_per_wg, batch), dim3(256),
                           0, s, src, scale, Y, yrs, ybs, n_rows, feat, rows_per_wg);
    } else if (vec_ok(Y, yrs, ybs, feat) && sgp::aligned16(src))
        hipLaunchKernelGGL(bcast_rows_kernel<4>, dim3(grid_for((long long)n_rows * feat / 4), batch), dim3(256),
                           0, s, src, scale, Y, yrs, ybs, n_rows, feat);
    else
        hipLaunchKernelGGL(bcast_rows_kernel<1>, dim3(grid_for((long long)n_rows * feat), batch), dim3(256),
                           0, s, src, scale, Y, yrs, ybs, n_rows, feat);
    return sgp::check_launch("bcast_rows");
}

int sgp_node_mean_bcast_f32(const float* X, int64_t xrs, int64_t xbs,
                            float* Y, int64_t yrs, int64_t ybs, float* partial,
                            int32_t n_rows, int32_t batch, int32_t feat, sgp_stream_t stream) {
    SGP_REQUIRE(X && (Y || partial), "sgp_node_mean_bcast_f32: null pointer");
    SGP_REQUIRE(n_rows >= 0 && batch >= 0 && feat >= 0 && batch <= 65535, "sgp_node_mean_bcast_f32: bad size");
    if (!batch || !feat) return 0;
    hipStream_t s = (hipStream_t)stream;
    // column sums go to `partial` if given, else are staged in the first row of the Y slot:
    // Y[b, 0, :] is overwritten by the broadcast afterwards, and X never aliases that slot.
    SGP_REQUIRE(partial != nullptr, "sgp_node_mean_bcast_f32: pass a [batch, feat] scratch as `partial`");
    // split rows across workgroups when the batch alone cannot fill the chip
    int nsplit = 1;
    const long long wgs = (long long)batch * ((feat + 255) / 256);
    if (wgs < 1024 && n_rows > 4096) {
        nsplit = (int)(2048 / (wgs < 1 ? 1 : wgs));
        if (nsplit > 64) nsplit = 64;
        if (nsplit < 1) nsplit = 1;
    }
    if (nsplit > 1) {
        hipError_t e = hipMemsetAsync(partial, 0, (size_t)batch * feat * sizeof(float), s);
        if (e != hipSuccess) return sgp::fail((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    if (n_rows > 0) {
        if (vec_ok(X, xrs, xbs, feat))
            hipLaunchKernelGGL(col_sum_kernel<4>, dim3((feat + 255) / 256, batch, nsplit), dim3(256), 0, s,
                               X, xrs, xbs, partial, n_rows, feat, 0);
        else
            hipLaunchKernelGGL(col_sum_kernel<1>, dim3((feat + 63) / 64, batch, nsplit), dim3(256), 0, s,
                               X, xrs, xbs, partial, n_rows, feat, 0);
        int rc = sgp::check_launch("col_sum");
        if (rc) return rc;
    }
    if (Y == nullptr) return 0;
    return sgp_bcast_rows_f32(partial, n_rows > 0 ? 1.0f / (float)n_rows : 0.f, Y, yrs, ybs, n_rows, batch,
                              feat, stream);
}

int sgp_copy_rows_f32(const float* X, int64_t xrs, int64_t xbs, float* Y, int64_t yrs, int64_t ybs,
                      int32_t n_rows, int32_t batch, int32_t feat, sgp_stream_t stream) {
    SGP_REQUIRE(X && Y, "sgp_copy_rows_f32: null pointer");
    SGP_REQUIRE(n_rows >= 0 && batch >= 0 && feat >= 0 && batch <= 65535, "sgp_copy_rows_f32: bad size");
    if (!n_rows || !batch || !feat) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (vec_ok(X, xrs, xbs, feat) && vec_ok(Y, yrs, ybs, feat))
        hipLaunchKernelGGL(copy_rows_kernel<4>, dim3(grid_for((long long)n_rows * feat / 4), batch), dim3(256),
                           0, s, X, xrs, xbs, Y, yrs, ybs, n_rows, feat);
    else
        hipLaunchKernelGGL(copy_rows_kernel<1>, dim3(grid_for((long long)n_rows * feat), batch), dim3(256),
                           0, s, X, xrs, xbs, Y, yrs, ybs, n_rows, feat);
    return sgp::check_launch("copy_rows");
}

int sgp_gather_rows_f32(const float* X, int64_t xrs, int64_t xbs,
                        const int32_t* step, const int32_t* node, int32_t n_index,
                        float* out, int64_t ors, int64_t obs,
                        int32_t batch, int32_t feat, sgp_stream_t stream) {
    SGP_REQUIRE(X && node && out, "sgp_gather_rows_f32: null pointer");
    SGP_REQUIRE(n_index >= 0 && batch >= 0 && feat >= 0 && batch <= 65535, "sgp_gather_rows_f32: bad size");
    if (step) batch = 1;
    if (!n_index || !batch || !feat) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (vec_ok(X, xrs, xbs, feat) && vec_ok(out, ors, obs, feat))
        hipLaunchKernelGGL(gather_rows_kernel<4>, dim3(grid_for((long long)n_index * feat / 4), batch),
                           dim3(256), 0, s, X, xrs, xbs, step, node, n_index, out, ors, obs, feat);
    else
        hipLaunchKernelGGL(gather_rows_kernel<1>, dim3(grid_for((long long)n_index * feat), batch), dim3(256),
                           0, s, X, xrs, xbs, step, node, n_index, out, ors, obs, feat);
    return sgp::check_launch("gather_rows");
}

int sgp_abs_max_f32(const float* X, int64_t xrs, int64_t xbs, int32_t n_rows, int32_t batch, int32_t feat,
                    float* out, sgp_stream_t stream) {
    SGP_REQUIRE(out, "sgp_abs_max_f32: null pointer");
    SGP_REQUIRE(n_rows >= 0 && batch >= 0 && feat >= 0, "sgp_abs_max_f32: bad size");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, 4, s);
    if (e != hipSuccess) return sgp::fail((int)e, "sgp_abs_max_f32: %s", hipGetErrorString(e));
    if (!n_rows || !batch || !feat) return 0;
    SGP_REQUIRE(X, "sgp_abs_max_f32: null pointer");
    const bool vec = (feat & 3) == 0 && sgp::aligned16(X) && xrs % 4 == 0 && xbs % 4 == 0;
    SGP_REQUIRE(vec || (feat & 3) != 0, "sgp_abs_max_f32: widths that are multiples of 4 need 16-byte aligned rows");
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        long long per = ((long long)n_rows * feat / (vec ? 4 : 1) + 256 * 8 - 1) / (256 * 8);
        int gx = (int)(per < 1 ? 1 : per > 1024 ? 1024 : per);
        hipLaunchKernelGGL(abs_max_kernel, dim3(gx, nb), dim3(256), 0, s, X + (long long)b0 * xbs, xrs, xbs,
                               n_rows, nb, feat, (unsigned*)out);
    }
    return sgp::check_launch("sgp_abs_max_f32");
}

}  // extern "C"
