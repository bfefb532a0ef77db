// First layer of the SGP decoder ("next" row f4): the grouped 1x1 convolution of
// lib/nn/models/sgp_model.py:41-52 (Rearrange 'b n f -> b f n', nn.Conv1d(input_size,
// out_channels, kernel_size=1, groups=order), Rearrange back, activation), i.e. a block-diagonal
// linear map applied to every (b, n) row:
//
//     y[row, g*oc + o] = act( bias[g*oc + o] + sum_i W[g*oc + o, i] * x[row, g*ic + i] )
//
// optionally fused with the IID gather of row f1 (row k = X[step[k], node[k], :], the batch never
// exists in HBM).  gfx950: one wave per 16 rows; the contraction runs transposed on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products): D[j, row] += W[j, k] * x[row, k],
// k order inside every 16-block permuted to (4 q + s) so that a lane's 16-byte load of its row
// IS the B operand of 4 consecutive MFMAs (the same trick as the reservoir kernel).
#include "common.h"

using sgp::f32x4;

namespace {

// packed weights: Wp[g][jt][kb][lane][s] = W[g*oc + 16 jt + (l & 15)][16 kb + 4 (l >> 4) + s]
__host__ __device__ inline long long packed_floats(int groups, int ic, int oc) {
    const long long JT = (oc + 15) / 16, KB = (ic + 15) / 16;
    return (long long)groups * JT * KB * 256;
}

__global__ void pack_grouped(const float* __restrict__ w, float* __restrict__ out,
                             int groups, int ic, int oc) {
    const int JT = (oc + 15) / 16, KB = (ic + 15) / 16;
    const long long total = packed_floats(groups, ic, oc);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(i & 3), l = (int)((i >> 2) & 63);
        long long r = i >> 8;
        const int kb = (int)(r % KB); r /= KB;
        const int jt = (int)(r % JT);
        const int g = (int)(r / JT);
        const int j = 16 * jt + (l & 15), k = 16 * kb + 4 * (l >> 4) + s;
        out[i] = (j < oc && k < ic) ? w[((long long)g * oc + j) * ic + k] : 0.f;
    }
}

struct GlArgs {
    const float* x; long long xrs, xbs;
    const int* step; const int* node;
    const float* wp; const float* bias;
    float* out; long long ors;
    int n_rows, groups, ic, oc, act;
    bool xvec;
};

__device__ __forceinline__ float activate(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);                                   // relu
    if (act == 2) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));     // silu = x * sigmoid(x)
    return v;
}

// One wave = 16 rows x one group (blockIdx.y).  JTC output tiles (16 channels each) are accumulated
// together so that every row piece is loaded once; the row pieces of 8 k-blocks are requested
// before the first MFMA consumes one (a wave's chain is one memory latency, not one per k-block).
template <int JTC>
__global__ __launch_bounds__(64) void grouped_linear(GlArgs a) {
    constexpr int KC = 8;
    const int lane = threadIdx.x & 63;
    const int b = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * 16 + b;
    const int g = blockIdx.y;
    const bool ok = row < a.n_rows;
    const float* xp = a.x;
    if (ok) {
        if (a.step) xp += (long long)a.step[row] * a.xbs + (long long)a.node[row] * a.xrs;
        else xp += (long long)row * a.xrs;
    }
    xp += (long long)g * a.ic;
    const int JT = (a.oc + 15) / 16, KB = (a.ic + 15) / 16;
    for (int jt0 = 0; jt0 < JT; jt0 += JTC) {
        f32x4 acc[JTC];
#pragma unroll
        for (int c = 0; c < JTC; ++c) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * (jt0 + c) + 4 * q + r;
                acc[c][r] = (jt0 + c < JT && j < a.oc) ? a.bias[g * a.oc + j] : 0.f;
            }
        }
        for (int kb0 = 0; kb0 < KB; kb0 += KC) {
            f32x4 xv[KC];
#pragma unroll
            for (int u = 0; u < KC; ++u) {
                const int k = 16 * (kb0 + u) + 4 * q;
                xv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok && kb0 + u < KB) {
                    const float* p = xp + k;
                    if (a.xvec) {
                        if (k < a.ic) xv[u] = *reinterpret_cast<const f32x4*>(p);
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            if (k + s < a.ic) xv[u][s] = p[s];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < KC; ++u) {
                if (kb0 + u < KB) {                                       // wave-uniform
#pragma unroll
                    for (int c = 0; c < JTC; ++c) {
                        if (jt0 + c < JT) {
                            const f32x4 wf = *reinterpret_cast<const f32x4*>(
                                a.wp + ((((long long)g * JT + jt0 + c) * KB + kb0 + u) * 64 + lane) * 4);
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xv[u][s], acc[c], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (ok) {
#pragma unroll
            for (int c = 0; c < JTC; ++c) {
                if (jt0 + c < JT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * (jt0 + c) + 4 * q + r;
                        if (j < a.oc)
                            a.out[(long long)row * a.ors + g * a.oc + j] = activate(acc[c][r], a.act);
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" {

int64_t sgp_grouped_linear_packed_floats(int32_t groups, int32_t ic, int32_t oc) {
    if (groups <= 0 || ic <= 0 || oc <= 0) return -1;
    return packed_floats(groups, ic, oc);
}

int sgp_grouped_linear_pack_f32(const float* w, float* packed, int32_t groups, int32_t ic, int32_t oc,
                                sgp_stream_t stream) {
    SGP_REQUIRE(w && packed, "sgp_grouped_linear_pack_f32: null pointer");
    SGP_REQUIRE(groups > 0 && ic > 0 && oc > 0, "sgp_grouped_linear_pack_f32: bad size");
    const long long total = packed_floats(groups, ic, oc);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pack_grouped, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, packed, groups, ic, oc);
    return sgp::check_launch("pack_grouped");
}

int sgp_grouped_linear_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                           const int32_t* step, const int32_t* node,
                           const float* w_packed, const float* bias, int32_t act,
                           float* out, int64_t out_row_stride,
                           int32_t n_rows, int32_t groups, int32_t ic, int32_t oc,
                           sgp_stream_t stream) {
    SGP_REQUIRE(X && w_packed && bias && out, "sgp_grouped_linear_f32: null pointer");
    SGP_REQUIRE((step == nullptr) == (node == nullptr), "sgp_grouped_linear_f32: step and node go together");
    SGP_REQUIRE(n_rows >= 0 && groups > 0 && ic > 0 && oc > 0, "sgp_grouped_linear_f32: bad size");
    SGP_REQUIRE(act >= 0 && act <= 2, "sgp_grouped_linear_f32: unknown activation %d", act);
    SGP_REQUIRE(sgp::aligned16(w_packed), "sgp_grouped_linear_f32: packed weights must be 16-byte aligned");
    if (n_rows == 0) return 0;
    GlArgs a;
    a.x = X; a.xrs = x_row_stride; a.xbs = x_batch_stride;
    a.step = step; a.node = node;
    a.wp = w_packed; a.bias = bias;
    a.out = out; a.ors = out_row_stride;
    a.n_rows = n_rows; a.groups = groups; a.ic = ic; a.oc = oc; a.act = act;
    a.xvec = ic % 4 == 0 && x_row_stride % 4 == 0 && x_batch_stride % 4 == 0 && sgp::aligned16(X);
    const int grid = (n_rows + 15) / 16;
    const int JT = (oc + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    SGP_REQUIRE(groups <= 65535, "sgp_grouped_linear_f32: more than 65535 groups");
    if (JT == 1) hipLaunchKernelGGL(grouped_linear<1>, dim3(grid, groups), dim3(64), 0, s, a);
    else if (JT == 2) hipLaunchKernelGGL(grouped_linear<2>, dim3(grid, groups), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(grouped_linear<4>, dim3(grid, groups), dim3(64), 0, s, a);
    return sgp::check_launch("grouped_linear");
}

}  // extern "C"
