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
    float* pre;                       // [n_rows, groups*oc] pre-activation values for the backward pass, or null
    int n_rows, groups, ic, oc, act;
    bool xvec;
    unsigned drop_thresh;             // Dropout(p) after the activation (sgp_model.py:50): element kept when its
    unsigned seed_lo, seed_hi;        // Philox word >= drop_thresh = p * 2^32, scaled by keep_scale = 1 / (1 - p);
    float keep_scale;                 // drop_thresh == 0: no dropout (eval mode / p = 0)
};

// Philox4x32-10 keyed by the call's seed, counter = flat element index (row * width + column): the
// backward pass recomputes the mask from (seed, index) instead of storing it.
__device__ __forceinline__ unsigned philox_word(unsigned long long idx, unsigned k0, unsigned k1) {
    unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = 0x53475021u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__device__ __forceinline__ float keep_factor(unsigned long long idx, unsigned thresh, unsigned k0, unsigned k1, float scale) {
    if (thresh == 0u) return 1.f;
    return philox_word(idx, k0, k1) >= thresh ? scale : 0.f;
}

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
                        if (j < a.oc) {
                            const long long e = (long long)row * a.groups * a.oc + g * a.oc + j;
                            a.out[(long long)row * a.ors + g * a.oc + j] =
                                activate(acc[c][r], a.act) * keep_factor(e, a.drop_thresh, a.seed_lo, a.seed_hi, a.keep_scale);
                            if (a.pre) a.pre[e] = acc[c][r];
                        }
                    }
                }
            }
        }
    }
}

// ---- backward of the layer (the decoder is trained: lib/nn/models/sgp_model.py:41-52 sits under
// the Lightning optimiser loop).  With z = W x + b, y = act(z) and the incoming gradient dy:
//   dz = dy * keep * act'(z)               (dact_kernel; z is the `pre` output of the forward, keep the
//                                           recomputed dropout factor)
//   dx[row, g*ic + i] = sum_o dz[row, g*oc + o] W[g*oc + o, i]     = the forward kernel on dz with W^T
//                                                                    packed as a [groups*ic, oc] weight
//   dW[g*oc + o, i]   = sum_row dz[row, g*oc + o] x[row, g*ic + i] (wgrad_kernel below)
//   db[g*oc + o]      = sum_row dz[row, g*oc + o]                  (column sums: sgp_node_mean_bcast_f32)
__device__ __forceinline__ float dactivate(float z, int act) {
    if (act == 1) return z > 0.f ? 1.f : 0.f;
    if (act == 2) { const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-z)); return sg * (1.f + z * (1.f - sg)); }
    return 1.f;
}

__global__ __launch_bounds__(256) void dact_kernel(const float* __restrict__ dy, long long dyrs,
                                                   const float* __restrict__ pre, float* __restrict__ dz,
                                                   long long n_rows, int width, int act,
                                                   unsigned thresh, unsigned k0, unsigned k1, float scale) {
    const long long total = n_rows * width;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / width;
        const int c = (int)(i % width);
        dz[i] = dy[r * dyrs + c] * dactivate(pre[i], act) * keep_factor(i, thresh, k0, k1, scale);
    }
}

// W [groups*oc, ic] -> W^T as the weight [groups*ic, oc] of a grouped layer with ic and oc swapped
__global__ void transpose_grouped(const float* __restrict__ w, float* __restrict__ wt, int groups, int ic, int oc) {
    const long long total = (long long)groups * oc * ic;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int o = (int)(i % oc);
        long long r = i / oc;
        const int k = (int)(r % ic);
        const int g = (int)(r / ic);
        wt[i] = w[((long long)g * oc + o) * ic + k];
    }
}

// dW tile (16 output channels x 16 input channels of one group) over a slice of the rows: one wave,
// v_mfma_f32_16x16x4_f32 with the ROWS as the contraction index: D[o, i] += dz[row, o] * x[row, i],
// A operand lane (k, o) = dz[row0 + k][o], B operand lane (k, i) = x[row0 + k][i] (64-byte row pieces,
// 4 rows per instruction); slices meet in dW through float atomics (dW is zeroed first).
struct WgArgs {
    const float* x; long long xrs, xbs;
    const int* step; const int* node;
    const float* dz;                  // [n_rows, groups*oc]
    float* dw;                        // [groups*oc, ic]
    int n_rows, groups, ic, oc, rows_per_slice;
};

__global__ __launch_bounds__(64) void wgrad_kernel(WgArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, k = lane >> 4;
    const int KB = (a.ic + 15) / 16, JT = (a.oc + 15) / 16;
    const int tile = blockIdx.x % (KB * JT), g = blockIdx.x / (KB * JT);
    const int jt = tile / KB, kb = tile % KB;
    const int o = 16 * jt + c, i = 16 * kb + c;
    const int r0 = blockIdx.y * a.rows_per_slice;
    const int r1 = min(a.n_rows, r0 + a.rows_per_slice);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; r += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = r + 4 * u + k;
            const bool ok = row < r1;
            av[u] = (ok && o < a.oc) ? a.dz[(long long)row * a.groups * a.oc + g * a.oc + o] : 0.f;
            float xv = 0.f;
            if (ok && i < a.ic) {
                const float* xp = a.step ? a.x + (long long)a.step[row] * a.xbs + (long long)a.node[row] * a.xrs
                                         : a.x + (long long)row * a.xrs;
                xv = xp[g * a.ic + i];
            }
            bv[u] = xv;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
    // D: lane (q = k, j = c), register r -> (o = 16 jt + 4 q + r, i = 16 kb + j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oo = 16 * jt + 4 * k + r;
        if (oo < a.oc && i < a.ic) atomicAdd(a.dw + ((long long)g * a.oc + oo) * a.ic + i, acc[r]);
    }
}

}  // namespace

static void set_dropout(unsigned& thresh, unsigned& k0, unsigned& k1, float& scale, double p, uint64_t seed) {
    double t = p * 4294967296.0;
    thresh = p > 0.0 ? (unsigned)(t < 1.0 ? 1.0 : (t > 4294967295.0 ? 4294967295.0 : t)) : 0u;
    k0 = (unsigned)seed; k1 = (unsigned)(seed >> 32);
    scale = (float)(1.0 / (1.0 - p));
}

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
    return sgp_grouped_linear_fwd_f32(X, x_row_stride, x_batch_stride, step, node, w_packed, bias, act, out,
                                      out_row_stride, nullptr, 0.0, 0, n_rows, groups, ic, oc, stream);
}

int sgp_grouped_linear_fwd_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                               const int32_t* step, const int32_t* node,
                               const float* w_packed, const float* bias, int32_t act,
                               float* out, int64_t out_row_stride, float* pre,
                               double dropout_p, uint64_t seed,
                               int32_t n_rows, int32_t groups, int32_t ic, int32_t oc,
                               sgp_stream_t stream) {
    SGP_REQUIRE(X && w_packed && bias && out, "sgp_grouped_linear_f32: null pointer");
    SGP_REQUIRE(dropout_p >= 0.0 && dropout_p < 1.0, "sgp_grouped_linear_fwd_f32: dropout_p must lie in [0, 1)");
    SGP_REQUIRE((step == nullptr) == (node == nullptr), "sgp_grouped_linear_f32: step and node go together");
    SGP_REQUIRE(n_rows >= 0 && groups > 0 && ic > 0 && oc > 0, "sgp_grouped_linear_f32: bad size");
    SGP_REQUIRE(act >= 0 && act <= 2, "sgp_grouped_linear_f32: unknown activation %d", act);
    SGP_REQUIRE(sgp::aligned16(w_packed), "sgp_grouped_linear_f32: packed weights must be 16-byte aligned");
    if (n_rows == 0) return 0;
    GlArgs a;
    a.x = X; a.xrs = x_row_stride; a.xbs = x_batch_stride;
    a.step = step; a.node = node;
    a.wp = w_packed; a.bias = bias;
    a.out = out; a.ors = out_row_stride; a.pre = pre;
    set_dropout(a.drop_thresh, a.seed_lo, a.seed_hi, a.keep_scale, dropout_p, seed);
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

int sgp_grouped_linear_dact_f32(const float* dy, int64_t dy_row_stride, const float* pre, int32_t act,
                                double dropout_p, uint64_t seed,
                                float* dz, int64_t n_rows, int32_t width, sgp_stream_t stream) {
    SGP_REQUIRE(dy && pre && dz, "sgp_grouped_linear_dact_f32: null pointer");
    SGP_REQUIRE(dropout_p >= 0.0 && dropout_p < 1.0, "sgp_grouped_linear_dact_f32: dropout_p must lie in [0, 1)");
    unsigned thresh, k0, k1; float scale;
    set_dropout(thresh, k0, k1, scale, dropout_p, seed);
    SGP_REQUIRE(n_rows >= 0 && width > 0 && act >= 0 && act <= 2, "sgp_grouped_linear_dact_f32: bad argument");
    if (n_rows == 0) return 0;
    long long grid = (n_rows * width + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(dact_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, dy, dy_row_stride, pre, dz,
                       (long long)n_rows, width, act, thresh, k0, k1, scale);
    return sgp::check_launch("dact");
}

int sgp_grouped_linear_transpose_f32(const float* w, float* wt, int32_t groups, int32_t ic, int32_t oc,
                                     sgp_stream_t stream) {
    SGP_REQUIRE(w && wt, "sgp_grouped_linear_transpose_f32: null pointer");
    SGP_REQUIRE(groups > 0 && ic > 0 && oc > 0, "sgp_grouped_linear_transpose_f32: bad size");
    long long grid = ((long long)groups * ic * oc + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(transpose_grouped, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w, wt, groups, ic, oc);
    return sgp::check_launch("transpose_grouped");
}

int sgp_grouped_linear_wgrad_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                                 const int32_t* step, const int32_t* node,
                                 const float* dz, float* dw,
                                 int32_t n_rows, int32_t groups, int32_t ic, int32_t oc,
                                 sgp_stream_t stream) {
    SGP_REQUIRE(X && dz && dw, "sgp_grouped_linear_wgrad_f32: null pointer");
    SGP_REQUIRE((step == nullptr) == (node == nullptr), "sgp_grouped_linear_wgrad_f32: step and node go together");
    SGP_REQUIRE(n_rows >= 0 && groups > 0 && ic > 0 && oc > 0, "sgp_grouped_linear_wgrad_f32: bad size");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(dw, 0, (size_t)groups * oc * ic * sizeof(float), s);
    if (e != hipSuccess) return sgp::fail((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    if (n_rows == 0) return 0;
    WgArgs a;
    a.x = X; a.xrs = x_row_stride; a.xbs = x_batch_stride; a.step = step; a.node = node;
    a.dz = dz; a.dw = dw; a.n_rows = n_rows; a.groups = groups; a.ic = ic; a.oc = oc;
    const long long tiles = (long long)groups * ((ic + 15) / 16) * ((oc + 15) / 16);
    SGP_REQUIRE(tiles < (1ll << 31), "sgp_grouped_linear_wgrad_f32: too many tiles");
    // enough row slices to fill the chip, each at least 64 rows (a multiple of 16)
    long long slices = 4096 / (tiles < 1 ? 1 : tiles);
    if (slices < 1) slices = 1;
    long long rps = (n_rows + slices - 1) / slices;
    if (rps < 64) rps = 64;
    rps = (rps + 15) / 16 * 16;
    a.rows_per_slice = (int)rps;
    const long long ny = (n_rows + rps - 1) / rps;
    SGP_REQUIRE(ny <= 65535, "sgp_grouped_linear_wgrad_f32: too many row slices");
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)tiles, (unsigned)ny), dim3(64), 0, s, a);
    return sgp::check_launch("grouped_linear_wgrad");
}

}  // extern "C"
