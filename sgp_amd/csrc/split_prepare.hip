// Operand profile of the split-fp16 hop (spmm_split.hip): per-column statistics of x, and from them the per-column
// power-of-two scales, the bound of the NEXT hop's operand and the device flag that admits the split kernel or sends
// the hop to the exact-fp32 kernels (launch predicate, common.h) -- all on the device, no host round trip.
// Reference semantics being protected: lib/sgp_preprocessing.py:200-203 is plain fp32 for ANY x
// (lib/nn/models/sgp_model.py:169-181 feeds it raw [B, N, F] batches; reservoir.py:37-41 allows relu).
//
// Error model of the split kernel for a column with bound B >= max |x[:, c]| (scale 2^e puts B at 2^13 .. 2^14):
//   |x| >= 2^-16 B : relative 2^-23 (two normal fp16 pieces, 22 bits)      -- as good as fp32
//   |x| <  2^-16 B : absolute <= 2^-38 B (the low piece is an fp16 subnormal)
// The kernel is admitted when the absolute term cannot exceed 2^-22 of the column's RMS:
//   B <= 2^16 * rms(x[:, c])        (checked as B * sqrt(s_eff) <= 2^16 * rms_sampled, see below)
// for every column that is not identically zero (bound 0: exact).  Columns whose bound is far above their data
// (outliers, a tiny input_scaling under the a-priori bound 1 of a tanh reservoir, raw features of mixed units) fail
// the test and the hop runs on the exact kernels instead.
#include "common.h"

namespace {

// per-column max |x| (bit pattern, atomicMax: IEEE order = integer order for non-negative floats, NaN patterns win)
// and sum of squares over the steps b = 0, t_stride, 2 t_stride, ...; feat % 4 == 0.
__global__ __launch_bounds__(256) void col_stats_kernel(const float* x, long long xrs, long long xbs, int n_rows,
                                                        int t_stride, int feat, unsigned* amax, float* ssq) {
    // (rows: the caller passes n_rows / r_stride and xrs * r_stride to read every r_stride-th row)
    const int q = feat >> 2;                                  // 16-byte pieces per row
    const int rpi = 256 / q;                                  // rows per pass of the block (q <= 256)
    const int r0 = threadIdx.x / q, c = threadIdx.x - r0 * q;
    const float* xb = x + (long long)blockIdx.y * t_stride * xbs;
    unsigned m[4] = {0, 0, 0, 0};
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (r0 < rpi) {
        auto take = [&](const float4 v) {
            m[0] = max(m[0], __float_as_uint(fabsf(v.x))); s[0] = fmaf(v.x, v.x, s[0]);
            m[1] = max(m[1], __float_as_uint(fabsf(v.y))); s[1] = fmaf(v.y, v.y, s[1]);
            m[2] = max(m[2], __float_as_uint(fabsf(v.z))); s[2] = fmaf(v.z, v.z, s[2]);
            m[3] = max(m[3], __float_as_uint(fabsf(v.w))); s[3] = fmaf(v.w, v.w, s[3]);
        };
        const long long step = (long long)gridDim.x * rpi;
        long long r = (long long)blockIdx.x * rpi + r0;
        for (; r + 3 * step < n_rows; r += 4 * step) {          // four rows in flight per thread
            const float4 v0 = *(const float4*)(xb + r * xrs + 4 * c);
            const float4 v1 = *(const float4*)(xb + (r + step) * xrs + 4 * c);
            const float4 v2 = *(const float4*)(xb + (r + 2 * step) * xrs + 4 * c);
            const float4 v3 = *(const float4*)(xb + (r + 3 * step) * xrs + 4 * c);
            take(v0); take(v1); take(v2); take(v3);
        }
        for (; r < n_rows; r += step) take(*(const float4*)(xb + r * xrs + 4 * c));
    }
    __shared__ unsigned pm[256][4];
    __shared__ float ps[256][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { pm[threadIdx.x][i] = m[i]; ps[threadIdx.x][i] = s[i]; }
    __syncthreads();
    if (threadIdx.x < q) {
        for (int r = 1; r < rpi; ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { m[i] = max(m[i], pm[r * q + c][i]); s[i] += ps[r * q + c][i]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { atomicMax(amax + 4 * c + i, m[i]); atomicAdd(ssq + 4 * c + i, s[i]); }
    }
}

struct PrepArgs {
    const unsigned* amax; const float* ssq;   // statistics (null: no guard, the caller vouches for the bound)
    double n_samples, s_eff;                  // rows behind ssq per column; rows of the operand / rows sampled
    int full;                                 // statistics cover every row: amax is the exact maximum (a measured bound needs it)
    const float* bound_in; float bound_scalar;// per-column bound (device) or one bound for all (> 0), else the measured maximum
    float norm_inf;
    int feat;
    float* tab; float* bound_out; int* flag;
};

__global__ __launch_bounds__(256) void split_prepare_kernel(PrepArgs a) {
    int ok = 1;
    for (int c = threadIdx.x; c < a.feat; c += 256) {
        float b;
        const float mx = a.amax ? __uint_as_float(a.amax[c]) : 0.f;
        if (a.bound_in) b = a.bound_in[c];
        else if (a.bound_scalar > 0.f) b = a.bound_scalar;
        else b = mx;                                           // (host: only with full statistics)
        if (a.amax && !(mx <= b)) ok = 0;                       // the data exceed the bound (or are NaN): the bound was wrong
        float scale = 1.f, inv = 1.f;
        if (!(b >= 0.f) || !(b < __builtin_inff())) { ok = 0; b = __builtin_inff(); }
        else if (b > 0.f) {
            int k;
            const float mant = frexpf(b, &k);                   // b = mant * 2^k, mant in [0.5, 1)
            int e = (mant == 0.5f ? 15 : 14) - k;               // floor(log2(16384 / b))
            e = min(126, max(-126, e));
            scale = ldexpf(1.f, e); inv = ldexpf(1.f, -e);
            if (a.amax) {
                const double rms = sqrt((double)a.ssq[c] / a.n_samples);
                // absolute error <= 2^-38 b must stay below 2^-22 rms; sampling every s_eff-th row can overstate
                // the mean square by at most s_eff
                if (!((double)b * sqrt(a.s_eff) <= 65536.0 * rms)) ok = 0;
            }
        }
        a.tab[c] = scale; a.tab[a.feat + c] = inv;
        if (a.bound_out) a.bound_out[c] = b * a.norm_inf * (1.f + 1e-6f);
    }
    ok = __syncthreads_and(ok);
    if (threadIdx.x == 0) a.flag[0] = ok;
}

}  // namespace

extern "C" {

int sgp_col_stats_f32(const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                      int32_t n_rows, int32_t batch, int32_t feat, int32_t t_stride, int32_t r_stride, int32_t accumulate,
                      float* stats, sgp_stream_t stream) {
    SGP_REQUIRE(X && stats, "sgp_col_stats_f32: null pointer");
    SGP_REQUIRE(n_rows >= 0 && batch >= 0 && feat > 0 && feat % 4 == 0 && feat <= 1024 && t_stride >= 1 && r_stride >= 1,
                "sgp_col_stats_f32: feat must be a multiple of 4 up to 1024, strides >= 1");
    n_rows = (n_rows + r_stride - 1) / r_stride;              // rows 0, r_stride, 2 r_stride, ..
    x_row_stride *= r_stride;
    SGP_REQUIRE(sgp::aligned16(X) && x_row_stride % 4 == 0 && x_batch_stride % 4 == 0, "sgp_col_stats_f32: rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(stats, 0, 2 * (size_t)feat * sizeof(float), s);
        if (e != hipSuccess) return sgp::fail((int)e, "sgp_col_stats_f32: memset: %s", hipGetErrorString(e));
    }
    if (n_rows == 0 || batch == 0) return 0;
    const int ns = (batch + t_stride - 1) / t_stride;
    SGP_REQUIRE(ns <= 65535, "sgp_col_stats_f32: more than 65535 sampled steps");
    const int rpi = 256 / (feat / 4);
    long long gx = ((long long)n_rows + rpi - 1) / rpi;
    const long long cap = 512 / ns > 0 ? 512 / ns : 1;         // ~two workgroups per CU: every one ends with 2 x feat atomics on the same
                                                               // 2 x feat words (2048 workgroups spent 0.15 ms queueing on them)
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(col_stats_kernel, dim3((unsigned)gx, (unsigned)ns), dim3(256), 0, s,
                       X, (long long)x_row_stride, (long long)x_batch_stride, n_rows, t_stride, feat,
                       (unsigned*)stats, stats + feat);
    return sgp::check_launch("col_stats");
}

int sgp_split_prepare_f32(const float* stats, double n_samples, double s_eff, int32_t full,
                          const float* bound_in, float bound_scalar, float norm_inf, int32_t feat,
                          float* x_tab, float* bound_out, int32_t* flag, sgp_stream_t stream) {
    SGP_REQUIRE(x_tab && flag, "sgp_split_prepare_f32: null pointer");
    SGP_REQUIRE(feat > 0 && feat <= 1024, "sgp_split_prepare_f32: feat out of range");
    SGP_REQUIRE(stats || bound_in || bound_scalar > 0.f, "sgp_split_prepare_f32: neither statistics nor a bound");
    SGP_REQUIRE(!stats || (n_samples >= 1.0 && s_eff >= 1.0), "sgp_split_prepare_f32: statistics need their sample counts");
    SGP_REQUIRE(stats || full == 0, "sgp_split_prepare_f32: full = 1 needs statistics");
    SGP_REQUIRE(bound_in || bound_scalar > 0.f || full, "sgp_split_prepare_f32: a measured bound needs statistics of every row");
    PrepArgs a;
    a.amax = (const unsigned*)stats; a.ssq = stats ? stats + feat : nullptr;
    a.n_samples = n_samples; a.s_eff = s_eff; a.full = full;
    a.bound_in = bound_in; a.bound_scalar = bound_scalar; a.norm_inf = norm_inf; a.feat = feat;
    a.tab = x_tab; a.bound_out = bound_out; a.flag = flag;
    hipLaunchKernelGGL(split_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return sgp::check_launch("split_prepare");
}

}  // extern "C"
