// DynGESN step (reference: lib/nn/reservoir/graph_reservoir.py:85-93, driven per time step by
// tsl/nn/blocks/encoders/gcrnn.py:67-93):
//     h' = (1 - a) h + a * act( x W_ih^T + b + A_hat (h W_hh^T) )
// Unlike the SGP encoder the graph product sits INSIDE the recurrence, so every (step, layer)
// is a global dependency: z = h W_hh^T for all nodes, then the SpMM, then the update.  Two
// small kernels per (step, layer): a dense fp32 GEMM on the matrix cores and a fused
// "SpMM + bias term + activation + leak" row kernel.  Graphs of this baseline are small
// (METR-LA 207 nodes, PEMS-BAY 325), so the kernels are latency-, not throughput-oriented.
#include "common.h"

using sgp::f32x4;

// gesn_persist.hip
namespace sgp_gesn {
struct PArgs;
int mode();
long long packed_floats(int R, int L);
int pack(const float* wcat, float* wpk, int R, int L, hipStream_t stream);
int run_chunk(const int32_t* rowptr, const int32_t* col, const float* val, const float* p0,
              const float* wcat, const float* wpk, const float* bcat, float* cbuf, float* h_state, float* out,
              long long ors, long long oss, unsigned* bar, const double* alpha, int act,
              int tc, int N, int R, int L, hipStream_t stream);
}

namespace {

// C[m, n] = sum_k A[m, k] * W[n, k] (+ bias[n]);  one 256-thread block per 16 x 16 output tile,
// v_mfma_f32_16x16x4_f32 (exact fp32), K split over the 4 waves and reduced through LDS.
// Operand lane (i = l & 15, q = l >> 4) holds row m0 + i (A) / n0 + i (W) at contraction index
// "q": VEC path -- one float4 per 16 k (index 16 c + 4 q + s feeds MFMA s of chunk c, the same
// bijection on both operands); scalar path for ragged K / unaligned rows.
// D: lane (j = l & 15, q), reg r -> C[m0 + 4 q + r][n0 + j].
template <bool VEC>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, long long lda,
                                                      const float* __restrict__ W, long long ldw,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ C, long long ldc,
                                                      int M, int N, int K) {
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
    const int am = min(m0 + i, M - 1), wn = min(n0 + i, N - 1);   // clamped rows are never stored
    const float* __restrict__ arow = A + (long long)am * lda;
    const float* __restrict__ wrow = W + (long long)wn * ldw;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC) {
        const int n_chunks = K >> 4;
        constexpr int U = 5;                         // chunks in flight: 10 float4 loads
        for (int c0 = wave; c0 < n_chunks; c0 += 4 * U) {
            f32x4 a[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = min(c0 + 4 * u, n_chunks - 1);
                a[u] = *reinterpret_cast<const f32x4*>(arow + 16 * c + 4 * q);
                w[u] = *reinterpret_cast<const f32x4*>(wrow + 16 * c + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c0 + 4 * u < n_chunks) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], w[u][s], acc, 0, 0, 0);
                }
            }
        }
    } else {
        for (int k0 = 4 * wave; k0 < K; k0 += 16) {
            const int k = k0 + q;
            const float a = k < K ? arow[k] : 0.f;
            const float w = k < K ? wrow[k] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w, acc, 0, 0, 0);
        }
    }
    if (wave) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const f32x4 o = red[w][lane];
        acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
    }
    const int n = n0 + i;
    if (n < N) {
        const float b = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * q + r;
            if (m < M) C[(long long)m * ldc + n] = acc[r] + b;
        }
    }
}

// one wave per node i:  h_out[i, f] = (1-a) h_in[i, f] + a * act(p[i, f] + sum_e val[e] z[col[e], f])
// also written to out_row[i * out_stride + f] (the [T, N, L*R] embedding slot of this step).
template <int NV>          // NV = ceil(R / 64) feature slices per lane
__global__ __launch_bounds__(64) void gesn_update_kernel(
        const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
        const float* __restrict__ z, long long ldz, const float* __restrict__ p, long long ldp,
        const float* __restrict__ h_in, float alpha, float one_minus_alpha, int act,
        float* __restrict__ h_out, float* __restrict__ out_row, long long out_stride,
        int n_nodes, int R) {
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float v[NV], hprev[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int f = c * 64 + lane;
        const bool ok = f < R;
        v[c] = ok ? p[(long long)i * ldp + f] : 0.f;
        hprev[c] = ok ? h_in[(long long)i * R + f] : 0.f;
    }
    // edges: 64 at a time into lanes, broadcast one by one; NV independent loads per edge
    for (int eb = e0; eb < e1; eb += 64) {
        const int ne = min(64, e1 - eb);
        const int my_col = lane < ne ? col[eb + lane] : 0;
        const float my_val = lane < ne ? val[eb + lane] : 0.f;
        for (int e = 0; e < ne; ++e) {
            const int cj = __builtin_amdgcn_readlane(my_col, e);
            const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_val), e));
            const float* __restrict__ zr = z + (long long)cj * ldz;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int f = c * 64 + lane;
                v[c] = fmaf(w, f < R ? zr[f] : 0.f, v[c]);
            }
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        if (act == SGP_ACT_TANH || act == SGP_ACT_TANH_REL) v[c] = tanhf(v[c]);
        else if (act == SGP_ACT_RELU) v[c] = fmaxf(v[c], 0.f);
        ss = fmaf(v[c], v[c], ss);               // lanes beyond R hold 0
    }
    float inv = 1.f;
    if (act == SGP_ACT_SELF_NORM) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    }
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int f = c * 64 + lane;
        if (f < R) {
            const float hn = one_minus_alpha * hprev[c] + alpha * (v[c] * inv);
            h_out[(long long)i * R + f] = hn;
            out_row[(long long)i * out_stride + f] = hn;
        }
    }
}

int launch_gemm(const float* A, long long lda, const float* W, long long ldw, const float* bias,
                float* C, long long ldc, int M, int N, int K, hipStream_t stream) {
    if (M == 0 || N == 0) return 0;
    const bool vec = K > 0 && K % 16 == 0 && lda % 4 == 0 && ldw % 4 == 0 &&
                     sgp::aligned16(A) && sgp::aligned16(W);
    const dim3 grid((M + 15) / 16, (N + 15) / 16), block(256);
    if (vec)
        hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, block, 0, stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, block, 0, stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
    return sgp::check_launch("gemm_nt");
}

int launch_update(const int32_t* rowptr, const int32_t* col, const float* val,
                  const float* z, long long ldz, const float* p, long long ldp, const float* h_in,
                  double alpha, int act, float* h_out, float* out_row, long long out_stride,
                  int n_nodes, int R, hipStream_t stream) {
#define SGP_GESN_LAUNCH(NV)                                                                     \
    hipLaunchKernelGGL(gesn_update_kernel<NV>, dim3(n_nodes), dim3(64), 0, stream, rowptr, col,  \
                       val, z, ldz, p, ldp, h_in, (float)alpha, (float)(1.0 - alpha), act,       \
                       h_out, out_row, out_stride, n_nodes, R)
    const int nv = (R + 63) / 64;
    if (nv <= 1) SGP_GESN_LAUNCH(1);
    else if (nv <= 2) SGP_GESN_LAUNCH(2);
    else if (nv <= 4) SGP_GESN_LAUNCH(4);
    else if (nv <= 6) SGP_GESN_LAUNCH(6);
    else SGP_GESN_LAUNCH(8);
#undef SGP_GESN_LAUNCH
    return sgp::check_launch("gesn_update");
}

constexpr int kStepChunk = 256;            // steps of layer-0 input term computed per GEMM

struct GesnWorkspace {                      // offsets in floats
    long long wcat, bcat, c, hb, p0, bar, wpk, total;
    GesnWorkspace(long long N, long long R, long long L) {
        wcat = 0;                           // L x [2R, R]   rows 0..R-1 = W_hh,i; R..2R-1 = W_ih,i+1
        bcat = wcat + L * 2 * R * R;        // L x [2R]      zeros | b_{i+1}
        c = bcat + L * 2 * R;               // 2 x L x [N, 2R]   z_i | p_{i+1}  (the persistent kernel
                                            // alternates two planes by tick; the stepwise path uses one)
        hb = c + 2 * L * N * 2 * R;         // [L, N, R]     ping-pong partner of h_state
        p0 = hb + L * N * R;                // [kStepChunk, N, R]
        bar = (p0 + (long long)kStepChunk * N * R + 3) / 4 * 4;   // grid barrier words
        wpk = bar + 512;                    // (kBarWords of gesn_persist.hip = 272); packed weight fragments
        total = wpk + (R % 16 == 0 && R <= 384 ? sgp_gesn::packed_floats((int)R, (int)L) : 0);
    }
};

}  // namespace


extern "C" {

int sgp_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                    float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, sgp_stream_t stream) {
    SGP_REQUIRE(M >= 0 && N >= 0 && K >= 0, "sgp_gemm_nt_f32: negative size");
    if (M == 0 || N == 0) return 0;
    SGP_REQUIRE(C && (K == 0 || (A && W)), "sgp_gemm_nt_f32: null pointer");
    SGP_REQUIRE((N + 15) / 16 <= 65535, "sgp_gemm_nt_f32: more than 1M output columns");
    return launch_gemm(A, lda, W, ldw, bias, C, ldc, M, N, K, (hipStream_t)stream);
}

int sgp_gesn_update_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                        const float* z, const float* p, const float* h_in,
                        double alpha, int32_t act,
                        float* h_out, float* out_row, int64_t out_stride,
                        int32_t n_nodes, int32_t R, sgp_stream_t stream) {
    SGP_REQUIRE(rowptr && col && val && z && p && h_in && h_out && out_row, "sgp_gesn_update_f32: null pointer");
    SGP_REQUIRE(n_nodes >= 0 && R > 0, "sgp_gesn_update_f32: bad size");
    SGP_REQUIRE(act >= SGP_ACT_TANH && act <= SGP_ACT_TANH_REL, "sgp_gesn_update_f32: unknown activation %d", act);
    if (R > 512) return sgp::fail(SGP_EUNSUP, "sgp_gesn_update_f32: reservoir size %d > 512 not supported", R);
    if (n_nodes == 0) return 0;
    return launch_update(rowptr, col, val, z, R, p, R, h_in, alpha, act, h_out, out_row, out_stride,
                         n_nodes, R, (hipStream_t)stream);
}

int64_t sgp_gesn_workspace_bytes(int32_t N, int32_t R, int32_t L) {
    if (N < 0 || R <= 0 || L <= 0) return 0;
    return GesnWorkspace(N, R, L).total * (int64_t)sizeof(float);
}

int sgp_gesn_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                 const float* x, int64_t x_row_stride, int64_t x_step_stride,
                 const float* const* w_ih, const float* const* w_hh, const float* const* b,
                 const double* alpha, int32_t act,
                 float* out, int64_t out_row_stride, int64_t out_step_stride,
                 float* h_state, void* workspace,
                 int32_t T, int32_t N, int32_t F, int32_t R, int32_t L, sgp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SGP_REQUIRE(T >= 0 && N >= 0 && F > 0 && R > 0 && L > 0, "sgp_gesn_f32: bad size");
    SGP_REQUIRE(act >= SGP_ACT_TANH && act <= SGP_ACT_TANH_REL, "sgp_gesn_f32: unknown activation %d", act);
    if (R > 512) return sgp::fail(SGP_EUNSUP, "sgp_gesn_f32: reservoir size %d > 512 not supported", R);
    if (T == 0 || N == 0) return 0;
    SGP_REQUIRE(rowptr && col && val && x && w_ih && w_hh && b && alpha && out && h_state && workspace,
                "sgp_gesn_f32: null pointer");
    SGP_REQUIRE(sgp::aligned16(workspace), "sgp_gesn_f32: workspace must be 16-byte aligned");
    for (int i = 0; i < L; ++i)
        SGP_REQUIRE(w_ih[i] && w_hh[i] && b[i], "sgp_gesn_f32: null weight pointer (layer %d)", i);
    const GesnWorkspace ws(N, R, L);
    float* base = static_cast<float*>(workspace);
    const size_t RR = (size_t)R * R;
    // concatenated weights: one GEMM on h_i(t) yields z_i (for step t+1) and p_{i+1} (for step t)
    hipError_t e = hipMemsetAsync(base + ws.bcat, 0, (size_t)L * 2 * R * sizeof(float), stream);
    for (int i = 0; i < L && e == hipSuccess; ++i) {
        float* wc = base + ws.wcat + (size_t)i * 2 * RR;
        e = hipMemcpyAsync(wc, w_hh[i], RR * sizeof(float), hipMemcpyDeviceToDevice, stream);
        if (i + 1 < L && e == hipSuccess)
            e = hipMemcpyAsync(wc + RR, w_ih[i + 1], RR * sizeof(float), hipMemcpyDeviceToDevice, stream);
        if (i + 1 < L && e == hipSuccess)
            e = hipMemcpyAsync(base + ws.bcat + (size_t)i * 2 * R + R, b[i + 1], (size_t)R * sizeof(float),
                               hipMemcpyDeviceToDevice, stream);
    }
    if (e != hipSuccess) return sgp::fail((int)e, "sgp_gesn_f32: weight staging: %s", hipGetErrorString(e));
    auto wcat = [&](int i) { return base + ws.wcat + (size_t)i * 2 * RR; };
    auto bcat = [&](int i) { return base + ws.bcat + (size_t)i * 2 * R; };
    auto cbuf = [&](int i) { return base + ws.c + (size_t)i * N * 2 * R; };
    auto n_out = [&](int i) { return i + 1 < L ? 2 * R : R; };
    const size_t NR = (size_t)N * R;
    const bool x_flat = x_step_stride == (int64_t)N * x_row_stride;
    int rc = 0;
    auto input_term = [&](int t0, int tc) {      // p_0 of steps t0 .. t0 + tc - 1 -> ws.p0
        float* p0 = base + ws.p0;
        if (x_flat)
            return launch_gemm(x + (size_t)t0 * x_step_stride, x_row_stride, w_ih[0], F, b[0], p0, R,
                               tc * N, R, F, stream);
        int r = 0;
        for (int s = 0; s < tc && !r; ++s)
            r = launch_gemm(x + (size_t)(t0 + s) * x_step_stride, x_row_stride, w_ih[0], F, b[0],
                            p0 + s * NR, R, N, R, F, stream);
        return r;
    };
    // ---- persistent path: one cooperative launch per chunk of steps (gesn_persist.hip); a refused
    // launch (shape not served, device busy with another cooperative kernel) leaves the rest of the
    // sequence to the stepwise path below, which restarts from h_state
    int t_done = 0;
    bool used_persistent = false;
    constexpr int kBarSticky = 400;           // (gesn_persist.hip) failure word outside the per-launch memset
    if (sgp_gesn::mode()) {
        e = hipMemsetAsync(reinterpret_cast<unsigned*>(base + ws.bar) + kBarSticky, 0, sizeof(unsigned), stream);
        if (e != hipSuccess) return sgp::fail((int)e, "sgp_gesn_f32: %s", hipGetErrorString(e));
        for (; t_done < T && !rc; t_done += kStepChunk) {
            const int tc = T - t_done < kStepChunk ? T - t_done : kStepChunk;
            rc = input_term(t_done, tc);
            if (rc) break;
            if (t_done == 0 && R % 16 == 0 && R <= 384) {
                rc = sgp_gesn::pack(base + ws.wcat, base + ws.wpk, R, L, stream);
                if (rc) break;
            }
            const int r = sgp_gesn::run_chunk(rowptr, col, val, base + ws.p0, base + ws.wcat, base + ws.wpk, base + ws.bcat,
                                              base + ws.c, h_state, out + (size_t)t_done * out_step_stride,
                                              out_row_stride, out_step_stride,
                                              reinterpret_cast<unsigned*>(base + ws.bar), alpha, act,
                                              tc, N, R, L, stream);
            if (r > 0) break;                      // refused: nothing was launched for this chunk
            if (r < 0) return r;
            used_persistent = true;
        }
        if (rc) return rc;
        if (used_persistent) {
            unsigned failed = 0;
            e = hipMemcpyAsync(&failed, reinterpret_cast<unsigned*>(base + ws.bar) + kBarSticky, sizeof(unsigned),
                               hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return sgp::fail((int)e, "sgp_gesn_f32: %s", hipGetErrorString(e));
            if (failed) return sgp::fail(SGP_EUNSUP, "sgp_gesn_f32: grid barrier of the persistent kernel timed out");
        }
        if (t_done >= T) return 0;
    }
    float* hcur = h_state;
    float* hnext = base + ws.hb;
    for (int i = 0; i < L && !rc; ++i)        // z_i for the first step from the initial states
        rc = launch_gemm(hcur + i * NR, R, wcat(i), R, bcat(i), cbuf(i), 2 * R, N, R, R, stream);
    for (int t0 = t_done; t0 < T && !rc; t0 += kStepChunk) {
        const int tc = T - t0 < kStepChunk ? T - t0 : kStepChunk;
        float* p0 = base + ws.p0;
        rc = input_term(t0, tc);
        for (int s = 0; s < tc && !rc; ++s) {
            float* out_t = out + (size_t)(t0 + s) * out_step_stride;
            for (int i = 0; i < L && !rc; ++i) {
                const float* p = i == 0 ? p0 + s * NR : cbuf(i - 1) + R;
                const long long ldp = i == 0 ? R : 2 * R;
                rc = launch_update(rowptr, col, val, cbuf(i), 2 * R, p, ldp, hcur + i * NR, alpha[i], act,
                                   hnext + i * NR, out_t + (size_t)i * R, out_row_stride, N, R, stream);
                if (!rc)
                    rc = launch_gemm(hnext + i * NR, R, wcat(i), R, bcat(i), cbuf(i), 2 * R, N, n_out(i), R, stream);
            }
            float* tmp = hcur; hcur = hnext; hnext = tmp;
        }
    }
    if (!rc && hcur != h_state) {
        e = hipMemcpyAsync(h_state, hcur, (size_t)L * NR * sizeof(float), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return sgp::fail((int)e, "sgp_gesn_f32: state copy: %s", hipGetErrorString(e));
    }
    return rc;
}

}  // extern "C"
