// C ABI of the reservoir layer + weight packing.  Kernels: reservoir_impl.h, instantiated
// per reservoir width in reservoir_jt*.hip (separate translation units build in parallel).
#include "reservoir_impl.h"

namespace sgp_res {
int launch_jt1(const ResArgs&, int, hipStream_t);
int launch_jt2(const ResArgs&, int, hipStream_t);
int launch_jt4(const ResArgs&, int, hipStream_t);
int launch_jt8(const ResArgs&, int, hipStream_t);
int launch_jt16(const ResArgs&, int, hipStream_t);
}

namespace {
using namespace sgp_res;
__global__ void pack_weights(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                             const float* __restrict__ b, float* __restrict__ out,
                             int F, int R, int JT, int NKX) {
    const long long n_bias = (long long)JT * 16;
    const long long n_wx = (long long)JT * NKX * 64;
    const long long total = packed_floats(JT, NKX);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < n_bias) {
            const int j = (int)i;
            v = j < R ? b[j] : 0.f;
        } else if (i < n_bias + n_wx) {
            const long long o = i - n_bias;
            int l, ks, jt;
            if (NKX % 4 == 0) {                      // [JT][NKX/4][64][4]
                const int s = (int)(o & 3);
                l = (int)((o >> 2) & 63);
                const int k4 = (int)((o >> 8) % (NKX / 4));
                jt = (int)((o >> 8) / (NKX / 4));
                ks = 4 * k4 + s;
            } else {                                 // [JT][NKX][64]
                l = (int)(o & 63);
                ks = (int)((o >> 6) % NKX);
                jt = (int)((o >> 6) / NKX);
            }
            const int j = 16 * jt + (l & 15);
            const int k = (l >> 4) * NKX + ks;
            v = (j < R && k < F) ? w_ih[(long long)j * F + k] : 0.f;
        } else {
            const long long o = i - n_bias - n_wx;
            const int s = (int)(o & 3);
            const int l = (int)((o >> 2) & 63);
            const int kb = (int)((o >> 8) % JT);
            const int jt = (int)((o >> 8) / JT);
            const int j = 16 * jt + (l & 15);
            const int k = 16 * kb + 4 * (l >> 4) + s;
            v = (j < R && k < R) ? w_hh[(long long)j * R + k] : 0.f;
        }
        out[i] = v;
    }
}

// one thread per (jt, kb, lane): 8 weights -> 3 x 16 bytes
__global__ void pack_weights_bf3(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                 const float* __restrict__ b, char* __restrict__ out, int F, int R, int JT, int NKX) {
    const int KBH = bf3_kbh(JT), KB = KBH + bf3_kbx(NKX);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < JT * 16) reinterpret_cast<float*>(out)[i] = i < R ? b[i] : 0.f;
    if (i >= JT * KB * 64) return;
    const int l = i & 63, kb = (i >> 6) % KB, jt = (i >> 6) / KB;
    const int j = 16 * jt + (l & 15), g = l >> 4;
    float w[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (kb < KBH) {
            const int tt = 2 * kb + (s >> 2), k = 16 * tt + 4 * g + (s & 3);
            w[s] = (j < R && tt < JT && k < R) ? w_hh[(long long)j * R + k] : 0.f;
        } else {
            const int ks = 8 * (kb - KBH) + s, k = bf3_feature(NKX, g, ks);
            w[s] = (j < R && ks < NKX && k < F) ? w_ih[(long long)j * F + k] : 0.f;
        }
    }
    u32x4 p1, p2, p3;
    bf3_split8(w, p1, p2, p3);
    u32x4* o = reinterpret_cast<u32x4*>(out + (long long)JT * 64) + ((long long)(jt * KB + kb) * 3) * 64 + l;
    o[0] = p1; o[64] = p2; o[128] = p3;
}


// 1 into *bad when any of the n state values lies outside [-1, 1] or is NaN (the word is cleared by the launcher)
__global__ void state_outside_unit_interval(const float* __restrict__ h, long long n, int* __restrict__ bad) {
    int out = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out |= !(fabsf(h[i]) <= 1.f);
    if (__syncthreads_or(out) && threadIdx.x == 0) atomicOr(bad, 1);
}

// Large-N form of the bounded-state loop (reservoir_layer_bf3 with H16): pack_weights_bf3's layout with the recurrent blocks'
// first two piece slots holding fp16 hi / lo of w_hh[j, :] 2^e_j, the input blocks and the bias as bf16 pieces / fp32 of
// the values times 2^(e_j + 14) (the accumulator then carries that factor as a whole), and the way back 2^(-e_j - 14) of
// the JT x 16 rows BEHIND the fragments.  Exact widths (R = 16 JT, F = 4 NKX).
__global__ void pack_weights_bf3h(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                  const float* __restrict__ b, char* __restrict__ out, int F, int R, int JT, int NKX) {
    const int KBH = bf3_kbh(JT), KB = KBH + bf3_kbx(NKX);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= JT * KB * 64) return;
    const int l = i & 63, kb = (i >> 6) % KB, jt = (i >> 6) / KB;
    const int j = 16 * jt + (l & 15), g = l >> 4;
    float amax = 0.f;
    for (int k = 0; k < R; ++k) amax = fmaxf(amax, fabsf(w_hh[(long long)j * R + k]));
    int e = 0;
    if (amax > 0.f && amax < __builtin_inff()) {
        int k;
        const float mant = frexpf(amax, &k);
        e = (mant == 0.5f ? 15 : 14) - k;
        e = min(40, max(-40, e));                              // (keeps 2^(e + 14) |w_ih x| far inside fp32)
    }
    const float ws = ldexpf(1.f, e), up = ldexpf(1.f, e + 14);
    if (kb == 0 && g == 0) {
        reinterpret_cast<float*>(out)[j] = b[j] * up;
        reinterpret_cast<float*>(out + bf3_packed_bytes(JT, NKX))[j] = ldexpf(1.f, -e - 14);
    }
    float w[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (kb < KBH) {
            const int tt = 2 * kb + (s >> 2), k = 16 * tt + 4 * g + (s & 3);
            w[s] = (tt < JT && k < R) ? w_hh[(long long)j * R + k] : 0.f;
        } else {
            const int ks = 8 * (kb - KBH) + s, k = bf3_feature(NKX, g, ks);
            w[s] = (ks < NKX && k < F) ? w_ih[(long long)j * F + k] * up : 0.f;
        }
    }
    u32x4* o = reinterpret_cast<u32x4*>(out + (long long)JT * 64) + ((long long)(jt * KB + kb) * 3) * 64 + l;
    if (kb < KBH) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) sj16_split2(w[2 * d], w[2 * d + 1], ws, hi[d], lo[d]);
        o[0] = u32x4{hi[0], hi[1], hi[2], hi[3]}; o[64] = u32x4{lo[0], lo[1], lo[2], lo[3]}; o[128] = u32x4{0u, 0u, 0u, 0u};
    } else {
        u32x4 p1, p2, p3;
        bf3_split8(w, p1, p2, p3);
        o[0] = p1; o[64] = p2; o[128] = p3;
    }
}

// Two-piece fp16 fragments of W_hh for the split-J kernel's bounded-state loop (reservoir_splitj_bf3.h): one thread per
// (jt, kb, lane), row j scaled by the power of two that puts its largest entry at 2^13 .. 2^14 (computed here: a row is
// at most 128 floats), 2^(-e_j - 14) -- the way back, the state's 2^14 included -- in front of the fragments.
__global__ void pack_weights_sj16(const float* __restrict__ w_hh, char* __restrict__ out, int R, int JT) {
    const int KBH = bf3_kbh(JT);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= JT * KBH * 64) return;
    const int l = i & 63, kb = (i >> 6) % KBH, jt = (i >> 6) / KBH;
    const int j = 16 * jt + (l & 15), g = l >> 4;
    float amax = 0.f;
    if (j < R)
        for (int k = 0; k < R; ++k) amax = fmaxf(amax, fabsf(w_hh[(long long)j * R + k]));
    int e = 0;
    if (amax > 0.f && amax < __builtin_inff()) {
        int k;
        const float mant = frexpf(amax, &k);                   // amax = mant 2^k, mant in [0.5, 1)
        e = (mant == 0.5f ? 15 : 14) - k;                       // floor(log2(16384 / amax))
        e = min(100, max(-100, e));
    }
    const float ws = ldexpf(1.f, e);
    if (kb == 0 && g == 0) reinterpret_cast<float*>(out)[j] = ldexpf(1.f, -e - 14);
    unsigned hi[4], lo[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        float w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s = 2 * d + h, tt = 2 * kb + (s >> 2), k = 16 * tt + 4 * g + (s & 3);
            w[h] = (j < R && tt < JT && k < R) ? w_hh[(long long)j * R + k] : 0.f;
        }
        sj16_split2(w[0], w[1], ws, hi[d], lo[d]);
    }
    u32x4* o = reinterpret_cast<u32x4*>(out + (long long)JT * 64) + ((long long)(jt * KBH + kb) * 2) * 64 + l;
    o[0] = u32x4{hi[0], hi[1], hi[2], hi[3]}; o[64] = u32x4{lo[0], lo[1], lo[2], lo[3]};
}

// streamed layout of the wide reservoirs: one thread per (sub-block = 2 k-block + half, tile of the half, lane)
__global__ void pack_weights_sbf3(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                  const float* __restrict__ b, char* __restrict__ out, int F, int R, int JT, int NKX) {
    const int KBH = JT / 2, KBX = NKX / 8, NSB = 2 * (KBH + KBX);     // input k-blocks first
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < JT * 16) reinterpret_cast<float*>(out)[i] = i < R ? b[i] : 0.f;
    if (i >= NSB * 8 * 64) return;
    const int l = i & 63, j8 = (i >> 6) & 7, sb = i >> 9;
    const int kb = sb >> 1, jt = 8 * (sb & 1) + j8;
    const int j = 16 * jt + (l & 15), g = l >> 4;
    float w[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (kb >= KBX) {
            const int k = 16 * (2 * (kb - KBX) + (s >> 2)) + 4 * g + (s & 3);
            w[s] = (j < R && k < R) ? w_hh[(long long)j * R + k] : 0.f;
        } else {
            const int k = bf3_feature(NKX, g, 8 * kb + s);
            w[s] = (j < R && k < F) ? w_ih[(long long)j * F + k] : 0.f;
        }
    }
    u32x4 p1, p2, p3;
    bf3_split8(w, p1, p2, p3);
    u32x4* o = reinterpret_cast<u32x4*>(out + 1024) + ((long long)(sb * 8 + j8) * 3) * 64 + l;
    o[0] = p1; o[64] = p2; o[128] = p3;
}

// Wide form of the bounded-state loop (reservoir_layer_stream_bf3 with H16): pack_weights_sbf3's streamed layout with the
// recurrent sub-blocks as [tile][2 fp16 pieces] in the first 16 of their 24 KB, bias and input fragments times 2^(e_j + 14)
// (pack_weights_bf3h), the rows' 2^(-e_j - 14) behind the kernel's dump area.
__global__ void pack_weights_sbf3h(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                   const float* __restrict__ b, char* __restrict__ out, int F, int R, int JT, int NKX) {
    const int KBH = JT / 2, KBX = NKX / 8, NSB = 2 * (KBH + KBX);     // input k-blocks first
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    auto row_exp = [&](int j) {
        float amax = 0.f;
        for (int k = 0; k < R; ++k) amax = fmaxf(amax, fabsf(w_hh[(long long)j * R + k]));
        int e = 0;
        if (amax > 0.f && amax < __builtin_inff()) {
            int k;
            const float mant = frexpf(amax, &k);
            e = (mant == 0.5f ? 15 : 14) - k;
            e = min(40, max(-40, e));
        }
        return e;
    };
    if (i < JT * 16) {
        const int e = i < R ? row_exp(i) : 0;
        reinterpret_cast<float*>(out)[i] = i < R ? b[i] * ldexpf(1.f, e + 14) : 0.f;
        reinterpret_cast<float*>(out + sbf3_packed_bytes(JT, NKX) + 1024)[i] = ldexpf(1.f, -e - 14);
    }
    if (i >= NSB * 8 * 64) return;
    const int l = i & 63, j8 = (i >> 6) & 7, sb = i >> 9;
    const int kb = sb >> 1, jt = 8 * (sb & 1) + j8;
    const int j = 16 * jt + (l & 15), g = l >> 4;
    const int e = j < R ? row_exp(j) : 0;
    const float ws = ldexpf(1.f, e), up = ldexpf(1.f, e + 14);
    float w[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (kb >= KBX) {
            const int k = 16 * (2 * (kb - KBX) + (s >> 2)) + 4 * g + (s & 3);
            w[s] = (j < R && k < R) ? w_hh[(long long)j * R + k] : 0.f;
        } else {
            const int k = bf3_feature(NKX, g, 8 * kb + s);
            w[s] = (j < R && k < F) ? w_ih[(long long)j * F + k] * up : 0.f;
        }
    }
    char* slot = out + 1024 + (long long)sb * (8 * 3 * 1024);
    if (kb >= KBX) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) sj16_split2(w[2 * d], w[2 * d + 1], ws, hi[d], lo[d]);
        u32x4* o = reinterpret_cast<u32x4*>(slot) + (long long)(j8 * 2) * 64 + l;
        o[0] = u32x4{hi[0], hi[1], hi[2], hi[3]}; o[64] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    } else {
        u32x4 p1, p2, p3;
        bf3_split8(w, p1, p2, p3);
        u32x4* o = reinterpret_cast<u32x4*>(slot) + (long long)(j8 * 3) * 64 + l;
        o[0] = p1; o[64] = p2; o[128] = p3;
    }
}

long long bf3_offset(int jt, int nkx) { return (packed_floats(jt, nkx) * 4 + 255) / 256 * 256; }

int pick_nkx(int F) {
    const int need = (F + 3) / 4;
    const int opts[] = {1, 2, 4, 8, 16, 32, 64};
    for (int o : opts) if (o >= need) return o;
    return 0;
}
int pick_jt(int R) {
    const int need = (R + 15) / 16;
    const int opts[] = {1, 2, 4, 8, 16};
    for (int o : opts) if (o >= need) return o;
    return 0;
}

}  // namespace


extern "C" {

int64_t sgp_reservoir_workspace_bytes(int32_t F, int32_t R) {
    const int jt = pick_jt(R), nkx = pick_nkx(F);
    if (!jt || !nkx) return -1;
    // the fp32 fragments, then (narrow reservoirs) the bf16 piece fragments of reservoir_bf3.h
    const long long body =
           bf3_offset(jt, nkx) + (bf3_supported(jt, nkx) || sjbf3_supported(jt, nkx) ? bf3_packed_bytes(jt, nkx) : 0) +
           (sjbf3_supported(jt, nkx) ? sj16_packed_bytes(jt) : 0) +
           (bf3_supported(jt, nkx) ? bf3_packed_bytes(jt, nkx) + jt * 64 + 256 : 0) +     // (+ the state test's word)
           (sbf3_supported(jt, nkx) ? 2 * (sbf3_packed_bytes(jt, nkx) + 1024) + 1024 + 256 : 0);   // + dump areas of the kernel; the
                                                                                   // two-piece fp16 copy + its row scales + the state test's word
    return (body + 255) / 256 * 256 + 1024;                                        // + the split-J form's dump area (last KB)
}

struct Pieces { int n, t_last, no_store; long long px, po, ps; const int* pred; int run_if; };

static int reservoir_run(const float* x, int64_t xrs, int64_t xss,
                         const float* w_ih, const float* w_hh, const float* b,
                         double alpha, int32_t act,
                         float* out, int64_t ors, int64_t oss,
                         float* h_state, void* workspace,
                         int32_t T, int32_t N, int32_t F, int32_t R, const Pieces& pc,
                         sgp_stream_t stream);

int sgp_reservoir_f32(const float* x, int64_t xrs, int64_t xss,
                      const float* w_ih, const float* w_hh, const float* b,
                      double alpha, int32_t act,
                      float* out, int64_t ors, int64_t oss,
                      float* h_state, void* workspace,
                      int32_t T, int32_t N, int32_t F, int32_t R,
                      sgp_stream_t stream) {
    return reservoir_run(x, xrs, xss, w_ih, w_hh, b, alpha, act, out, ors, oss, h_state, workspace, T, N, F, R,
                         Pieces{1, T, 0, 0, 0, 0, nullptr, 0}, stream);
}

int sgp_reservoir_pieces_f32(const float* x, int64_t xrs, int64_t xss,
                             const float* w_ih, const float* w_hh, const float* b,
                             double alpha, int32_t act,
                             float* out, int64_t ors, int64_t oss,
                             float* h_state, void* workspace,
                             int32_t t_piece, int32_t t_last, int32_t n_pieces,
                             int64_t x_piece_stride, int64_t out_piece_stride, int32_t no_store,
                             int32_t N, int32_t F, int32_t R,
                             const int32_t* pred, int32_t run_if, sgp_stream_t stream) {
    SGP_REQUIRE(n_pieces >= 1 && t_piece >= 0 && t_last >= 0 && t_last <= t_piece, "sgp_reservoir_pieces_f32: bad piece sizes");
    SGP_REQUIRE(n_pieces == 1 || h_state, "sgp_reservoir_pieces_f32: several pieces need their states [n_pieces][N][R]");
    SGP_REQUIRE(n_pieces <= 65535, "sgp_reservoir_pieces_f32: at most 65535 pieces");
    return reservoir_run(x, xrs, xss, w_ih, w_hh, b, alpha, act, out, ors, oss, h_state, workspace, t_piece, N, F, R,
                         Pieces{n_pieces, n_pieces > 1 ? t_last : t_piece, no_store != 0, x_piece_stride, out_piece_stride,
                                (long long)N * R, pred, run_if}, stream);
}

}  // extern "C"

static int reservoir_run(const float* x, int64_t xrs, int64_t xss,
                         const float* w_ih, const float* w_hh, const float* b,
                         double alpha, int32_t act,
                         float* out, int64_t ors, int64_t oss,
                         float* h_state, void* workspace,
                         int32_t T, int32_t N, int32_t F, int32_t R, const Pieces& pc,
                         sgp_stream_t stream) {
    SGP_REQUIRE(x && w_ih && w_hh && b && out && workspace, "sgp_reservoir_f32: null pointer");
    SGP_REQUIRE(T >= 0 && N >= 0 && F > 0 && R > 0, "sgp_reservoir_f32: bad size");
    SGP_REQUIRE(act >= SGP_ACT_TANH && act <= SGP_ACT_TANH_REL, "sgp_reservoir_f32: unknown activation %d", act);
    SGP_REQUIRE(sgp::aligned16(workspace), "sgp_reservoir_f32: workspace must be 16-byte aligned");
    if (T == 0 || N == 0) return 0;
    const int jt = pick_jt(R), nkx = pick_nkx(F);
    if (!jt) return sgp::fail(SGP_EUNSUP, "sgp_reservoir_f32: reservoir size %d > 256 not supported", R);
    if (!nkx) return sgp::fail(SGP_EUNSUP, "sgp_reservoir_f32: input size %d > 256 not supported", F);
    hipStream_t s = (hipStream_t)stream;
    const long long total = packed_floats(jt, nkx);
    int pg = (int)((total + 255) / 256);
    if (pg > 1024) pg = 1024;
    hipLaunchKernelGGL(pack_weights, dim3(pg), dim3(256), 0, s, w_ih, w_hh, b, (float*)workspace, F, R, jt, nkx);
    int rc = sgp::check_launch("pack_weights");
    if (rc) return rc;

    ResArgs a;
    // the device word "some initial state lies outside [-1, 1] (or is NaN)" for the two-piece fp16 instances' launch predicate
    auto test_state = [&](int* bad) -> int {
        hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), s);
        if (e != hipSuccess) return sgp::fail((int)e, "sgp_reservoir_f32: memset: %s", hipGetErrorString(e));
        const long long n = (long long)N * R;
        const long long want = (n + 256 * 16 - 1) / (256 * 16);
        hipLaunchKernelGGL(state_outside_unit_interval, dim3((unsigned)(want < 1024 ? want : 1024)), dim3(256), 0, s, h_state, n, bad);
        int rc2 = sgp::check_launch("state_outside_unit_interval");
        if (!rc2) a.bad_state = bad;
        return rc2;
    };
    a.x = x; a.xrs = xrs; a.xss = xss;
    a.wp = (const float*)workspace;
    a.wp_bf3 = nullptr;
    a.wp_h16 = nullptr;
    a.wp_h16l = nullptr; a.wp_h16s = nullptr;
    a.dump = reinterpret_cast<float*>((char*)workspace + sgp_reservoir_workspace_bytes(F, R) - 1024);
    a.bad_state = nullptr; a.pred = pc.pred; a.pred_want = pc.run_if;
    a.n_pieces = pc.n; a.t_last = pc.t_last; a.no_store = pc.no_store; a.px = pc.px; a.po = pc.po; a.ps = pc.ps;
    // res_bf3 = 0 (SGP_TUNE) keeps the exact-fp32 products for narrow reservoirs too
    static const bool use_bf3 = sgp::tune("res_bf3", 1) != 0;
    // (the split-J form for small N -- R = 64 / 128, up to 32 input features -- takes any R <= 16 jt, F <= 4 nkx: padded
    // units and features carry zero weights; the large-N form needs the exact widths)
    if (use_bf3 && ((bf3_supported(jt, nkx) && R == 16 * jt && F == 4 * nkx) || sjbf3_supported(jt, nkx))) {
        char* wb = (char*)workspace + bf3_offset(jt, nkx);
        const int threads = jt * (bf3_kbh(jt) + bf3_kbx(nkx)) * 64;
        hipLaunchKernelGGL(pack_weights_bf3, dim3((threads + 255) / 256), dim3(256), 0, s, w_ih, w_hh, b, wb, F, R, jt, nkx);
        rc = sgp::check_launch("pack_weights_bf3");
        if (rc) return rc;
        a.wp_bf3 = wb;
        // res_h16 = 0 (SGP_TUNE) keeps three bf16 pieces for the bounded (tanh) state of the split-J form too
        // (the state stays in [-1, 1] only under a convex leak: a leaking rate outside [0, 1], which the reference accepts,
        // keeps three bf16 pieces)
        static const bool h16_on = sgp::tune("res_h16", 1) != 0;
        const bool use_h16 = h16_on && alpha >= 0.0 && alpha <= 1.0;
        if (use_h16 && sjbf3_supported(jt, nkx) && act == SGP_ACT_TANH) {
            char* wh = wb + bf3_packed_bytes(jt, nkx);
            const int th = jt * bf3_kbh(jt) * 64;
            hipLaunchKernelGGL(pack_weights_sj16, dim3((th + 255) / 256), dim3(256), 0, s, w_hh, wh, R, jt);
            rc = sgp::check_launch("pack_weights_sj16");
            if (rc) return rc;
            a.wp_h16 = wh;
        }
        if (use_h16 && bf3_supported(jt, nkx) && R == 16 * jt && F == 4 * nkx && act == SGP_ACT_TANH) {
            char* wl = wb + bf3_packed_bytes(jt, nkx) + (sjbf3_supported(jt, nkx) ? sj16_packed_bytes(jt) : 0);
            hipLaunchKernelGGL(pack_weights_bf3h, dim3((threads + 255) / 256), dim3(256), 0, s, w_ih, w_hh, b, wl, F, R, jt, nkx);
            rc = sgp::check_launch("pack_weights_bf3h");
            if (rc) return rc;
            a.wp_h16l = wl;
            if (h_state) {
                rc = test_state(reinterpret_cast<int*>(wl + bf3_packed_bytes(jt, nkx) + jt * 64));
                if (rc) return rc;
            }
        }
    }
    if (use_bf3 && sbf3_supported(jt, nkx) && R == 16 * jt && F == 4 * nkx) {
        char* wb = (char*)workspace + bf3_offset(jt, nkx);
        const int threads = 2 * (jt / 2 + nkx / 8) * 8 * 64;
        hipLaunchKernelGGL(pack_weights_sbf3, dim3((threads + 255) / 256), dim3(256), 0, s, w_ih, w_hh, b, wb, F, R, jt, nkx);
        rc = sgp::check_launch("pack_weights_sbf3");
        if (rc) return rc;
        a.wp_bf3 = wb;
        static const bool h16s_on = sgp::tune("res_h16", 1) != 0;
        if (h16s_on && act == SGP_ACT_TANH && alpha >= 0.0 && alpha <= 1.0) {
            char* wh = wb + sbf3_packed_bytes(jt, nkx) + 1024;
            hipLaunchKernelGGL(pack_weights_sbf3h, dim3((threads + 255) / 256), dim3(256), 0, s, w_ih, w_hh, b, wh, F, R, jt, nkx);
            rc = sgp::check_launch("pack_weights_sbf3h");
            if (rc) return rc;
            a.wp_h16s = wh;
            if (h_state) {
                rc = test_state(reinterpret_cast<int*>(wh + sbf3_packed_bytes(jt, nkx) + 2048));
                if (rc) return rc;
            }
        }
    }
    a.out = out; a.ors = ors; a.oss = oss;
    a.h_state = h_state;
    a.alpha = (float)alpha;                      // scalar operands are rounded to fp32 like
    a.one_minus_alpha = (float)(1.0 - alpha);    // torch does for `(1 - alpha) * h` (reservoir.py:80)
    a.act = act; a.T = T; a.N = N; a.F = F; a.R = R;
    a.tiles_per_wave = 0; a.n_tiles = 0;
    switch (jt) {
        case 1: return launch_jt1(a, nkx, s);
        case 2: return launch_jt2(a, nkx, s);
        case 4: return launch_jt4(a, nkx, s);
        case 8: return launch_jt8(a, nkx, s);
        case 16: return launch_jt16(a, nkx, s);
    }
    return sgp::fail(SGP_EUNSUP, "sgp_reservoir_f32: unreachable");
}
