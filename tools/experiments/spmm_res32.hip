// 32-feature form of the register-resident two-phase row-group SpMM (reference call site:
// lib/sgp_preprocessing.py:202, x = adj @ x).  gfx950 / wave64 only.  Same plan arrays and
// arithmetic as spmm_res; a workgroup owns a 64-row tile and a 32-FEATURE slice, has 8 waves and
// stages 128-byte rows (56 KiB), so that TWO workgroups share a CU: while one sits in a barrier,
// ramps up its operand reads or folds and stores, the other one's waves keep the matrix pipe busy
// (timeline of spmm_res: the compute phases are 83 % of a step and run the pipe at 64 %; the rest
// is barrier skew the second workgroup can fill).  A wave carries TWO row groups, one per 32-lane
// half: lane (h, q, c) reads chunk c (16 bytes) of class q's staged row for group 2 w + h; the
// MFMA's 16 blocks are (h, q, c >> 2) with the A operand shared by the two blocks of a class
// (cbsz = 1, abid = s & 1): one weight register per 2 super-steps.  The halves run the longer of
// their two ranges (padding: staged row 0, weight 0).  Stream ranges beyond the resident SH
// super-steps read the plan arrays from global memory (no room for an LDS copy at 2 workgroups
// per CU).
#include "common.h"
#include <stdlib.h>

using sgp::f32x4;

namespace {

struct Src2 {
    const float* x;  long long xrs, xbs;
    const float* xh; long long xhrs, xhbs;
    int n_own;
};

struct R32Args {
    const int* uptr; const int* ucol; const int* usplit;
    const int* gptr;                       // [2 * GT * n_tiles + 1]: (A, B) quad ranges per group
    const int* gsup;                       // [2 * GT * n_tiles]: super-steps per range
    const int* gidx; const float* gw; const int* rowmap;
    int n_tiles;
    Src2 src;
    float* Y; long long yrs, ybs;
    int n_rows, batch, feat;
    int t_chunk, n_tchunks;
    unsigned* dbg;                         // timeline stamps (ablation builds only)
};

constexpr int SH = 20;                     // super-steps per range held in registers (80 columns)
constexpr int WH = (SH + 3) / 4;           // weight registers per range

__device__ __forceinline__ void dma16_saddr(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
__device__ __forceinline__ void dma16_vaddr(const void* vaddr, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(vaddr), "s"(lds_off) : "memory");
}


template <bool HALO, int D, int ABL = 0>
__global__ __launch_bounds__(512, 4) void spmm_res32(R32Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NW = 8, GT = 16, PASSES = 7;
    constexpr int RPP = NW * 8;                           // staged rows per DMA pass (8 per 1-KiB piece)
    constexpr int WH2 = SH / 2;                           // weight registers per range

    const int nwg = a.n_tiles * a.n_tchunks;
    const int orig = blockIdx.x;
    const int qq = nwg >> 3, rr = nwg & 7, x8 = orig & 7;
    const int wg = (x8 < rr ? x8 * (qq + 1) : rr * (qq + 1) + (x8 - rr) * qq) + (orig >> 3);
    const int tile = wg % a.n_tiles;
    const int tchunk = wg / a.n_tiles;
    const int f_base = blockIdx.y * 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int c8 = lane & 7;                              // 16-byte chunk of the 128-byte row
    const int q = (lane >> 3) & 3;                        // column class
    const int hh = lane >> 5;                             // which of the wave's two groups
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = a.uptr[tile];
    const int nU = a.uptr[tile + 1] - u0;
    const int uA = a.usplit[tile];

    const int t_begin = tchunk * a.t_chunk;
    const int t_end = min(a.batch, t_begin + a.t_chunk);
    if (t_begin >= t_end) return;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // ---- DMA bookkeeping: a 1-KiB piece = 8 staged rows x 128 bytes; lane (r8 = lane >> 3, c8)
    unsigned voff[PASSES];
    unsigned halo_mask = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int u = p * RPP + wave * 8 + (lane >> 3);
        const int c = (u < nU) ? a.ucol[u0 + u] : (nU > 0 ? a.ucol[u0] : 0);
        if (HALO && c >= a.src.n_own) {
            halo_mask |= 1u << p;
            voff[p] = (unsigned)((c - a.src.n_own) * (int)a.src.xhrs + f_base + c8 * 4) * 4u;
        } else {
            voff[p] = (unsigned)(c * (int)a.src.xrs + f_base + c8 * 4) * 4u;
        }
    }
    // (uA is a multiple of 4, not of 8: a piece may straddle the A | B boundary -- it then belongs
    // to BOTH segments' requests; re-fetching 4 rows is harmless, they are read-only per step)
    unsigned piecesA = 0, piecesB = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int r0 = p * RPP + wave * 8;
        if (r0 < uA) piecesA |= 1u << p;
        if (r0 + 8 > uA && r0 < nU) piecesB |= 1u << p;
    }
    piecesA = __builtin_amdgcn_readfirstlane(piecesA);
    piecesB = __builtin_amdgcn_readfirstlane(piecesB);
    const unsigned piece0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    const char* x_step = reinterpret_cast<const char*>(a.src.x + (long long)t_begin * a.src.xbs);
    const char* h_step = reinterpret_cast<const char*>(a.src.xh + (long long)t_begin * a.src.xhbs);
    const long long x_inc = a.src.xbs * 4, h_inc = a.src.xhbs * 4;
    const char* const x_step0 = x_step;
    const char* const h_step0 = h_step;
    auto dma_segment = [&](const char* xt, const char* ht, unsigned pieces) {
        if constexpr (ABL & 1) return;
        if constexpr (ABL & 4) { xt = x_step0; ht = h_step0; }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if (pieces & (1u << p)) {                     // scalar
                const unsigned dst = piece0 + (unsigned)p * (unsigned)(RPP * 128);
                if constexpr (HALO) {
                    const char* b = ((halo_mask >> p) & 1u) ? ht : xt;
                    dma16_vaddr(b + voff[p], __builtin_amdgcn_readfirstlane(dst));
                } else {
                    dma16_saddr(voff[p], xt, dst);
                }
            }
        }
    };

    // ---- the two groups' streams -> registers (once per workgroup); per-lane group = 2 wave + hh
    unsigned addr[2][SH];
    float w[2][WH2];
    int n[2];
    int qb_l[2];                                           // first quad of the lane's range (per half)
    int n_l[2];                                            // super-steps of the lane's own range
    {
        const int grp = (tile * GT + wave * 2 + hh) * 2;
        const int q0 = a.gptr[grp], q1 = a.gptr[grp + 1], q2 = a.gptr[grp + 2];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int qb = ph ? q1 : q0, qe = ph ? q2 : q1;
            qb_l[ph] = qb;
            n_l[ph] = a.gsup[grp + ph];
            const int other = __shfl_xor(n_l[ph], 32);
            n[ph] = __builtin_amdgcn_readfirstlane(max(n_l[ph], other));
#pragma unroll
            for (int p = 0; p < WH2; ++p) {
                // lane (hh, q, b2 = c8 >> 2, i = c8 & 3): weight of row i, class q, super-step 2 p + b2
                const int s = 2 * p + (c8 >> 2);
                const int quad = qb + (s >> 2);
                w[ph][p] = (quad < qe) ? a.gw[(long long)quad * 64 + q * 16 + (s & 3) * 4 + (c8 & 3)] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < SH; ++s) {
                const int quad = qb + (s >> 2);
                const unsigned off = (quad < qe) ? (unsigned)a.gidx[(long long)quad * 16 + q * 4 + (s & 3)] : 0u;
                addr[ph][s] = lds0 + (off >> 1) + c8 * 16;   // plan offsets are for 256-byte rows
            }
        }
    }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int s = 0; s < SH; ++s) asm volatile("" : "+v"(addr[ph][s]));
#pragma unroll
        for (int p = 0; p < WH2; ++p) asm volatile("" : "+v"(w[ph][p]));
    }
    const int row = a.rowmap[tile * (GT * 4) + (wave * 2 + hh) * 4 + q];
    const bool has_row = row >= 0;
    const unsigned yoff = (unsigned)((long long)(row < 0 ? 0 : row) * a.yrs + f_base + c8 * 4) * 4u;
    char* y_step = reinterpret_cast<char*>(a.Y + (long long)t_begin * a.ybs);
    const long long y_inc = a.ybs * 4;
    const bool q_odd = (q & 1) != 0;

    f32x4 ring[D];
    f32x4 acc[4];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; acc[2] = acc[0]; acc[3] = acc[0];

    // fold of the 4 column classes of a half (lane bits 3 and 4) so that class q keeps row q:
    // across bit 3 with a DPP rotate inside the 16-lane row (the lane sends the value its partner
    // wants and keeps the other), across bit 4 with v_permlane16_swap as in spmm_res
    auto emit = [&](char* ys) {
        f32x4 out;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float keep01 = q_odd ? acc[m].y : acc[m].x, send01 = q_odd ? acc[m].x : acc[m].y;
            const float keep23 = q_odd ? acc[m].w : acc[m].z, send23 = q_odd ? acc[m].z : acc[m].w;
            const float r01 = keep01 + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send01), 0x128, 0xf, 0xf, false));
            const float r23 = keep23 + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send23), 0x128, 0xf, 0xf, false));
            auto hsw = __builtin_amdgcn_permlane16_swap(__float_as_uint(r01), __float_as_uint(r23), false, false);
            out[m] = __uint_as_float(hsw[0]) + __uint_as_float(hsw[1]);
        }
        if (has_row) __builtin_nontemporal_store(out, reinterpret_cast<f32x4*>(ys + yoff));
    };

    typedef const __attribute__((address_space(3))) f32x4* lds_f4_t;
    // super-steps SH .. n-1 of a long range, from the plan arrays (per lane: its own group's quads;
    // a half whose range is shorter runs on with weight 0 / staged row 0)
    auto overflow = [&](int ph, int nsteps) {
        for (int s0 = SH; s0 < nsteps; s0 += 2) {
            const int sw = s0 + (c8 >> 2);
            const bool okw = sw < n_l[ph];
            const float wv = okw ? a.gw[(long long)(qb_l[ph] + (sw >> 2)) * 64 + q * 16 + (sw & 3) * 4 + (c8 & 3)] : 0.f;
            f32x4 xs[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int s = s0 + b;
                const unsigned off = (s < n_l[ph]) ? (unsigned)a.gidx[(long long)(qb_l[ph] + (s >> 2)) * 16 + q * 4 + (s & 3)] : 0u;
                xs[b] = *(lds_f4_t)(lds0 + (off >> 1) + c8 * 16);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].x, acc[0], 1, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].y, acc[1], 1, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].z, acc[2], 1, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].w, acc[3], 1, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].x, acc[0], 1, 1, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].y, acc[1], 1, 1, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].z, acc[2], 1, 1, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].w, acc[3], 1, 1, 0);
        }
    };

#define SGP_RD(P_, S_) asm volatile("ds_read_b128 %0, %1" : "=v"(ring[(S_) % D]) : "v"(addr[P_][S_]))
#define SGP_WAITN(S_) ((S_) >= D ? ((SH - 1 - (S_)) < (D - 1) ? (SH - 1 - (S_)) : (D - 1)) : (D - 1))
#define SGP_WT(S_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[(S_) % D]) : "n"(SGP_WAITN(S_)))
#define SGP_MF(ACC_, W_, X_, AB_) __builtin_amdgcn_mfma_f32_4x4x1f32(W_, X_, ACC_, 1, AB_, 0)
#define SGP_SLOT2(P_, S_, AB_, FIRST_)                                                             \
    {                                                                                              \
        const f32x4 x = ring[(S_) % D];                                                            \
        const float wv = w[P_][(S_) >> 1];                                                         \
        if (FIRST_) {                                                                              \
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};                                                  \
            acc[0] = SGP_MF(z, wv, x.x, AB_); acc[1] = SGP_MF(z, wv, x.y, AB_);                    \
            acc[2] = SGP_MF(z, wv, x.z, AB_); acc[3] = SGP_MF(z, wv, x.w, AB_);                    \
        } else {                                                                                   \
            acc[0] = SGP_MF(acc[0], wv, x.x, AB_); acc[1] = SGP_MF(acc[1], wv, x.y, AB_);          \
            acc[2] = SGP_MF(acc[2], wv, x.z, AB_); acc[3] = SGP_MF(acc[3], wv, x.w, AB_);          \
        }                                                                                          \
    }
#define SGP_SLOT(P_, S_, FIRST_)                                                                   \
    if (((S_) & 1) == 0) SGP_SLOT2(P_, S_, 0, FIRST_) else SGP_SLOT2(P_, S_, 1, FIRST_)
#define SGP_PHASE(P_, MID_)                                                                        \
    { _Pragma("unroll") for (int s = 0; s < D; ++s) SGP_RD(P_, s); }                               \
    { MID_ }                                                                                       \
    if (n[P_] > 0) {                                                                               \
        _Pragma("unroll") for (int s = 0; s < SH; ++s) {                                           \
            SGP_WT(s);                                                                             \
            SGP_SLOT(P_, s, (P_) == 0 && s == 0)                                                   \
            if (s + D < SH) SGP_RD(P_, s + D);                                                     \
            if (s + 1 == n[P_]) break;                                                             \
        }                                                                                          \
        if (n[P_] > SH) overflow(P_, n[P_]);                                                       \
    } else if ((P_) == 0) {                                                                        \
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; acc[2] = acc[0]; acc[3] = acc[0];     \
    }

    __syncthreads();
    dma_segment(x_step, h_step, piecesA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool dma_first = wave >= NW / 2;
    for (int t = t_begin; t < t_end; ++t) {
        asm volatile("" : "+s"(n[0]), "+s"(n[1]));
        asm volatile("s_barrier" ::: "memory");
        if (dma_first) dma_segment(x_step, h_step, piecesB);
        SGP_PHASE(0, if (t > t_begin) emit(y_step - y_inc);)
        if (!dma_first) dma_segment(x_step, h_step, piecesB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);
        SGP_PHASE(1, )
        if (!dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        x_step += x_inc; h_step += h_inc; y_step += y_inc;
    }
    emit(y_step - y_inc);
#undef SGP_PHASE
#undef SGP_SLOT
#undef SGP_SLOT2
#undef SGP_MF
#undef SGP_WT
#undef SGP_WAITN
#undef SGP_RD
}

int r32_chunk_cap() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SGP_SPMM_CHUNK"); v = e ? atoi(e) : 32; if (v < 1) v = 32; }
    return v;
}

template <bool HALO>
int launch_r32(const R32Args& a, hipStream_t s) {
    const size_t lds_bytes = 7 * 64 * 128;                // 56 KiB: two workgroups per CU
    dim3 grid((unsigned)(a.n_tiles * a.n_tchunks), a.feat / 32);
#ifdef SGP_ABLATION
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("SGP_PIPE_ABL"); abl = e ? atoi(e) : 0; }
#define SGP_ABL(V)                                                                                 \
    if (abl == V) {                                                                                \
        auto k4 = spmm_res32<HALO, 4, V>;                                                          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL(k4, grid, dim3(512), lds_bytes, s, a);                                  \
        return sgp::check_launch("spmm_res32");                                                    \
    }
    SGP_ABL(1) SGP_ABL(4)
#undef SGP_ABL
#endif
    auto kern = spmm_res32<HALO, 4>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_res32: LDS opt-in: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds_bytes, s, a);
    return sgp::check_launch("spmm_res32");
}

}  // namespace

extern "C" {

int sgp_spmm_res32_f32(const int32_t* uptr, const int32_t* ucol, const int32_t* usplit,
                       const int32_t* gptr, const int32_t* gsup, const int32_t* gidx, const float* gw,
                       const int32_t* rowmap,
                       int32_t n_tiles, int32_t max_union, int32_t max_tile_quads,
                       const float* X, int64_t xrs, int64_t xbs,
                       const float* Xh, int64_t xhrs, int64_t xhbs, int32_t n_own,
                       float* Y, int64_t yrs, int64_t ybs,
                       int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                       sgp_stream_t stream) {
    SGP_REQUIRE(uptr && ucol && usplit && gptr && gsup && gidx && gw && rowmap && X && Y,
                "sgp_spmm_res32_f32: null pointer");
    SGP_REQUIRE(n_tiles >= 0 && n_rows >= 0 && batch >= 0 && max_union >= 0, "sgp_spmm_res32_f32: bad size");
    {
        const long long own = Xh ? n_own : n_cols, far = Xh ? n_cols - n_own : 0;
        SGP_REQUIRE(n_cols >= 0 && own >= 0 && far >= 0 && own * xrs < (1ll << 30) && far * xhrs < (1ll << 30) &&
                    (long long)n_rows * yrs < (1ll << 30),
                    "sgp_spmm_res32_f32: row offsets exceed 32 bits (use sgp_spmm_csr_f32)");
    }
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    if (feat % 32 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_res32_f32: feat=%d is not a multiple of 32", feat);
    if (max_union > 7 * 64)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_res32_f32: tile working set (%d rows) exceeds the stage (448)", max_union);
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 && sgp::aligned16(X) &&
                sgp::aligned16(Y) && (!Xh || (xhrs % 4 == 0 && xhbs % 4 == 0 && sgp::aligned16(Xh))),
                "sgp_spmm_res32_f32: strides/pointers must be 16-byte aligned");
    R32Args a;
    a.uptr = uptr; a.ucol = ucol; a.usplit = usplit; a.gptr = gptr; a.gsup = gsup; a.gidx = gidx; a.gw = gw;
    a.rowmap = rowmap;
    a.n_tiles = n_tiles;
    a.src = Src2{X, xrs, xbs, Xh ? Xh : X, xhrs, xhbs, Xh ? n_own : 0x7fffffff};
    a.Y = Y; a.yrs = yrs; a.ybs = ybs;
    a.n_rows = n_rows; a.batch = batch; a.feat = feat;
    const int nft = feat / 32;
    long long want = (long long)batch * n_tiles * nft / 8192;
    int tc = (int)(want < 16 ? 16 : (want > r32_chunk_cap() ? r32_chunk_cap() : want));
    if (tc > batch) tc = batch;
    a.t_chunk = tc;
    a.n_tchunks = (batch + tc - 1) / tc;
    a.dbg = nullptr;
    hipStream_t s = (hipStream_t)stream;
    return Xh ? launch_r32<true>(a, s) : launch_r32<false>(a, s);
}

}  // extern "C"
