// Row-block form of the row-group SpMM (reference call site: lib/sgp_preprocessing.py:202,
// x = adj @ x).  gfx950 / wave64 only.  Host plan: sgp_amd/rowblock.py.
//
// A workgroup of NW = 8 waves owns a tile of up to 128 rows and a chunk of time steps; per step the
// tile's distinct source rows are staged in LDS by LDS-DMA in two alternating segments (as in
// spmm_pipe / spmm_res).  What differs is the work of a wave: it owns FOUR row groups, one per
// 16-lane class of v_mfma_f32_4x4x1_16b_f32, and every class walks its OWN column list.  Lane
// (q, li) reads 16 bytes (features 4 li .. 4 li + 3) of class q's next source row; the 4 MFMAs of
// the super-step multiply them with the 4 row weights of that column (A operand, broadcast inside
// the class with cbsz = 2 / abid).  The accumulators of a lane therefore hold FINISHED sums
// (row i of group q, features 4 li + m): no fold across lanes, 4 float4 stores per wave and step.
// Against the 64-row tiles of spmm_res: half the barriers and 0.75x the staged bytes per row
// (4.3 instead of 5.8 staged rows per result row on the 100-NN target graph), 8 instead of 16
// waves so that the whole stream of a wave -- per-lane LDS addresses of every super-step and one
// weight register per 4 super-steps -- stays in its 256 VGPRs for the time chunk.  A super-step is
//         s_waitcnt lgkmcnt(n) | 4 x v_mfma_f32_4x4x1_16b_f32 | ds_read_b128 (D super-steps ahead)
// with no VALU work.  Ranges longer than SH super-steps continue from the plan arrays in global
// memory (L2-resident; ~2 % of the super-steps of the target graph).
//
// Operand reads and their waits are inline asm (hipcc sinks a plain LDS load to its use across the
// scalar exit branches); LDS operations return in order, which makes the wait counts static.
#include "common.h"
#include <stdlib.h>

using sgp::f32x4;

namespace {

struct Src2 {
    const float* x;  long long xrs, xbs;
    const float* xh; long long xhrs, xhbs;
    int n_own;
};

struct BlkArgs {
    const int* uptr; const int* ucol; const int* usplit;
    const int* wptr; const int* nsteps; const int* soff; const float* sw; const int* rowmap;
    int n_tiles;
    Src2 src;
    float* Y; long long yrs, ybs;
    int n_rows, batch, feat;
    int t_chunk, n_tchunks;
};

__device__ __forceinline__ void dma16_saddr(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
__device__ __forceinline__ void dma16_vaddr(const void* vaddr, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(vaddr), "s"(lds_off) : "memory");
}

// NW waves (tile = 16 NW rows), SH super-steps per phase held in registers, operand ring D
// super-steps deep, PASSES x (4 NW) staged rows.  ABL (ablation builds): bit0 no staging DMA,
// bit2 staging always reads the chunk's first step.
template <bool HALO, int NW, int SH, int D, int PASSES, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void spmm_blk(BlkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    static_assert(SH % 4 == 0 && D <= SH, "SH must be a multiple of 4");
    constexpr int RPP = NW * 4;                           // staged rows per DMA pass
    constexpr int WH = SH / 4;
    constexpr int kStageBytes = PASSES * RPP * 256;

    const int nwg = a.n_tiles * a.n_tchunks;
    const int orig = blockIdx.x;
    // consecutive workgroup ids go to different XCDs: give every XCD a contiguous range of (tile,
    // chunk) pairs so that neighbouring tiles of one time chunk share an L2
    const int qq = nwg >> 3, rr = nwg & 7, x8 = orig & 7;
    const int wg = (x8 < rr ? x8 * (qq + 1) : rr * (qq + 1) + (x8 - rr) * qq) + (orig >> 3);
    const int tile = wg % a.n_tiles;
    const int tchunk = wg / a.n_tiles;
    const int f_base = blockIdx.y * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = tid & 15;
    const int q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = a.uptr[tile];
    const int nU = a.uptr[tile + 1] - u0;
    const int uA = a.usplit[tile];

    const int t_begin = tchunk * a.t_chunk;
    const int t_end = min(a.batch, t_begin + a.t_chunk);
    if (t_begin >= t_end) return;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // ---- DMA bookkeeping: source byte offset of every staged row in an LDS table behind the
    // stage (bit 0 set = the row lives in the halo buffer; offsets are multiples of 16)
    unsigned* tab = reinterpret_cast<unsigned*>(lds + kStageBytes);
    for (int u = tid; u < PASSES * RPP; u += NW * 64) {
        const int c = (u < nU) ? a.ucol[u0 + u] : (nU > 0 ? a.ucol[u0] : 0);
        unsigned o;
        if (HALO && c >= a.src.n_own) o = ((unsigned)((c - a.src.n_own) * (int)a.src.xhrs + f_base) * 4u) | 1u;
        else o = (unsigned)(c * (int)a.src.xrs + f_base) * 4u;
        tab[u] = o;
    }
    const unsigned* tab_lane = tab + wave * 4 + q;
    const unsigned li16 = li * 16;
    unsigned piecesA = 0, piecesB = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int r0 = p * RPP + wave * 4;
        if (r0 < uA) piecesA |= 1u << p;
        else if (r0 < nU) piecesB |= 1u << p;
    }
    piecesA = __builtin_amdgcn_readfirstlane(piecesA);
    piecesB = __builtin_amdgcn_readfirstlane(piecesB);
    unsigned piece0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    const char* x_step = reinterpret_cast<const char*>(a.src.x + (long long)t_begin * a.src.xbs);
    const char* h_step = reinterpret_cast<const char*>(a.src.xh + (long long)t_begin * a.src.xhbs);
    const long long x_inc = a.src.xbs * 4, h_inc = a.src.xhbs * 4;
    const char* const x_step0 = x_step;
    const char* const h_step0 = h_step;
    auto dma_segment = [&](const char* xt, const char* ht, unsigned pieces) {
        if constexpr (ABL & 1) return;
        if constexpr (ABL & 4) { xt = x_step0; ht = h_step0; }
        constexpr int B = 5;                              // offsets fetched per batch (temporaries)
#pragma unroll
        for (int p0 = 0; p0 < PASSES; p0 += B) {
            if (((pieces >> p0) & ((1u << B) - 1u)) == 0) continue;      // scalar
            unsigned o[B];
#pragma unroll
            for (int j = 0; j < B; ++j)
                if (p0 + j < PASSES) o[j] = tab_lane[(p0 + j) * RPP];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int p = p0 + j;
                if (p < PASSES && (pieces & (1u << p))) {                  // scalar
                    const unsigned dst = piece0 + (unsigned)p * (unsigned)(RPP * 256);
                    if constexpr (HALO) {
                        const char* b = (o[j] & 1u) ? ht : xt;
                        dma16_vaddr(b + ((o[j] & ~1u) + li16), __builtin_amdgcn_readfirstlane(dst));
                    } else {
                        dma16_saddr(o[j] + li16, xt, dst);
                    }
                }
            }
        }
    };

    // ---- the wave's stream -> registers (once per workgroup)
    // addresses: lane (q, li) holds the LDS byte address of chunk li of class q's staged row in
    // super-step s; weights: lane (q, b = li >> 2, i = li & 3) holds row i's weight for class q's
    // column in super-step 4 p + b (the MFMA of super-step s takes block s & 3 of its class).
    // Padding reads staged row 0 with weight 0.
    unsigned addr[2][SH];
    float w[2][WH];
    int n[2], s0[2];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        const int r = (tile * NW + wave) * 2 + ph;
        s0[ph] = __builtin_amdgcn_readfirstlane(a.wptr[r]);
        n[ph] = __builtin_amdgcn_readfirstlane(a.nsteps[r]);
#pragma unroll
        for (int p = 0; p < WH; ++p)
            w[ph][p] = (4 * p < n[ph]) ? a.sw[(long long)((s0[ph] >> 2) + p) * 64 + lane] : 0.f;
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            const unsigned off = (s < n[ph]) ? (unsigned)a.soff[(long long)(s0[ph] + s) * 4 + q] : 0u;
            addr[ph][s] = lds0 + off + li16;
        }
    }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int s = 0; s < SH; ++s) asm volatile("" : "+v"(addr[ph][s]));
#pragma unroll
        for (int p = 0; p < WH; ++p) asm volatile("" : "+v"(w[ph][p]));
    }
    unsigned yoff[4];
    unsigned has_row = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = a.rowmap[(tile * NW + wave) * 16 + q * 4 + i];
        if (row >= 0) has_row |= 1u << i;
        yoff[i] = (unsigned)((long long)(row < 0 ? 0 : row) * a.yrs + f_base + li * 4) * 4u;
    }
    char* y_step = reinterpret_cast<char*>(a.Y + (long long)t_begin * a.ybs);
    const long long y_inc = a.ybs * 4;

    f32x4 ring[D];
    f32x4 acc[4];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; acc[2] = acc[0]; acc[3] = acc[0];

    // a step's result: lane (q, li) holds rows 0..3 of group q x features 4 li + m in acc[m][i]
    // (streamed stores: they must not displace staged rows from L2)
    auto emit = [&](char* ys) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 out = {acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
            if (has_row & (1u << i)) __builtin_nontemporal_store(out, reinterpret_cast<f32x4*>(ys + yoff[i]));
        }
    };

    typedef const __attribute__((address_space(3))) f32x4* lds_f4_t;
    // super-steps SH .. n-1 of a long range, from the plan arrays (whole groups of 4: ranges are
    // padded to multiples of 4 super-steps with weight 0 / staged row 0)
    auto overflow = [&](int first, int nsteps) {
        for (int s = SH; s < nsteps; s += 4) {
            const float wv = a.sw[(long long)((first + s) >> 2) * 64 + lane];
            f32x4 xs[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                xs[b] = *(lds_f4_t)(lds0 + (unsigned)a.soff[(long long)(first + s + b) * 4 + q] + li16);
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].x, acc[0], 2, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].y, acc[1], 2, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].z, acc[2], 2, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].w, acc[3], 2, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].x, acc[0], 2, 1, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].y, acc[1], 2, 1, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].z, acc[2], 2, 1, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].w, acc[3], 2, 1, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[2].x, acc[0], 2, 2, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[2].y, acc[1], 2, 2, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[2].z, acc[2], 2, 2, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[2].w, acc[3], 2, 2, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[3].x, acc[0], 2, 3, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[3].y, acc[1], 2, 3, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[3].z, acc[2], 2, 3, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[3].w, acc[3], 2, 3, 0);
        }
    };

#define SGP_RD(P_, S_) asm volatile("ds_read_b128 %0, %1" : "=v"(ring[(S_) % D]) : "v"(addr[P_][S_]))
    // LDS reads issued after r(s) when super-step s starts: the rest of the look-ahead request and
    // the refills of super-steps 0 .. s-1 (s < D), or the refills s+1 .. s+D-1 (capped at SH-1)
#define SGP_WAITN(S_) ((S_) >= D ? ((SH - 1 - (S_)) < (D - 1) ? (SH - 1 - (S_)) : (D - 1)) : (D - 1))
#define SGP_WT(S_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[(S_) % D]) : "n"(SGP_WAITN(S_)))
#define SGP_MF(ACC_, W_, X_, AB_) __builtin_amdgcn_mfma_f32_4x4x1f32(W_, X_, ACC_, 2, AB_, 0)
#define SGP_SLOT4(P_, S_, AB_, FIRST_)                                                             \
    {                                                                                              \
        const f32x4 x = ring[(S_) % D];                                                            \
        const float wv = w[P_][(S_) >> 2];                                                         \
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
    if (((S_) & 3) == 0) SGP_SLOT4(P_, S_, 0, FIRST_) else if (((S_) & 3) == 1) SGP_SLOT4(P_, S_, 1, FIRST_) \
    else if (((S_) & 3) == 2) SGP_SLOT4(P_, S_, 2, FIRST_) else SGP_SLOT4(P_, S_, 3, FIRST_)
    // One phase = segment P_ of the stage.  The look-ahead reads are issued first; MID_ (the stores
    // of the previous step) runs under their latency.  In phase A a wave with columns restarts its
    // accumulators through the first MFMAs (C = 0); one without clears them.
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
        if (n[P_] > SH) overflow(s0[P_], n[P_]);                                                   \
    } else if ((P_) == 0) {                                                                        \
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; acc[2] = acc[0]; acc[3] = acc[0];     \
    }

    __syncthreads();                                      // offset table complete
    dma_segment(x_step, h_step, piecesA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool dma_first = wave >= NW / 2;
    for (int t = t_begin; t < t_end; ++t) {
        // the range lengths are re-made opaque every step: otherwise hipcc hoists all exit
        // comparisons out of the time loop as 64-bit masks and spills them to VGPR lanes
        asm volatile("" : "+s"(n[0]), "+s"(n[1]));
        // (likewise the piece masks and the LDS destination base of the DMA pieces: 19 hoisted
        // branch masks + 19 M0 values would live in SGPRs spilled to VGPR lanes)
        asm volatile("" : "+s"(piecesA), "+s"(piecesB), "+s"(piece0));
        // ---- phase A: segment A holds step t once every wave's pieces have landed
        asm volatile("s_barrier" ::: "memory");
        // the refill of the other segment is issued first by the younger half of the waves and
        // after their super-steps by the older half (waves w and w + NW/2 share a SIMD)
        if (dma_first) dma_segment(x_step, h_step, piecesB);
        SGP_PHASE(0, if (t > t_begin) emit(y_step - y_inc);)
        if (!dma_first) dma_segment(x_step, h_step, piecesB);
        // ---- phase B
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);
        SGP_PHASE(1, )
        if (!dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);
        // this wave's pieces of A(t+1) (and its stores) retired before the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        x_step += x_inc; h_step += h_inc; y_step += y_inc;
    }
    emit(y_step - y_inc);
#undef SGP_PHASE
#undef SGP_SLOT
#undef SGP_SLOT4
#undef SGP_MF
#undef SGP_WT
#undef SGP_WAITN
#undef SGP_RD
}

constexpr int kNW = 8, kPasses = 19;                       // 608 staged rows = 152 KiB + offset table

int blk_chunk_cap() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SGP_SPMM_CHUNK"); v = e ? atoi(e) : 32; if (v < 1) v = 32; }
    return v;
}

int g_blk_cfg = -1;
int blk_cfg() {                                            // (SH, D): 0 = (72, 4), 1 = (64, 6), 2 = (56, 8)
    if (g_blk_cfg < 0) { const char* e = getenv("SGP_SPMM_BLK_CFG"); g_blk_cfg = e ? atoi(e) : 0; }
    return g_blk_cfg;
}

template <bool HALO, int SH, int D>
int launch_blk(const BlkArgs& a, hipStream_t s) {
    const size_t lds_bytes = 160 * 1024;
    dim3 grid((unsigned)(a.n_tiles * a.n_tchunks), a.feat / 64);
#ifdef SGP_ABLATION
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("SGP_PIPE_ABL"); abl = e ? atoi(e) : 0; }
#define SGP_ABL(V)                                                                                 \
    if (abl == V) {                                                                                \
        auto k4 = spmm_blk<HALO, kNW, SH, D, kPasses, V>;                                          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL(k4, grid, dim3(kNW * 64), lds_bytes, s, a);                             \
        return sgp::check_launch("spmm_blk");                                                      \
    }
    SGP_ABL(1) SGP_ABL(4)
#undef SGP_ABL
#endif
    auto kern = spmm_blk<HALO, kNW, SH, D, kPasses>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_blk: LDS opt-in: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, grid, dim3(kNW * 64), lds_bytes, s, a);
    return sgp::check_launch("spmm_blk");
}

}  // namespace

extern "C" {

int32_t sgp_spmm_blk_max_union(void) { return kPasses * kNW * 4; }
int32_t sgp_spmm_blk_waves(void) { return kNW; }
int sgp_spmm_blk_tune(int32_t cfg) { if (cfg >= 0) g_blk_cfg = cfg; return 0; }

int sgp_spmm_blk_f32(const int32_t* uptr, const int32_t* ucol, const int32_t* usplit,
                     const int32_t* wptr, const int32_t* nsteps, const int32_t* soff, const float* sw,
                     const int32_t* rowmap,
                     int32_t n_tiles, int32_t waves, int32_t max_union,
                     const float* X, int64_t xrs, int64_t xbs,
                     const float* Xh, int64_t xhrs, int64_t xhbs, int32_t n_own,
                     float* Y, int64_t yrs, int64_t ybs,
                     int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                     sgp_stream_t stream) {
    SGP_REQUIRE(uptr && ucol && usplit && wptr && nsteps && soff && sw && rowmap && X && Y,
                "sgp_spmm_blk_f32: null pointer");
    SGP_REQUIRE(n_tiles >= 0 && n_rows >= 0 && batch >= 0 && max_union >= 0,
                "sgp_spmm_blk_f32: bad size");
    SGP_REQUIRE(waves == kNW, "sgp_spmm_blk_f32: the plan was built for %d waves, the kernel for %d", waves, kNW);
    {
        const long long own = Xh ? n_own : n_cols, far = Xh ? n_cols - n_own : 0;
        SGP_REQUIRE(n_cols >= 0 && own >= 0 && far >= 0 && own * xrs < (1ll << 30) && far * xhrs < (1ll << 30) &&
                    (long long)n_rows * yrs < (1ll << 30),
                    "sgp_spmm_blk_f32: row offsets exceed 32 bits (use sgp_spmm_csr_f32)");
    }
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    if (feat % 64 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_blk_f32: feat=%d is not a multiple of 64", feat);
    if (max_union > sgp_spmm_blk_max_union())
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_blk_f32: tile working set (%d rows) exceeds LDS (%d)",
                         max_union, sgp_spmm_blk_max_union());
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 && sgp::aligned16(X) &&
                sgp::aligned16(Y) && (!Xh || (xhrs % 4 == 0 && xhbs % 4 == 0 && sgp::aligned16(Xh))),
                "sgp_spmm_blk_f32: strides/pointers must be 16-byte aligned");
    BlkArgs a;
    a.uptr = uptr; a.ucol = ucol; a.usplit = usplit; a.wptr = wptr; a.nsteps = nsteps; a.soff = soff;
    a.sw = sw; a.rowmap = rowmap;
    a.n_tiles = n_tiles;
    a.src = Src2{X, xrs, xbs, Xh ? Xh : X, xhrs, xhbs, Xh ? n_own : 0x7fffffff};
    a.Y = Y; a.yrs = yrs; a.ybs = ybs;
    a.n_rows = n_rows; a.batch = batch; a.feat = feat;
    const int nft = feat / 64;
    long long want = (long long)batch * n_tiles * nft / 2048;
    int tc = (int)(want < 16 ? 16 : (want > blk_chunk_cap() ? blk_chunk_cap() : want));
    if (tc > batch) tc = batch;
    a.t_chunk = tc;
    a.n_tchunks = (batch + tc - 1) / tc;
    hipStream_t s = (hipStream_t)stream;
    switch (blk_cfg()) {
    case 1: return Xh ? launch_blk<true, 64, 6>(a, s) : launch_blk<false, 64, 6>(a, s);
    case 2: return Xh ? launch_blk<true, 56, 8>(a, s) : launch_blk<false, 56, 8>(a, s);
    default: return Xh ? launch_blk<true, 72, 4>(a, s) : launch_blk<false, 72, 4>(a, s);
    }
}

}  // extern "C"
