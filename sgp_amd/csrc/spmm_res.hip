// Register-resident form of the two-phase row-group SpMM (reference call site:
// lib/sgp_preprocessing.py:202, x = adj @ x).  gfx950 / wave64 only.  Same tiles, groups, column
// classes, two-phase LDS-DMA staging and v_mfma_f32_4x4x1_16b_f32 arithmetic as spmm_pipe; what
// differs is where a group's stream lives: the weights (one VGPR per 4 super-steps) and the
// per-lane LDS address of every staged row it reads (one VGPR per super-step) are loaded ONCE per
// workgroup into registers and stay there for the whole time chunk.  A super-step is then
//         s_waitcnt lgkmcnt(n) | 4 x v_mfma_f32_4x4x1_16b_f32 | ds_read_b128 (D super-steps ahead)
// with no VALU address arithmetic, no weight / offset reads from LDS and no read that depends on
// another read -- on gfx950 every VALU instruction and every VGPR write of an LDS return takes
// issue time from the fp32 matrix pipe (tools/ubench/res_loop2.hip: 42-46 cycles per 32 cycles of
// MFMAs, fold and store included, against ~55 for the spmm_pipe body).
// Registers hold the first SH super-steps of either range of a group (99 % of the ranges of the
// 100-NN target graph are shorter); what lies beyond is walked from an LDS copy of the stream the
// way spmm_pipe does (not software-pipelined: those groups mix rows of two distant clusters and
// are the slowest of their tile in any case).
//
// The compiler is kept out of the inner loop's memory scheduling on purpose: hipcc sinks a plain LDS
// load to its use across the scalar exit branches (read -> wait -> 4 MFMAs, serialised), so operand
// reads and their waits are inline asm.  LDS operations return in order, which makes the wait
// counts static (WAITN below).
#include "common.h"
#include <stdlib.h>

using sgp::f32x4;

namespace {

struct Src2 {
    const float* x;  long long xrs, xbs;
    const float* xh; long long xhrs, xhbs;
    int n_own;
    const int* pred; int pred_want;        // launch predicate (common.h)
    __device__ __forceinline__ bool skip() const { return pred != nullptr && pred[0] != pred_want; }
};

struct ResArgs {
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

// Super-steps per range held in registers: 24 (96 columns; 126 VGPRs with the halo source) / 28 when
// there is no halo pointer to carry.  On the 100-NN target graph 1.0 % / 0.0 % of the ranges are
// longer (17 % / 1 % of the tiles have one); with 20, 1.8 % / 30 % -- and a tile waits for its
// overflow walk, which is not software-pipelined: 12.7 -> 12.45 ms per 512 steps for 20 -> 24.
template <bool HALO> struct ResidentSteps { static constexpr int value = HALO ? 24 : 28; };

// cache policy bits of the staging reads (experiment builds: -DSGP_DMA_MOD='" sc0"' etc.)
#ifndef SGP_DMA_MOD
#define SGP_DMA_MOD ""
#endif
__device__ __forceinline__ void dma16_saddr(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" SGP_DMA_MOD
                 :: "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
__device__ __forceinline__ void dma16_vaddr(const void* vaddr, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" SGP_DMA_MOD
                 :: "v"(vaddr), "s"(lds_off) : "memory");
}

// NW waves per workgroup, G row groups per wave (tile = 4 * NW * G rows), operand ring D super-steps
// deep per group parity, PASSES x (4 NW) staged rows.  ABL: bit0 no staging DMA, bit2 staging
// always reads the chunk's first step, bit4 / bit5 every staged row set is read for 2 / 4 consecutive
// steps: 50 % / 75 % of the staging reads are forced L2 hits, bit7 per-wave s_memtime timeline of one
// workgroup (tools/timeline_res.py) (ablation builds).
template <bool HALO, int NW, int G, int D, int PASSES, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void spmm_res(ResArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SH = (G == 1) ? ResidentSteps<HALO>::value : 20;   // (two groups per wave: 20, the register budget of cfg 1)
    constexpr int WH = (SH + 3) / 4;                      // weight registers per range
    constexpr int GT = NW * G;
    constexpr int RPP = NW * 4;                           // staged rows per DMA pass

    if (a.src.skip()) return;
    const int nwg = a.n_tiles * a.n_tchunks;
    const int orig = blockIdx.x;
    const int qq = nwg >> 3, rr = nwg & 7, x8 = orig & 7;
    const int wg = (x8 < rr ? x8 * (qq + 1) : rr * (qq + 1) + (x8 - rr) * qq) + (orig >> 3);
    const int tile = wg % a.n_tiles;
    const int tchunk = wg / a.n_tiles;
    const int f_base = blockIdx.y * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = tid & 15;
    const int eg = tid >> 4;
    const int q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = a.uptr[tile];
    const int nU = a.uptr[tile + 1] - u0;
    const int uA = a.usplit[tile];

    const int t_begin = tchunk * a.t_chunk;
    const int t_end = min(a.batch, t_begin + a.t_chunk);
    if (t_begin >= t_end) return;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // ---- DMA bookkeeping: per-lane source offsets of the staged rows this lane feeds
    unsigned voff[PASSES];
    unsigned halo_mask = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int u = p * RPP + eg;
        const int c = (u < nU) ? a.ucol[u0 + u] : (nU > 0 ? a.ucol[u0] : 0);
        if (HALO && c >= a.src.n_own) {
            halo_mask |= 1u << p;
            voff[p] = (unsigned)((c - a.src.n_own) * (int)a.src.xhrs + f_base + li * 4) * 4u;
        } else {
            voff[p] = (unsigned)(c * (int)a.src.xrs + f_base + li * 4) * 4u;
        }
    }
    unsigned piecesA = 0, piecesB = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int r0 = p * RPP + wave * 4;
        if (r0 < uA) piecesA |= 1u << p;
        else if (r0 < nU) piecesB |= 1u << p;
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
        if constexpr ((ABL & 48) != 0) {                  // every row set is staged for 2 (16) / 4 (32) steps in a row
            const long long k = (xt - x_step0) / x_inc / ((ABL & 16) ? 2 : 4) * ((ABL & 16) ? 2 : 4);
            xt = x_step0 + k * x_inc; ht = h_step0 + k * h_inc;
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if constexpr ((ABL & 512) != 0) { if (p % 3 == 2) continue; }     // (a third fewer staging pieces)
            if constexpr ((ABL & 1024) != 0) { if (p % 2 == 1) continue; }    // (half the staging pieces)
            if (pieces & (1u << p)) {                     // scalar
                const unsigned dst = piece0 + (unsigned)p * (unsigned)(RPP * 256);
                if constexpr (HALO) {
                    const char* b = ((halo_mask >> p) & 1u) ? ht : xt;
                    dma16_vaddr(b + voff[p], __builtin_amdgcn_readfirstlane(dst));
                } else {
                    dma16_saddr(voff[p], xt, dst);
                }
            }
        }
    };

    // ---- the tile's stream -> LDS behind the stage (read only by ranges longer than SH)
    constexpr int kStageBytes = PASSES * RPP * 256;
    const int tile_q0 = a.gptr[tile * (2 * GT)], tile_q1 = a.gptr[tile * (2 * GT) + 2 * GT];
    const int tile_quads = tile_q1 - tile_q0;
    // (only a tile that HAS a longer range: 1 % of the 64-row tiles of the 100-NN graph at SH = 28)
    // (voted through the first word of the still unused stage: __syncthreads_or would add a static
    // LDS word to the 160 KB of dynamic LDS)
#ifndef SGP_RES_NO_VOTE
    int* vote = reinterpret_cast<int*>(lds);
    if (tid == 0) *vote = 0;
    __syncthreads();
    if (a.gsup[tile * (2 * GT) + (tid % (2 * GT))] > SH) *vote = 1;
    __syncthreads();
    const bool tile_has_long_range = *vote != 0;
    __syncthreads();
#else
    const bool tile_has_long_range = true;
#endif
    if ((ABL & 8) == 0 && tile_has_long_range) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.gw) + (long long)tile_q0 * 16;
        f32x4* dst = reinterpret_cast<f32x4*>(lds + kStageBytes);
        for (int i = tid; i < tile_quads * 16; i += NW * 64) dst[i] = src[i];
        const f32x4* isrc = reinterpret_cast<const f32x4*>(a.gidx) + (long long)tile_q0 * 4;
        f32x4* idst = reinterpret_cast<f32x4*>(lds + kStageBytes + tile_quads * 256);
        for (int i = tid; i < tile_quads * 4; i += NW * 64) idst[i] = isrc[i];
    }

    // (ablation 8: per-lane source offsets of EVERY piece, for the wave that issues all staging reads;
    // the table takes the place of the LDS copy of the stream, long ranges are cut at SH)
    unsigned* dma_tab = reinterpret_cast<unsigned*>(lds + kStageBytes);
    if constexpr ((ABL & 8) != 0) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PASSES; ++p) dma_tab[p * (NW * 64) + tid] = voff[p];
    }
    // ablation 8: the last wave issues every piece of a segment (the others none)
    auto dma_all = [&](const char* xt, bool seg_b) {
        if constexpr ((ABL & 8) != 0) {
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                unsigned vo[NW];
#pragma unroll
                for (int v = 0; v < NW; ++v) vo[v] = dma_tab[p * (NW * 64) + v * 64 + lane];
#pragma unroll
                for (int v = 0; v < NW; ++v) {
                    const int r0 = p * RPP + v * 4;
                    const bool in_a = r0 < uA, in_b = r0 >= uA && r0 < nU;
                    if (seg_b ? in_b : in_a)
                        dma16_saddr(vo[v], xt, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)v * 1024u + (unsigned)p * (unsigned)(RPP * 256)));
                }
            }
        }
    };
    // ---- the wave's stream -> registers (once per workgroup)
    // weights: lane (q, b = li >> 2, i = li & 3) holds row i's weight for class q's column in
    // super-step 4 p + b (gw is stored one float per lane and quad: the MFMA of super-step s takes
    // block s & 3 of its class, cbsz = 2 / abid); addresses: lane (q, li) holds the LDS byte address
    // of chunk li of class q's staged row in super-step s.  Padding reads row 0 with weight 0.
    unsigned addr[2][G][SH];
    float w[2][G][WH];
    int n[2][G];
    int qrel[2][G];                                        // first quad of the range, relative to the tile
    unsigned yoff[G];
    bool has_row[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int grp = (tile * GT + wave * G + g) * 2;
        const int q0 = __builtin_amdgcn_readfirstlane(a.gptr[grp]);
        const int q1 = __builtin_amdgcn_readfirstlane(a.gptr[grp + 1]);
        const int q2 = __builtin_amdgcn_readfirstlane(a.gptr[grp + 2]);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int qb = ph ? q1 : q0, qe = ph ? q2 : q1;
            n[ph][g] = __builtin_amdgcn_readfirstlane(a.gsup[grp + ph]);
            if constexpr ((ABL & 64) != 0) { if (n[ph][g] > SH) n[ph][g] = SH; if (wave == NW - 1) n[ph][g] = 0; }
            qrel[ph][g] = qb - tile_q0;
#pragma unroll
            for (int p = 0; p < WH; ++p)
                w[ph][g][p] = (qb + p < qe) ? a.gw[(long long)(qb + p) * 64 + lane] : 0.f;
#pragma unroll
            for (int s = 0; s < SH; ++s) {
                const int quad = qb + (s >> 2);
                const unsigned off = (quad < qe) ? (unsigned)a.gidx[(long long)quad * 16 + q * 4 + (s & 3)] : 0u;
                addr[ph][g][s] = lds0 + off + li * 16;
            }
        }
        const int row = a.rowmap[tile * (GT * 4) + (wave * G + g) * 4 + q];
        has_row[g] = row >= 0;
        yoff[g] = (unsigned)((long long)(row < 0 ? 0 : row) * a.yrs + f_base + li * 4) * 4u;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
            for (int s = 0; s < SH; ++s) asm volatile("" : "+v"(addr[ph][g][s]));
#pragma unroll
            for (int p = 0; p < WH; ++p) asm volatile("" : "+v"(w[ph][g][p]));
        }
    }
    char* y_step = reinterpret_cast<char*>(a.Y + (long long)t_begin * a.ybs);
    const long long y_inc = a.ybs * 4;

    f32x4 ring[2][D];
    f32x4 acc[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = acc[g][0]; acc[g][2] = acc[g][0]; acc[g][3] = acc[g][0]; }

    // result of a step held in group g's accumulators -> its rows: the 4 column classes are summed,
    // class q keeps row q (streamed store: it must not displace staged rows from L2)
    auto emit = [&](int g, char* ys) {
        f32x4 out;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            auto p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[g][m].x), __float_as_uint(acc[g][m].y), false, false);
            auto p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[g][m].z), __float_as_uint(acc[g][m].w), false, false);
            const float r01 = __uint_as_float(p01[0]) + __uint_as_float(p01[1]);
            const float r23 = __uint_as_float(p23[0]) + __uint_as_float(p23[1]);
            auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(r01), __float_as_uint(r23), false, false);
            out[m] = __uint_as_float(h[0]) + __uint_as_float(h[1]);
        }
        if (has_row[g]) __builtin_nontemporal_store(out, reinterpret_cast<f32x4*>(ys + yoff[g]));
    };

    // super-steps SH .. n-1 of a long range, from the LDS copy of the stream (whole quads: the
    // padding of the last one has weight 0)
    typedef const __attribute__((address_space(3))) f32x4* lds_f4_t;
    typedef const __attribute__((address_space(3))) float* lds_f1_t;
    typedef const __attribute__((address_space(3))) unsigned* lds_u1_t;
    auto overflow = [&](int g, int qr, int nsteps) {
        const unsigned wl = lds0 + kStageBytes + (unsigned)qr * 256u + lane * 4;
        const unsigned il = lds0 + kStageBytes + (unsigned)tile_quads * 256u + (unsigned)qr * 64u + q * 16;
        for (int c = WH; c < ((nsteps + 3) >> 2); ++c) {
            const float wv = *(lds_f1_t)(wl + c * 256);
            f32x4 xs[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) xs[b] = *(lds_f4_t)(lds0 + *(lds_u1_t)(il + c * 64 + b * 4) + li * 16);
            acc[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].x, acc[g][0], 2, 0, 0);
            acc[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].y, acc[g][1], 2, 0, 0);
            acc[g][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].z, acc[g][2], 2, 0, 0);
            acc[g][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[0].w, acc[g][3], 2, 0, 0);
            acc[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].x, acc[g][0], 2, 1, 0);
            acc[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].y, acc[g][1], 2, 1, 0);
            acc[g][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].z, acc[g][2], 2, 1, 0);
            acc[g][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[1].w, acc[g][3], 2, 1, 0);
            acc[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[2].x, acc[g][0], 2, 2, 0);
            acc[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[2].y, acc[g][1], 2, 2, 0);
            acc[g][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[2].z, acc[g][2], 2, 2, 0);
            acc[g][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[2].w, acc[g][3], 2, 2, 0);
            acc[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[3].x, acc[g][0], 2, 3, 0);
            acc[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[3].y, acc[g][1], 2, 3, 0);
            acc[g][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[3].z, acc[g][2], 2, 3, 0);
            acc[g][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv, xs[3].w, acc[g][3], 2, 3, 0);
        }
    };

#define SGP_RD(P_, G_, S_) asm volatile("ds_read_b128 %0, %1" : "=v"(ring[(G_) & 1][(S_) % D]) : "v"(addr[P_][G_][S_]))
    // LDS reads issued after r(g, s) when super-step (g, s) starts: its ring refills (s >= D), or,
    // for the D super-steps requested ahead, the rest of that request + the look-ahead request of
    // group g + 1 + the refills of super-steps 0 .. s-1 (the previous group's refills in between
    // only make the count conservative)
#define SGP_WAITN(G_, S_) ((S_) >= D ? ((SH - 1 - (S_)) < (D - 1) ? (SH - 1 - (S_)) : (D - 1)) : (D - 1 + ((G_) + 1 < G ? D : 0)))
#define SGP_WT(G_, S_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[(G_) & 1][(S_) % D]) : "n"(SGP_WAITN(G_, S_)))
#define SGP_MF(ACC_, W_, X_, AB_) __builtin_amdgcn_mfma_f32_4x4x1f32(W_, X_, ACC_, 2, AB_, 0)
#define SGP_SLOT4(P_, G_, S_, AB_, FIRST_)                                                         \
    {                                                                                              \
        const f32x4 x = ring[(G_) & 1][(S_) % D];                                                  \
        const float wv = w[P_][G_][(S_) >> 2];                                                     \
        if (FIRST_) {                                                                              \
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};                                                  \
            acc[G_][0] = SGP_MF(z, wv, x.x, AB_); acc[G_][1] = SGP_MF(z, wv, x.y, AB_);            \
            acc[G_][2] = SGP_MF(z, wv, x.z, AB_); acc[G_][3] = SGP_MF(z, wv, x.w, AB_);            \
        } else {                                                                                   \
            acc[G_][0] = SGP_MF(acc[G_][0], wv, x.x, AB_); acc[G_][1] = SGP_MF(acc[G_][1], wv, x.y, AB_); \
            acc[G_][2] = SGP_MF(acc[G_][2], wv, x.z, AB_); acc[G_][3] = SGP_MF(acc[G_][3], wv, x.w, AB_); \
        }                                                                                          \
    }
#define SGP_SLOT(P_, G_, S_, FIRST_)                                                               \
    if (((S_) & 3) == 0) SGP_SLOT4(P_, G_, S_, 0, FIRST_) else if (((S_) & 3) == 1) SGP_SLOT4(P_, G_, S_, 1, FIRST_) \
    else if (((S_) & 3) == 2) SGP_SLOT4(P_, G_, S_, 2, FIRST_) else SGP_SLOT4(P_, G_, S_, 3, FIRST_)
    // One phase = region P_ of the stage.  The first D super-steps of group g + 1 are requested at
    // the start of group g (other ring parity), so only a phase's first group starts cold, and that
    // wait is covered by the fold + store of the previous step (MID_).  In phase A a group with
    // columns restarts its accumulators through the first MFMAs (C = 0); one without is cleared.
#define SGP_PHASE(P_, MID_)                                                                        \
    { _Pragma("unroll") for (int s = 0; s < D; ++s) SGP_RD(P_, 0, s); }                            \
    _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                \
        if (g + 1 < G) { _Pragma("unroll") for (int s = 0; s < D; ++s) SGP_RD(P_, g + 1, s); }     \
        if (g == 0) { MID_ }                                                                       \
        if ((P_) == 1 && g > 0) emit(g - 1, y_step);                                               \
        if (n[P_][g] > 0) {                                                                        \
            _Pragma("unroll") for (int s = 0; s < SH; ++s) {                                       \
                SGP_WT(g, s);                                                                      \
                SGP_SLOT(P_, g, s, (P_) == 0 && s == 0)                                            \
                if (s + D < SH) SGP_RD(P_, g, s + D);                                              \
                if (s + 1 == n[P_][g]) break;                                                      \
            }                                                                                      \
            if (n[P_][g] > SH) overflow(g, qrel[P_][g], n[P_][g]);                                 \
        } else if ((P_) == 0) {                                                                    \
            acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = acc[g][0]; acc[g][2] = acc[g][0]; acc[g][3] = acc[g][0]; \
        }                                                                                          \
    }

    // debug timeline (ABL & 128): workgroup 777 records s_memtime at 10 points of 4 steps
    auto stamp = [&](int t, int point) {
        if constexpr ((ABL & 128) != 0) {
            const int ts = t - t_begin - 8;
            if (wg == 777 && ts >= 0 && ts < 4) {
                const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
                if (lane == 0) a.dbg[((ts * NW + wave) * 10 + point)] = now;
            }
        }
    };

    __syncthreads();
    constexpr bool kOneIssuer = (ABL & 8) != 0;
    if constexpr (kOneIssuer) { if (wave == NW - 1) dma_all(x_step, false); }
    else dma_segment(x_step, h_step, piecesA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool dma_first = kOneIssuer ? false : wave >= NW / 2;
    for (int t = t_begin; t < t_end; ++t) {
        // the range lengths are re-made opaque every step: otherwise hipcc hoists all exit
        // comparisons out of the time loop as 64-bit masks and spills them to VGPR lanes
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("" : "+s"(n[0][g]), "+s"(n[1][g]));
        // ---- phase A: region A holds step t once every wave's pieces have landed
        stamp(t, 0);
        asm volatile("s_barrier" ::: "memory");
        stamp(t, 1);
        // the refill of the other region is issued first by the younger half of the waves (they
        // would wait for the matrix pipe anyway) and after their super-steps by the older half
        if (dma_first) dma_segment(x_step, h_step, piecesB);
        if constexpr (kOneIssuer) { if (wave == NW - 1) dma_all(x_step, true); }
        stamp(t, 2);
        SGP_PHASE(0, if (t > t_begin) emit(G - 1, y_step - y_inc);)
        stamp(t, 3);
        if (!kOneIssuer && !dma_first) dma_segment(x_step, h_step, piecesB);
        // ---- phase B
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(t, 4);
        if constexpr ((ABL & 256) == 0) asm volatile("s_barrier" ::: "memory");   // (ablation 256: one barrier per step)
        stamp(t, 5);
        if (dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);
        if constexpr (kOneIssuer) { if (wave == NW - 1 && t + 1 < t_end) dma_all(x_step + x_inc, false); }
        stamp(t, 6);
        SGP_PHASE(1, )
        stamp(t, 7);
        if (!kOneIssuer && !dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);
        // this wave's pieces of A(t+1) (and its stores) retired before the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(t, 8);
        x_step += x_inc; h_step += h_inc; y_step += y_inc;
    }
    emit(G - 1, y_step - y_inc);
#undef SGP_PHASE
#undef SGP_SLOT
#undef SGP_SLOT4
#undef SGP_MF
#undef SGP_WT
#undef SGP_WAITN
#undef SGP_RD
}

#ifdef SGP_ABLATION
unsigned* res_dbg_buffer() {
    static unsigned* p = nullptr;
    if (!p) { (void)hipMalloc(&p, 4 * 16 * 10 * sizeof(unsigned)); (void)hipMemset(p, 0, 4 * 16 * 10 * sizeof(unsigned)); }
    return p;
}
#endif

int g_res_cfg = -1;
int res_cfg() {                                            // 0: 16 waves x 1 group, 1: 8 waves x 2 groups
    if (g_res_cfg < 0) g_res_cfg = (int)sgp::tune("res_cfg", 0);
    return g_res_cfg;
}
int res_chunk_cap() {
    static int v = -1;
    if (v < 0) { v = (int)sgp::tune("spmm_chunk", 32); if (v < 1) v = 32; }
    return v;
}

template <bool HALO, int NW, int G, int D, int PASSES>
int launch_res(const ResArgs& a, hipStream_t s) {
    const size_t lds_bytes = 160 * 1024;
    dim3 grid((unsigned)(a.n_tiles * a.n_tchunks), a.feat / 64);
#ifdef SGP_ABLATION
    static int abl = -1;
    if (abl < 0) abl = (int)sgp::tune("abl", 0);
#define SGP_ABL(V)                                                                                 \
    if (abl == V) {                                                                                \
        auto k4 = spmm_res<HALO, NW, G, D, PASSES, V>;                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL(k4, grid, dim3(NW * 64), lds_bytes, s, a);                              \
        return sgp::check_launch("spmm_res");                                                      \
    }
    SGP_ABL(1) SGP_ABL(4) SGP_ABL(16) SGP_ABL(32) SGP_ABL(128) SGP_ABL(129) SGP_ABL(132) SGP_ABL(64) SGP_ABL(72) SGP_ABL(256) SGP_ABL(512) SGP_ABL(768) SGP_ABL(257) SGP_ABL(1024)
#undef SGP_ABL
#endif
    auto kern = spmm_res<HALO, NW, G, D, PASSES>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_res: LDS opt-in: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, s, a);
    return sgp::check_launch("spmm_res");
}

}  // namespace

#ifdef SGP_ABLATION
extern "C" int sgp_spmm_res_debug_read(unsigned* host) {
    return (int)hipMemcpy(host, res_dbg_buffer(), 4 * 16 * 10 * sizeof(unsigned), hipMemcpyDeviceToHost);
}
#endif

extern "C" {

int32_t sgp_spmm_res_max_union(void) { return 7 * 64; }
int32_t sgp_spmm_res_max_quads(void) { return (160 * 1024 - 7 * 64 * 256) / (256 + 64); }
int sgp_spmm_res_tune(int32_t cfg) { if (cfg >= 0) g_res_cfg = cfg; return 0; }

int sgp_spmm_res_f32(const int32_t* uptr, const int32_t* ucol, const int32_t* usplit,
                     const int32_t* gptr, const int32_t* gsup, const int32_t* gidx, const float* gw,
                     const int32_t* rowmap,
                     int32_t n_tiles, int32_t max_union, int32_t max_tile_quads,
                     const float* X, int64_t xrs, int64_t xbs,
                     const float* Xh, int64_t xhrs, int64_t xhbs, int32_t n_own,
                     float* Y, int64_t yrs, int64_t ybs,
                     int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                     const int32_t* pred, int32_t run_if, sgp_stream_t stream) {
    const sgp::Predicate pr{pred, run_if};
    SGP_REQUIRE(uptr && ucol && usplit && gptr && gsup && gidx && gw && rowmap && X && Y,
                "sgp_spmm_res_f32: null pointer");
    SGP_REQUIRE(n_tiles >= 0 && n_rows >= 0 && batch >= 0 && max_union >= 0 && max_tile_quads >= 0,
                "sgp_spmm_res_f32: bad size");
    {
        const long long own = Xh ? n_own : n_cols, far = Xh ? n_cols - n_own : 0;
        SGP_REQUIRE(n_cols >= 0 && own >= 0 && far >= 0 && own * xrs < (1ll << 30) && far * xhrs < (1ll << 30) &&
                    (long long)n_rows * yrs < (1ll << 30),
                    "sgp_spmm_res_f32: row offsets exceed 32 bits (use sgp_spmm_csr_f32)");
    }
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    if (feat % 64 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_res_f32: feat=%d is not a multiple of 64", feat);
    if (max_union > sgp_spmm_res_max_union() || max_tile_quads > sgp_spmm_res_max_quads())
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_res_f32: tile working set (%d rows, %d quads) exceeds LDS (%d, %d)",
                         max_union, max_tile_quads, sgp_spmm_res_max_union(), sgp_spmm_res_max_quads());
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 && sgp::aligned16(X) &&
                sgp::aligned16(Y) && (!Xh || (xhrs % 4 == 0 && xhbs % 4 == 0 && sgp::aligned16(Xh))),
                "sgp_spmm_res_f32: strides/pointers must be 16-byte aligned");
    ResArgs a;
    a.uptr = uptr; a.ucol = ucol; a.usplit = usplit; a.gptr = gptr; a.gsup = gsup; a.gidx = gidx; a.gw = gw;
    a.rowmap = rowmap;
    a.n_tiles = n_tiles;
    a.src = Src2{X, xrs, xbs, Xh ? Xh : X, xhrs, xhbs, Xh ? n_own : 0x7fffffff, pr.flag, pr.want};
    a.Y = Y; a.yrs = yrs; a.ybs = ybs;
    a.n_rows = n_rows; a.batch = batch; a.feat = feat;
    const int nft = feat / 64;
    long long want = (long long)batch * n_tiles * nft / 4096;
    int tc = (int)(want < 16 ? 16 : (want > res_chunk_cap() ? res_chunk_cap() : want));
    if (tc > batch) tc = batch;
    a.t_chunk = tc;
    a.n_tchunks = (batch + tc - 1) / tc;
    hipStream_t s = (hipStream_t)stream;
    a.dbg = nullptr;
#ifdef SGP_ABLATION
    a.dbg = res_dbg_buffer();
#endif
    if (res_cfg() == 1)
        return Xh ? launch_res<true, 8, 2, 4, 14>(a, s) : launch_res<false, 8, 2, 4, 14>(a, s);
    return Xh ? launch_res<true, 16, 1, 4, 7>(a, s) : launch_res<false, 16, 1, 4, 7>(a, s);
}

}  // extern "C"
