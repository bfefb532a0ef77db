// Mixed dense / sparse form of the two-phase row-group SpMM (reference call site:
// lib/sgp_preprocessing.py:200-203, x = adj @ x per hop).  gfx950 / wave64 only.
//
// Tiles, staged rows, the parity cut into segments A | B, LDS-DMA staging and the register-resident
// 4x4x1 row-group stream are those of spmm_res.hip.  What is new (plan: sgp_amd/mixplan.py):
//   * the 16 row groups of a tile form 4 blocks of 16 rows; the columns that (nearly) all 4 groups
//     of a block use -- about half of the (group, column) pairs of a 100-NN graph -- leave the
//     4x4x1 stream and go through v_mfma_f32_16x16x4_f32: 16 rows x 4 columns x 16 features per
//     instruction, ONE 4-byte LDS read per lane and no cross-lane fold per 1024 FMAs (the 4x4x1 form:
//     one 16-byte read per 1024 FMAs spread over 4 instructions of 8 cycles, whose issue slots the
//     waits, reads and exit tests compete for);
//   * wave w keeps its row group w (4 rows x 64 features, sparse part) AND computes the dense part
//     of block w / 4 for the feature quarter w % 4 (16 rows x 16 features).  The two accumulator
//     layouts do not match, so the dense result of a step crosses a 16 KB LDS slab: written at the
//     end of phase B (4 x ds_write_b32), read as ONE ds_read_b128 by the wave that folds and stores the
//     rows at the top of the next step -- the two barriers of a step already order the slab.
// Both streams live in registers for the whole time chunk, in ONE file of NF = WH + SH + 2 DH
// registers per phase (struct Regs): the weights (one register per 4 super-steps) and per-lane LDS
// addresses of the first SH sparse super-steps from the bottom, the dense instructions (address +
// weight per lane) from the TOP downwards.  What a block does not need for dense instructions serves
// its groups' longer sparse ranges as "extension quads" of 5 registers (4 addresses + 1 weight) that
// grow upwards from the resident part: a block without dense columns (a tile that straddles two
// distant clusters) has 28 resident super-steps like spmm_res, a typical one 12 + 10 instructions.
// What still does not fit is read from the plan arrays in global memory (L2) every step.  LDS operations return in order, so every
// wait count is static (see the W_* macros); reads that a loop issues past its end ("garbage") only
// ever use registers that hold valid LDS addresses (Regs: weights never share a position's parity
// with a dense address, and the extension keeps one quad of slack below the dense instructions).
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

struct MixArgs {
    const int* uptr; const int* ucol; const int* usplit;
    const int* gptr; const int* gsup; const int* gidx; const float* gw; const int* rowmap;
    const int* dptr; const int* didx; const float* dw;
    int n_tiles;
    Src2 src;
    float* Y; long long yrs, ybs;
    int n_rows, batch, feat;
    int t_chunk, n_tchunks;
    int mode;
    unsigned* dbg;
};

constexpr int NW = 16;                                    // waves per workgroup = row groups per tile
constexpr int PASSES = 7;                                 // x 64 staged rows
constexpr int kStageBytes = PASSES * NW * 4 * 256;        // 114 688

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

// Register file of one phase (positions are compile-time constants everywhere).
template <int SH, int DH> struct Regs {
    static constexpr int WH = (SH + 3) / 4;
    static constexpr int NF = WH + SH + 2 * DH;
    static constexpr int KX = (2 * DH) / 5;               // extension quads when no dense instruction is resident
    static constexpr int SHX = SH + 4 * KX;               // sparse super-steps addressable from registers
    // sparse super-step s: address / weight position
    static constexpr int ext_base(int k) { return WH + SH + 5 * k; }
    static constexpr int ext_w(int k) { return ((ext_base(k) & 1) != (NF & 1)) ? ext_base(k) : ext_base(k) + 1; }
    static constexpr int ext_a(int k, int j) { return ext_base(k) + j >= ext_w(k) ? ext_base(k) + j + 1 : ext_base(k) + j; }
    static constexpr int sa(int s) { return s < SH ? WH + s : ext_a((s - SH) / 4, (s - SH) % 4); }
    static constexpr int sw(int s) { return s < SH ? s / 4 : ext_w((s - SH) / 4); }
    // dense instruction m: address (parity of NF: never an extension weight) / weight
    static constexpr int da(int m) { return NF - 2 * (m + 1); }
    static constexpr int dw(int m) { return NF - 2 * (m + 1) + 1; }
};

// SH / DH: resident super-steps / dense instructions per phase; D / DD: operand rings.
// ABL (ablation builds): bit0 no staging DMA, bit1 no dense instructions, bit2 no sparse super-steps,
// bit7 per-wave s_memtime timeline of one workgroup.
template <bool HALO, int SH, int DH, int D, int DD, bool ILV, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void spmm_mix(MixArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using R = Regs<SH, DH>;
    constexpr int WH = R::WH, NF = R::NF, SHX = R::SHX;
    constexpr int RPP = NW * 4;

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

    // ---- DMA bookkeeping (as spmm_res): per-lane source offsets of the staged rows this lane feeds
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
        if constexpr ((ABL & 2048) != 0) { xt = x_step0; ht = h_step0; }   // (ablation: every staging read hits the L2)
        if constexpr ((ABL & 4096) != 0) {                                    // (ablation: every row set staged for 2 steps in a row: half hit)
            const long long kk = (xt - x_step0) / x_inc / 2 * 2;
            xt = x_step0 + kk * x_inc; ht = h_step0 + kk * h_inc;
        }
        if (a.mode & 16) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if constexpr ((ABL & 512) != 0) { if (p % 3 == 2) continue; }     // (ablation: a third fewer staging pieces)
            if constexpr ((ABL & 1024) != 0) { if (p % 2 == 1) continue; }    // (ablation: half the staging pieces)
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
        if (a.mode & 16) __builtin_amdgcn_s_setprio(0);
    };

    // ---- this wave's streams -> registers: sparse group `wave` (layout of gw / gidx: spmm_res.hip), dense
    // block wave / 4 for feature quarter wave % 4.  Dense A operand: lane 16 k + i = weight of (row i
    // of the block, column k); B operand: lane 16 k + j = feature 16 fq + j of column k's staged row;
    // D: lane 16 g + j, register r = (row 4 g + r, feature 16 fq + j)
    const int tile_q0 = a.gptr[tile * (2 * NW)];
    const int rb = wave >> 2, fq = wave & 3;
    unsigned F[2][NF];
    int n[2], nl[2], nd[2], kk[2], qrel[2];
    {
        const int grp = (tile * NW + wave) * 2;
        const int q0 = __builtin_amdgcn_readfirstlane(a.gptr[grp]);
        const int q1 = __builtin_amdgcn_readfirstlane(a.gptr[grp + 1]);
        const int q2 = __builtin_amdgcn_readfirstlane(a.gptr[grp + 2]);
        const int db = (tile * 4 + rb) * 2;
        const int d0 = __builtin_amdgcn_readfirstlane(a.dptr[db]);
        const int d1 = __builtin_amdgcn_readfirstlane(a.dptr[db + 1]);
        const int d2 = __builtin_amdgcn_readfirstlane(a.dptr[db + 2]);
        const unsigned pad = lds0 + li * 16;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int qb = ph ? q1 : q0, qe = ph ? q2 : q1;
            const int b = ph ? d1 : d0, e = ph ? d2 : d1;
            n[ph] = __builtin_amdgcn_readfirstlane(a.gsup[grp + ph]);
            nd[ph] = min(e - b, DH);
            if constexpr ((ABL & 4) != 0) n[ph] = 0;
            if constexpr ((ABL & 2) != 0) nd[ph] = 0;
            qrel[ph] = qb - tile_q0;
            // extension quads that stay clear of the dense instructions, one of them as slack for
            // the reads issued past the end of the range
            const int free_q = (2 * DH - 2 * nd[ph]) / 5;
            const int need_q = (n[ph] + D - SH + 3) >> 2;
            nl[ph] = n[ph] <= SH ? n[ph] : (need_q <= free_q ? n[ph] : SH + 4 * max(free_q - 1, 0));
            kk[ph] = max(nd[ph], nl[ph]);                 // iterations of the interleaved loop
            // resident part: weights of quads 0 .. WH-1, addresses of super-steps 0 .. SH-1
#pragma unroll
            for (int p = 0; p < WH; ++p)
                F[ph][p] = __float_as_uint(qb + p < qe ? a.gw[(long long)(qb + p) * 64 + lane] : 0.f);
#pragma unroll
            for (int s = 0; s < SH; ++s) {
                const int quad = qb + (s >> 2);
                const unsigned off = (quad < qe) ? (unsigned)a.gidx[(long long)quad * 16 + q * 4 + (s & 3)] : 0u;
                F[ph][WH + s] = lds0 + off + li * 16;
            }
            // shared part, one assignment per position: dense instruction (from the top), else
            // extension quad of the sparse range (from the bottom), else a harmless address
#pragma unroll
            for (int pos = WH + SH; pos < NF; ++pos) {
                const int m = (NF - 1 - pos) / 2;                     // dense instruction of this position
                const bool m_addr = ((NF - pos) & 1) == 0;
                const int k = (pos - (WH + SH)) / 5;                  // extension quad of this position
                const bool k_in = k < R::KX;
                const bool k_w = k_in && pos == R::ext_w(k);
                const int j = pos - R::ext_base(k) - (pos > R::ext_w(k) ? 1 : 0);
                const int s = SH + 4 * k + (k_w ? 0 : j);             // sparse super-step of this position
                const int quad = qb + (s >> 2);
                unsigned v = pad;
                if (k_in && quad < qe && s < nl[ph]) {
                    v = k_w ? __float_as_uint(a.gw[(long long)quad * 64 + lane])
                            : lds0 + (unsigned)a.gidx[(long long)quad * 16 + q * 4 + (s & 3)] + li * 16;
                }
                if (m < nd[ph]) {
                    v = m_addr ? lds0 + (unsigned)a.didx[(long long)(b + m) * 4 + q] + fq * 64 + li * 4
                               : __float_as_uint(a.dw[(long long)(b + m) * 64 + lane]);
                }
                F[ph][pos] = v;
            }
        }
    }
    const int row = a.rowmap[tile * (NW * 4) + wave * 4 + q];
    const bool has_row = row >= 0;
    const unsigned yoff = (unsigned)((long long)(row < 0 ? 0 : row) * a.yrs + f_base + li * 4) * 4u;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int i = 0; i < NF; ++i) asm volatile("" : "+v"(F[ph][i]));
    }

    // slab addresses: this wave's dense results (write) / its rows' dense sums (read)
    const unsigned slab_w = lds0 + kStageBytes + (unsigned)((16 * rb + 4 * q) * 256 + (16 * fq + li) * 4);
    const unsigned slab_r = lds0 + kStageBytes + (unsigned)((4 * wave + q) * 256 + li * 16);

    char* y_step = reinterpret_cast<char*>(a.Y + (long long)t_begin * a.ybs);
    const long long y_inc = a.ybs * 4;

    f32x4 ring[D];
    float dring[DD];
#pragma unroll
    for (int i = 0; i < D; ++i) ring[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < DD; ++i) dring[i] = 0.f;
    f32x4 acc[4];
    f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; acc[2] = acc[0]; acc[3] = acc[0];

    // result of a step: the 4 column classes are summed, class q keeps row q, the block's dense sums of
    // that row come from the slab (streamed store: it must not displace staged rows from L2)
    auto emit = [&](char* ys) {
        if (a.mode & 8) __builtin_amdgcn_s_setprio(3);
        f32x4 out;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            auto p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[m].x), __float_as_uint(acc[m].y), false, false);
            auto p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[m].z), __float_as_uint(acc[m].w), false, false);
            const float r01 = __uint_as_float(p01[0]) + __uint_as_float(p01[1]);
            const float r23 = __uint_as_float(p23[0]) + __uint_as_float(p23[1]);
            auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(r01), __float_as_uint(r23), false, false);
            out[m] = __uint_as_float(h[0]) + __uint_as_float(h[1]);
        }
        // the block's dense sums of this row: read AFTER the fold (into registers the accumulators just
        // freed; the operand ties the read behind the fold), everything older has returned by then
        f32x4 slab;
        asm volatile("ds_read_b128 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(slab), "+v"(out) : "v"(slab_r));
        out += slab;
        if (has_row) __builtin_nontemporal_store(out, reinterpret_cast<f32x4*>(ys + yoff));
        if (a.mode & 8) __builtin_amdgcn_s_setprio(0);
    };

    // super-steps of a range that the registers do not hold (a group that mixes rows of distant
    // clusters AND sits in a block with dense columns: < 1 % of the ranges of a k-NN graph), straight
    // from the plan arrays: the same few hundred bytes every step, L2-resident
    typedef const __attribute__((address_space(3))) f32x4* lds_f4_t;
    auto overflow = [&](int qr, int c0, int nsteps) {
        for (int c = c0; c < ((nsteps + 3) >> 2); ++c) {
            int qd32 = __builtin_amdgcn_readfirstlane(tile_q0 + qr + c);            // uniform: scalar base + lane offset
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(qd32) :: "memory");
            const long long qd = qd32;
            const float wv = (a.gw + qd * 64)[lane];
            const int* iq = a.gidx + qd * 16;
            f32x4 xs[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) xs[b] = *(lds_f4_t)(size_t)(lds0 + (unsigned)iq[q * 4 + b] + li * 16);
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

    // ---- LDS traffic of one phase, in issue order (everything inline asm, returns in order):
        //   S0 .. S(D-1)   first D sparse operand reads
    //   N0 .. N(DD-1)  first DD dense operand reads
    //   dense loop m:  wait N(m) | mfma 16x16x4 | issue N(m + DD)
    //   [W0 .. W3]     phase B only: dense results -> slab
    //   sparse loop s: wait S(s) | 4 mfma 4x4x1 | issue S(s + D)
    // A wait lgkmcnt(k) is safe for an operation when at least k operations are ALWAYS issued after it
    // before the wait; reads of a loop that exits early stay in flight as garbage and only make later
    // waits conservative (they are older than anything those waits are for).
// (operand registers are read-write operands of every read: a loop that exits early leaves reads in
// flight, and a destination the compiler considered dead would be handed to some temporary -- seen:
// the flag of a DMA branch -- that the late return then overwrites.  "+v" keeps ring / dring live
// across the whole time loop, so nothing else ever lives in them.)
#define SGP_RD(P_, S_) asm volatile("ds_read_b128 %0, %1" : "+v"(ring[(S_) % D]) : "v"(F[P_][R::sa(S_)]))
#define SGP_DRD(P_, M_) asm volatile("ds_read_b32 %0, %1" : "+v"(dring[(M_) % DD]) : "v"(F[P_][R::da(M_)]))
#define W_DENSE(M_) ((DH - 1 - (M_)) < (DD - 1) ? (DH - 1 - (M_)) : (DD - 1))
// (sparse: a range that ends inside the resident part issues no read past SH - 1, a longer one none
// past SHX - 1; the count assumes the shorter tail, which is the safe side)
#define W_TAIL(S_) ((S_) < SH ? ((SH - 1 - (S_)) < (D - 1) ? (SH - 1 - (S_)) : (D - 1)) \
                              : ((SHX - 1 - (S_)) < (D - 1) ? (SHX - 1 - (S_)) : (D - 1)))
#define W_SPARSE(P_, S_) ((S_) >= D ? W_TAIL(S_) : (D - 1 - (S_)) + DD + ((P_) == 1 ? 4 : 0) + ((S_) < (SH - D) ? (S_) : (SH - D)))
#define SGP_WT(P_, S_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[(S_) % D]) : "n"(W_SPARSE(P_, S_)))
#define SGP_DWT(M_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dring[(M_) % DD]) : "n"(W_DENSE(M_)))
#define SGP_MF(ACC_, W_, X_, AB_) __builtin_amdgcn_mfma_f32_4x4x1f32(W_, X_, ACC_, 2, AB_, 0)
#define SGP_SLOT4(P_, S_, AB_, FIRST_)                                                             \
    {                                                                                              \
        const f32x4 x = ring[(S_) % D];                                                            \
        const float wv = __uint_as_float(F[P_][R::sw(S_)]);                                        \
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
#define SGP_PHASE(P_, PRE_, MID_)                                                                  \
    { _Pragma("unroll") for (int s = 0; s < D; ++s) SGP_RD(P_, s); }                               \
    { _Pragma("unroll") for (int m = 0; m < DD; ++m) SGP_DRD(P_, m); }                             \
    { PRE_ }                                                                                       \
    if ((P_) == 0) stamp(t, 9);                                                                    \
    if (nd[P_] > 0) {                                                                              \
        _Pragma("unroll") for (int m = 0; m < DH; ++m) {                                           \
            SGP_DWT(m);                                                                            \
            if ((P_) == 0 && m == 0) {                                                             \
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};                                              \
                dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(F[P_][R::dw(m)]), dring[m % DD], z, 0, 0, 0); \
            } else {                                                                               \
                dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(F[P_][R::dw(m)]), dring[m % DD], dacc, 0, 0, 0); \
            }                                                                                      \
            if (m + DD < DH) SGP_DRD(P_, m + DD);                                                  \
            if (m + 1 == nd[P_]) break;                                                            \
        }                                                                                          \
    } else if ((P_) == 0) {                                                                        \
        dacc = f32x4{0.f, 0.f, 0.f, 0.f};                                                          \
    }                                                                                              \
    stamp(t, (P_) == 0 ? 10 : 11);                                                                 \
    if ((P_) == 1) {                                                                               \
        asm volatile("s_nop 7\n\ts_nop 7\n\tds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:256\n\t" \
                     "ds_write_b32 %0, %3 offset:512\n\tds_write_b32 %0, %4 offset:768"           \
                     :: "v"(slab_w), "v"(dacc.x), "v"(dacc.y), "v"(dacc.z), "v"(dacc.w) : "memory"); \
    }                                                                                              \
    { MID_ }                                                                                       \
    if (n[P_] > 0) {                                                                               \
        _Pragma("unroll") for (int s = 0; s < SHX; ++s) {                                          \
            SGP_WT(P_, s);                                                                         \
            SGP_SLOT(P_, s, (P_) == 0 && s == 0)                                                   \
            if (s + D < SH) { SGP_RD(P_, s + D); }                                                 \
            else if (s < SH) { if (nl[P_] > SH) SGP_RD(P_, s + D); }                               \
            else if (s + D < SHX) { SGP_RD(P_, s + D); }                                           \
            if (s + 1 == nl[P_]) break;                                                            \
        }                                                                                          \
        if (n[P_] > nl[P_]) overflow(qrel[P_], nl[P_] >> 2, n[P_]);                                \
    } else if ((P_) == 0) {                                                                        \
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; acc[2] = acc[0]; acc[3] = acc[0];     \
    }

    // ---- interleaved form of a phase (ILV): ONE loop in which the wave alternates a 16x16x4 instruction
    // of its block with a super-step (4 x 4x4x1) of its group,
    //     [i < nd:  wait N(i) | mfma 16x16x4]  issue N(i + DD)  [i < nl:  wait S(i) | 4 mfma 4x4x1]  issue S(i + D)
    // so that the 32 matrix-pipe cycles of either cover the LDS latency of the other's operands -- a
    // wave whose 4x4x1 super-steps run back to back waits ~100 cycles per super-step for its reads
    // (3 in flight against ~300 cycles of loaded LDS latency), and with two of a SIMD's four waves
    // parked at the barrier nobody fills the gaps.  Only the MFMAs and their waits are conditional; the
    // READS of the first DH iterations are issued whatever nd and nl are (past a list's end they fetch
    // a valid address for nothing), so the issue order is static and every count exact:
    //   issued before iteration i: P0 + a(i) + b(i), P0 = D + DD, a(i) = min(i, DH - DD) dense refills,
    //   b(i) = min(i, SH - D) sparse refills; N(m) sits at D + m (m < DD) or P0 + a(j) + b(j), j = m - DD;
    //   S(s) at s (s < D) or P0 + a(j) + b(j) + [j < DH - DD], j = s - D.
    // Past iteration SH - D the sparse refills depend on the range using its extension quads; the dense
    // reads are over by then (static_assert), and the counts of the plain form apply.
    static_assert(!ILV || (DH + D <= SH && DH <= SH - 2 * D + 1 + DD), "interleaved part must stay inside the resident super-steps");
#define I_A(I_) ((I_) < (DH - DD) ? (I_) : (DH - DD))
#define I_B(I_) ((I_) < (SH - D) ? (I_) : (SH - D))
#define I_POSN(M_) ((M_) < DD ? D + (M_) : D + DD + I_A((M_) - DD) + I_B((M_) - DD))
#define I_POSS(S_) ((S_) < D ? (S_) : D + DD + I_A((S_) - D) + I_B((S_) - D) + (((S_) - D) < (DH - DD) ? 1 : 0))
#define W_IDENSE(I_) (D + DD + I_A(I_) + I_B(I_) - I_POSN(I_) - 1)
#define W_ISPARSE(I_) ((I_) + D - 1 >= SH ? W_TAIL(I_) : (D + DD + I_A(I_) + I_B(I_) + ((I_) < (DH - DD) ? 1 : 0) - I_POSS(I_) - 1))
#define SGP_DENSE_MFMA(P_, M_)                                                                     \
    if ((P_) == 0 && (M_) == 0) {                                                                  \
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};                                                      \
        dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(F[P_][R::dw(M_)]), dring[(M_) % DD], z, 0, 0, 0); \
    } else {                                                                                       \
        dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(F[P_][R::dw(M_)]), dring[(M_) % DD], dacc, 0, 0, 0); \
    }
#define SGP_PHASE_ILV(P_, PRE_, MID_, POST_)                                                       \
    { _Pragma("unroll") for (int s = 0; s < D; ++s) SGP_RD(P_, s); }                               \
    { _Pragma("unroll") for (int m = 0; m < DD; ++m) SGP_DRD(P_, m); }                             \
    { PRE_ }                                                                                       \
    if ((P_) == 0) stamp(t, 9);                                                                    \
    if ((P_) == 0 && nd[P_] == 0) dacc = f32x4{0.f, 0.f, 0.f, 0.f};                                \
    if ((P_) == 0 && n[P_] == 0) { acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; acc[2] = acc[0]; acc[3] = acc[0]; } \
    if (kk[P_] > 0) {                                                                              \
        _Pragma("unroll") for (int i = 0; i < SHX; ++i) {                                          \
            if (i < DH) {                                                                          \
                if (i < nd[P_]) {                                                                  \
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dring[i % DD]) : "n"(W_IDENSE(i < DH ? i : 0))); \
                    SGP_DENSE_MFMA(P_, (i < DH ? i : 0))                                           \
                }                                                                                  \
                if (i + DD < DH) SGP_DRD(P_, (i + DD < DH ? i + DD : 0));                          \
            }                                                                                      \
            if (i >= DH || i < nl[P_]) {                                                           \
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[i % D]) : "n"(W_ISPARSE(i)));     \
                SGP_SLOT(P_, i, (P_) == 0 && i == 0)                                               \
            }                                                                                      \
            if (i + D < SH) { SGP_RD(P_, i + D); }                                                 \
            else if (i < SH) { if (nl[P_] > SH) SGP_RD(P_, i + D); }                               \
            else if (i + D < SHX) { SGP_RD(P_, i + D); }                                           \
            if (i == 2) { MID_ }                                                                   \
            if (i + 1 == kk[P_]) break;                                                            \
        }                                                                                          \
    }                                                                                              \
    stamp(t, (P_) == 0 ? 10 : 11);                                                                 \
    if ((P_) == 1) {                                                                               \
        asm volatile("s_nop 7\n\ts_nop 7\n\tds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:256\n\t" \
                     "ds_write_b32 %0, %3 offset:512\n\tds_write_b32 %0, %4 offset:768"           \
                     :: "v"(slab_w), "v"(dacc.x), "v"(dacc.y), "v"(dacc.z), "v"(dacc.w) : "memory"); \
    }                                                                                              \
    { POST_ }                                                                                      \
    if (n[P_] > nl[P_]) overflow(qrel[P_], nl[P_] >> 2, n[P_]);

    auto stamp = [&](int t, int point) {
        if constexpr ((ABL & 128) != 0) {
            const int ts = t - t_begin - 8;
            if (wg == 777 && ts >= 0 && ts < 4) {
                const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
                if (lane == 0) a.dbg[((ts * NW + wave) * 12 + point)] = now;
            }
        }
    };

    __syncthreads();
    dma_segment(x_step, h_step, piecesA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Who does what when (mode, wave-uniform):
    //  * fold + store of the previous step ("emit", ~30 VALU instructions) -- bit 2: the waves of blocks 1
    //    and 3 run their dense instructions FIRST and emit afterwards, so that on every SIMD two waves
    //    fold while the other two keep the matrix pipe busy (all 16 waves folding right after the barrier
    //    leaves it idle for ~1000 of a step's ~8000 cycles);
    //  * refill of the other region -- the younger half of the waves issues its pieces at the start of a
    //    phase; the older half at the end (bits 0-1 = 0), also at the start (1) or after its dense
    //    instructions (2).
    const int dma_mode = a.mode & 3;
    const bool dma_first = wave >= NW / 2 || dma_mode == 1;
    const bool dma_mid = !dma_first && dma_mode == 2;
    const bool dma_last = !dma_first && !dma_mid;
    const bool emit_early = (a.mode & 4) == 0 || (rb & 1) == 0;
    for (int t = t_begin; t < t_end; ++t) {
        asm volatile("" : "+s"(n[0]), "+s"(n[1]), "+s"(nd[0]), "+s"(nd[1]), "+s"(nl[0]), "+s"(nl[1]), "+s"(kk[0]), "+s"(kk[1]));
        // ---- phase A: region A holds step t once every wave's pieces have landed; the slab holds
        // the dense sums of step t - 1 (written before this barrier, next written after the next one)
        stamp(t, 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(t, 1);
        if (dma_first) dma_segment(x_step, h_step, piecesB);
        stamp(t, 2);
        if constexpr (ILV) {
            // (the older half's refill pieces after iteration 2 of the loop or, if it is shorter, after it)
            bool dma_due = dma_mid;
            SGP_PHASE_ILV(0, if (emit_early && t > t_begin) emit(y_step - y_inc);,
                             if (dma_due) { dma_segment(x_step, h_step, piecesB); dma_due = false; },
                             if (!emit_early && t > t_begin) emit(y_step - y_inc);
                             if (dma_due) dma_segment(x_step, h_step, piecesB);)
        } else {
            SGP_PHASE(0, if (emit_early && t > t_begin) emit(y_step - y_inc);,
                         if (!emit_early && t > t_begin) emit(y_step - y_inc);
                         if (dma_mid) dma_segment(x_step, h_step, piecesB);)
        }
        stamp(t, 3);
        if (dma_last) dma_segment(x_step, h_step, piecesB);
        // ---- phase B
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(t, 4);
        asm volatile("s_barrier" ::: "memory");
        stamp(t, 5);
        if (dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);
        stamp(t, 6);
        if constexpr (ILV) {
            bool dma_due = dma_mid && t + 1 < t_end;
            SGP_PHASE_ILV(1, , if (dma_due) { dma_segment(x_step + x_inc, h_step + h_inc, piecesA); dma_due = false; },
                             if (dma_due) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);)
        } else {
            SGP_PHASE(1, , if (dma_mid && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);)
        }
        stamp(t, 7);
        if (dma_last && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(t, 8);
        x_step += x_inc; h_step += h_inc; y_step += y_inc;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    emit(y_step - y_inc);
#undef SGP_PHASE
#undef SGP_PHASE_ILV
#undef SGP_DENSE_MFMA
#undef W_IDENSE
#undef W_ISPARSE
#undef I_POSN
#undef I_POSS
#undef I_A
#undef I_B
#undef SGP_SLOT
#undef SGP_SLOT4
#undef SGP_MF
#undef SGP_WT
#undef SGP_DWT
#undef W_SPARSE
#undef W_TAIL
#undef W_DENSE
#undef SGP_DRD
#undef SGP_RD
}

#ifdef SGP_ABLATION
unsigned* mix_dbg_buffer() {
    static unsigned* p = nullptr;
    if (!p) { (void)hipMalloc(&p, 4 * 16 * 12 * sizeof(unsigned)); (void)hipMemset(p, 0, 4 * 16 * 12 * sizeof(unsigned)); }
    return p;
}
#endif

int g_mix_mode = -1;
int mix_mode() {
    if (g_mix_mode < 0) g_mix_mode = (int)sgp::tune("mix_mode", 6);
    return g_mix_mode;
}
int mix_chunk_cap() {
    static int v = -1;
    if (v < 0) { v = (int)sgp::tune("spmm_chunk", 32); if (v < 1) v = 32; }
    return v;
}

#ifndef SGP_MIX_SH
#define SGP_MIX_SH 12
#define SGP_MIX_DH 10
#define SGP_MIX_SHH 8
#define SGP_MIX_DHH 7
#define SGP_MIX_D 2
#define SGP_MIX_DD 2
#endif
#ifndef SGP_MIX_DDH
#define SGP_MIX_DDH 3
#endif
constexpr int kSH = SGP_MIX_SH, kDH = SGP_MIX_DH;             // resident sparse super-steps / dense instructions per phase
constexpr int kSHh = SGP_MIX_SHH, kDHh = SGP_MIX_DHH;           // with a halo source

template <bool HALO, int SH, int DH, int D, int DD, bool ILV>
int launch_mix(const MixArgs& a, hipStream_t s) {
    const size_t lds_bytes = 160 * 1024;
    dim3 grid((unsigned)(a.n_tiles * a.n_tchunks), a.feat / 64);
#ifdef SGP_ABLATION
    static int abl = -1;
    if (abl < 0) abl = (int)sgp::tune("abl", 0);
#define SGP_ABL(V)                                                                                 \
    if (abl == V) {                                                                                \
        auto k4 = spmm_mix<HALO, SH, DH, D, DD, ILV, V>;                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL(k4, grid, dim3(NW * 64), lds_bytes, s, a);                              \
        return sgp::check_launch("spmm_mix");                                                      \
    }
    SGP_ABL(1) SGP_ABL(2) SGP_ABL(4) SGP_ABL(3) SGP_ABL(5) SGP_ABL(128) SGP_ABL(129) SGP_ABL(512) SGP_ABL(1024) SGP_ABL(640) SGP_ABL(1152) SGP_ABL(2048) SGP_ABL(4096)
#undef SGP_ABL
#endif
    auto kern = spmm_mix<HALO, SH, DH, D, DD, ILV>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_mix: LDS opt-in: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, s, a);
    return sgp::check_launch("spmm_mix");
}

}  // namespace

#ifdef SGP_ABLATION
extern "C" int sgp_spmm_mix_debug_read(unsigned* host) {
    return (int)hipMemcpy(host, mix_dbg_buffer(), 4 * 16 * 12 * sizeof(unsigned), hipMemcpyDeviceToHost);
}
#endif

extern "C" {

int32_t sgp_spmm_mix_max_union(void) { return PASSES * 64; }
int32_t sgp_spmm_mix_max_dense(int32_t halo) { return halo ? kDHh : kDH; }

int sgp_spmm_mix_f32(const int32_t* uptr, const int32_t* ucol, const int32_t* usplit,
                     const int32_t* gptr, const int32_t* gsup, const int32_t* gidx, const float* gw,
                     const int32_t* rowmap,
                     const int32_t* dptr, const int32_t* didx, const float* dw,
                     int32_t n_tiles, int32_t max_union, int32_t max_dense,
                     const float* X, int64_t xrs, int64_t xbs,
                     const float* Xh, int64_t xhrs, int64_t xhbs, int32_t n_own,
                     float* Y, int64_t yrs, int64_t ybs,
                     int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                     const int32_t* pred, int32_t run_if, sgp_stream_t stream) {
    const sgp::Predicate pr{pred, run_if};
    SGP_REQUIRE(uptr && ucol && usplit && gptr && gsup && gidx && gw && rowmap && dptr && didx && dw && X && Y,
                "sgp_spmm_mix_f32: null pointer");
    SGP_REQUIRE(n_tiles >= 0 && n_rows >= 0 && batch >= 0 && max_union >= 0 && max_dense >= 0,
                "sgp_spmm_mix_f32: bad size");
    {
        const long long own = Xh ? n_own : n_cols, far = Xh ? n_cols - n_own : 0;
        SGP_REQUIRE(n_cols >= 0 && own >= 0 && far >= 0 && own * xrs < (1ll << 30) && far * xhrs < (1ll << 30) &&
                    (long long)n_rows * yrs < (1ll << 30),
                    "sgp_spmm_mix_f32: row offsets exceed 32 bits (use sgp_spmm_csr_f32)");
    }
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    if (feat % 64 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_mix_f32: feat=%d is not a multiple of 64", feat);
    if (max_union > sgp_spmm_mix_max_union() || max_dense > sgp_spmm_mix_max_dense(Xh != nullptr))
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_mix_f32: tile working set (%d staged rows, %d dense instructions) exceeds (%d, %d)",
                         max_union, max_dense, sgp_spmm_mix_max_union(), sgp_spmm_mix_max_dense(Xh != nullptr));
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 && sgp::aligned16(X) &&
                sgp::aligned16(Y) && (!Xh || (xhrs % 4 == 0 && xhbs % 4 == 0 && sgp::aligned16(Xh))),
                "sgp_spmm_mix_f32: strides/pointers must be 16-byte aligned");
    MixArgs a;
    a.uptr = uptr; a.ucol = ucol; a.usplit = usplit; a.gptr = gptr; a.gsup = gsup; a.gidx = gidx; a.gw = gw;
    a.rowmap = rowmap; a.dptr = dptr; a.didx = didx; a.dw = dw;
    a.n_tiles = n_tiles;
    a.src = Src2{X, xrs, xbs, Xh ? Xh : X, xhrs, xhbs, Xh ? n_own : 0x7fffffff, pr.flag, pr.want};
    a.Y = Y; a.yrs = yrs; a.ybs = ybs;
    a.n_rows = n_rows; a.batch = batch; a.feat = feat;
    const int nft = feat / 64;
    long long want = (long long)batch * n_tiles * nft / 4096;
    int tc = (int)(want < 16 ? 16 : (want > mix_chunk_cap() ? mix_chunk_cap() : want));
    if (tc > batch) tc = batch;
    a.t_chunk = tc;
    a.n_tchunks = (batch + tc - 1) / tc;
    hipStream_t s = (hipStream_t)stream;
    a.mode = mix_mode();
    a.dbg = nullptr;
#ifdef SGP_ABLATION
    a.dbg = mix_dbg_buffer();
#endif
    if (Xh) return launch_mix<true, kSHh, kDHh, SGP_MIX_D, SGP_MIX_DDH, false>(a, s);
    // (the interleaved form of a phase -- mode bit 5 -- measured slower than the plain one: 10.7 vs 10.1 ms
    // per 512 steps without staging; kept for the record, built only where its static counts hold)
    constexpr bool kIlvOk = kDH + SGP_MIX_D <= kSH && kDH <= kSH - 2 * SGP_MIX_D + 1 + SGP_MIX_DD;
    if constexpr (kIlvOk) {
        if (mix_mode() & 32) return launch_mix<false, kSH, kDH, SGP_MIX_D, SGP_MIX_DD, true>(a, s);
    }
    return launch_mix<false, kSH, kDH, SGP_MIX_D, SGP_MIX_DD, false>(a, s);
}

}  // extern "C"
