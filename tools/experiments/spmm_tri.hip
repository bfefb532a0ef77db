// Three-slot form of the register-resident two-phase row-group SpMM (reference call site:
// lib/sgp_preprocessing.py:202, x = adj @ x).  gfx950 / wave64 only.  Arithmetic, groups, column
// classes and the register-resident stream are those of spmm_res (bit-identical results for the
// same plan); what differs is how far ahead the staging runs.  With two LDS regions a segment can
// only be refilled while the other one is consumed: every LDS-DMA piece has ONE phase (~1.5 us) to
// land, and a phase ends when its slowest piece has -- forcing 50 / 75 % of the staging reads to
// hit the L2 buys 2 / 5 % (ablation bits 4 / 5 of spmm_res), forcing all of them 12 %: it is the
// miss latency of the slowest piece, not the miss count, that is exposed.  Here the stage has
// THREE slots of 192 rows: phase g computes on slot g mod 3 while the segment of phase g + 2 is
// requested into slot (g + 2) mod 3 and the segment of phase g + 1 (requested one phase ago)
// finishes landing, so every piece has two phases.  Slots are interleaved at DMA-piece (1 KiB)
// granularity -- staged row j of a segment lives at 3072 (j / 4) + 1024 slot + 256 (j % 4) -- so
// the slot enters an operand read as the IMMEDIATE offset of ds_read_b128 and the per-lane
// addresses stay put in their registers; the time loop is unrolled by 3 steps (the slot pattern
// repeats every 6 phases).  Price: 384 instead of 448 staged rows per tile (plan built for it,
// graph.ShiftOperator.tri_plan) and no room for an LDS copy of the stream -- ranges beyond the
// resident 20 super-steps read the plan arrays from global memory.
#include "common.h"
#include <stdlib.h>

using sgp::f32x4;

namespace {

struct Src2 {
    const float* x;  long long xrs, xbs;
    const float* xh; long long xhrs, xhbs;
    int n_own;
};

struct TriArgs {
    const int* uptr; const int* ucol; const int* usplit;
    const int* gptr;                       // [2 * GT * n_tiles + 1]: (A, B) quad ranges per group
    const int* gsup;                       // [2 * GT * n_tiles]: super-steps per range
    const int* gidx; const float* gw; const int* rowmap;
    int n_tiles;
    Src2 src;
    float* Y; long long yrs, ybs;
    int n_rows, batch, feat;
    int t_chunk, n_tchunks;
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

// NW waves per workgroup, G row groups per wave (tile = 4 * NW * G rows), operand ring D super-steps
// deep per group parity, SEGP x (4 NW) staged rows per segment and slot.  ABL: bit0 no staging DMA,
// bit2 staging always reads the chunk's first step (ablation builds).
template <bool HALO, int NW, int G, int D, int SEGP, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void spmm_tri(TriArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int GT = NW * G;
    constexpr int RPP = NW * 4;                           // staged rows per DMA pass

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

    // ---- DMA bookkeeping: per-lane source offsets of the staged rows this lane feeds.  Pass p of
    // segment s covers the segment's rows p RPP + 4 wave .. + 3 (piece p NW + wave).
    unsigned voff[2][SEGP];
    unsigned halo_mask = 0;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        const int base = sg ? uA : 0, rows = sg ? nU - uA : uA;
#pragma unroll
        for (int p = 0; p < SEGP; ++p) {
            const int u = p * RPP + eg;
            const int c = (u < rows) ? a.ucol[u0 + base + u] : (nU > 0 ? a.ucol[u0] : 0);
            if (HALO && c >= a.src.n_own) {
                halo_mask |= 1u << (sg * SEGP + p);
                voff[sg][p] = (unsigned)((c - a.src.n_own) * (int)a.src.xhrs + f_base + li * 4) * 4u;
            } else {
                voff[sg][p] = (unsigned)(c * (int)a.src.xrs + f_base + li * 4) * 4u;
            }
        }
    }
    unsigned piecesA = 0, piecesB = 0;
    int nA = 0, nB = 0;                                   // DMA instructions of this wave per segment
#pragma unroll
    for (int p = 0; p < SEGP; ++p) {
        const int r0 = p * RPP + wave * 4;
        if (r0 < uA) { piecesA |= 1u << p; ++nA; }
        if (r0 < nU - uA) { piecesB |= 1u << p; ++nB; }
    }
    piecesA = __builtin_amdgcn_readfirstlane(piecesA);
    piecesB = __builtin_amdgcn_readfirstlane(piecesB);
    nA = __builtin_amdgcn_readfirstlane(nA);
    nB = __builtin_amdgcn_readfirstlane(nB);
    // piece (p, wave) of a segment in slot k: lds0 + 3072 (p NW + wave) + 1024 k
    const unsigned piece0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 3072u);
    const char* x_step = reinterpret_cast<const char*>(a.src.x + (long long)t_begin * a.src.xbs);
    const char* h_step = reinterpret_cast<const char*>(a.src.xh + (long long)t_begin * a.src.xhbs);
    const long long x_inc = a.src.xbs * 4, h_inc = a.src.xhbs * 4;
    const char* const x_step0 = x_step;
    const char* const h_step0 = h_step;
    auto dma_segment = [&](const char* xt, const char* ht, int sg, unsigned pieces, unsigned slot) {
        if constexpr (ABL & 1) return;
        if constexpr (ABL & 4) { xt = x_step0; ht = h_step0; }
#pragma unroll
        for (int p = 0; p < SEGP; ++p) {
            if (pieces & (1u << p)) {                     // scalar
                const unsigned dst = piece0 + (unsigned)p * (unsigned)(NW * 3072) + slot * 1024u;
                const unsigned vo = sg ? voff[1][p] : voff[0][p];
                if constexpr (HALO) {
                    const char* b = ((halo_mask >> (sg * SEGP + p)) & 1u) ? ht : xt;
                    dma16_vaddr(b + vo, __builtin_amdgcn_readfirstlane(dst));
                } else {
                    dma16_saddr(vo, xt, dst);
                }
            }
        }
    };
    // wait until at most `younger` of this wave's memory operations are outstanding: the pieces
    // requested before them have landed (s_waitcnt takes an immediate)
    auto wait_pieces = [&](int younger) {
        if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    };
    static_assert(SEGP <= 3, "wait_pieces covers up to 3 pieces per segment and wave");

    // ---- the wave's stream -> registers (once per workgroup)
    // weights: lane (q, b = li >> 2, i = li & 3) holds row i's weight for class q's column in
    // super-step 4 p + b; addresses: lane (q, li) holds the slot-0 LDS byte address of chunk li of
    // class q's staged row in super-step s.  Padding reads staged row 0 with weight 0.
    unsigned addr[2][G][SH];
    float w[2][G][WH];
    int n[2][G];
    int q0s[2][G];                                         // first quad of the range (global)
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
            q0s[ph][g] = qb;
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

    // super-steps SH .. n-1 of a long range, from the plan arrays in global memory (whole quads:
    // the padding of the last one has weight 0 / staged row 0)
    typedef const __attribute__((address_space(3))) f32x4* lds_f4_t;
    auto overflow = [&](int g, int qfirst, int nsteps, unsigned slot_off) {
        for (int c = WH; c < ((nsteps + 3) >> 2); ++c) {
            const float wv = a.gw[(long long)(qfirst + c) * 64 + lane];
            f32x4 xs[4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                xs[b] = *(lds_f4_t)(lds0 + (unsigned)a.gidx[(long long)(qfirst + c) * 16 + q * 4 + b] + slot_off + li * 16);
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

    // K_ = slot of the segment (compile-time): it enters the read as the immediate offset
#define SGP_RD(P_, G_, S_, K_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[(G_) & 1][(S_) % D]) : "v"(addr[P_][G_][S_]), "n"((K_) * 1024))
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
#define SGP_PHASE(P_, K_, MID_)                                                                    \
    { _Pragma("unroll") for (int s = 0; s < D; ++s) SGP_RD(P_, 0, s, K_); }                        \
    _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                \
        if (g + 1 < G) { _Pragma("unroll") for (int s = 0; s < D; ++s) SGP_RD(P_, g + 1, s, K_); } \
        if (g == 0) { MID_ }                                                                       \
        if ((P_) == 1 && g > 0) emit(g - 1, y_step);                                               \
        if (n[P_][g] > 0) {                                                                        \
            _Pragma("unroll") for (int s = 0; s < SH; ++s) {                                       \
                SGP_WT(g, s);                                                                      \
                SGP_SLOT(P_, g, s, (P_) == 0 && s == 0)                                            \
                if (s + D < SH) SGP_RD(P_, g, s + D, K_);                                          \
                if (s + 1 == n[P_][g]) break;                                                      \
            }                                                                                      \
            if (n[P_][g] > SH) overflow(g, q0s[P_][g], n[P_][g], (K_) * 1024u);                    \
        } else if ((P_) == 0) {                                                                    \
            acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = acc[g][0]; acc[g][2] = acc[g][0]; acc[g][3] = acc[g][0]; \
        }                                                                                          \
    }
    // One step with compile-time slots: segment A of the step sits in slot KA_, B in KB_; the
    // segments of the NEXT step are requested into the two slots that become free, two phases
    // before they are consumed.
#define SGP_STEP(KA_, KB_, KNA_, KNB_)                                                             \
    {                                                                                              \
        const bool more = t + 1 < t_end;                                                           \
        _Pragma("unroll") for (int g = 0; g < G; ++g) asm volatile("" : "+s"(n[0][g]), "+s"(n[1][g])); \
        asm volatile("s_barrier" ::: "memory");            /* A(t) landed, slot KNA_ consumed */   \
        if (dma_first && more) dma_segment(x_step + x_inc, h_step + h_inc, 0, piecesA, KNA_);      \
        SGP_PHASE(0, KA_, if (t > t_begin) emit(G - 1, y_step - y_inc);)                           \
        if (!dma_first && more) dma_segment(x_step + x_inc, h_step + h_inc, 0, piecesA, KNA_);     \
        wait_pieces(more ? nA : 0);                         /* this wave's pieces of B(t) */        \
        asm volatile("s_barrier" ::: "memory");                                                    \
        if (dma_first && more) dma_segment(x_step + x_inc, h_step + h_inc, 1, piecesB, KNB_);      \
        SGP_PHASE(1, KB_, )                                                                        \
        if (!dma_first && more) dma_segment(x_step + x_inc, h_step + h_inc, 1, piecesB, KNB_);     \
        wait_pieces(more ? nB : 0);                         /* this wave's pieces of A(t+1) */      \
        x_step += x_inc; h_step += h_inc; y_step += y_inc;                                         \
        ++t;                                                                                       \
    }

    __syncthreads();
    dma_segment(x_step, h_step, 0, piecesA, 0);
    dma_segment(x_step, h_step, 1, piecesB, 1);
    wait_pieces(nB);                                       // A(t_begin) landed, B may still fly
    const bool dma_first = wave >= NW / 2;
    int t = t_begin;
    while (t < t_end) {
        // the piece masks are re-made opaque every round (hoisted branch masks would be spilled)
        asm volatile("" : "+s"(piecesA), "+s"(piecesB), "+s"(nA), "+s"(nB));
        SGP_STEP(0, 1, 2, 0)                               // phases 6k, 6k+1: slots 0, 1; next -> 2, 0
        if (t >= t_end) break;
        SGP_STEP(2, 0, 1, 2)                               // phases 6k+2, 6k+3
        if (t >= t_end) break;
        SGP_STEP(1, 2, 0, 1)                               // phases 6k+4, 6k+5
    }
    emit(G - 1, y_step - y_inc);
#undef SGP_STEP
#undef SGP_PHASE
#undef SGP_SLOT
#undef SGP_SLOT4
#undef SGP_MF
#undef SGP_WT
#undef SGP_WAITN
#undef SGP_RD
}

int g_tri_cfg = -1;
int tri_cfg() {                                            // 0: 16 waves x 1 group, 1: 8 waves x 2 groups
    if (g_tri_cfg < 0) { const char* e = getenv("SGP_SPMM_TRI_CFG"); g_tri_cfg = e ? atoi(e) : 0; }
    return g_tri_cfg;
}
int tri_chunk_cap() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SGP_SPMM_CHUNK"); v = e ? atoi(e) : 32; if (v < 1) v = 32; }
    return v;
}

template <bool HALO, int NW, int G, int D, int SEGP>
int launch_tri(const TriArgs& a, hipStream_t s) {
    const size_t lds_bytes = 3 * SEGP * NW * 4 * 256;    // three slots
    dim3 grid((unsigned)(a.n_tiles * a.n_tchunks), a.feat / 64);
#ifdef SGP_ABLATION
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("SGP_PIPE_ABL"); abl = e ? atoi(e) : 0; }
#define SGP_ABL(V)                                                                                 \
    if (abl == V) {                                                                                \
        auto k4 = spmm_tri<HALO, NW, G, D, SEGP, V>;                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL(k4, grid, dim3(NW * 64), lds_bytes, s, a);                              \
        return sgp::check_launch("spmm_tri");                                                      \
    }
    SGP_ABL(1) SGP_ABL(4)
#undef SGP_ABL
#endif
    auto kern = spmm_tri<HALO, NW, G, D, SEGP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_res: LDS opt-in: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, s, a);
    return sgp::check_launch("spmm_tri");
}

}  // namespace

extern "C" {

// two segments of at most 3 x 64 = 192 rows; the parity cut pads segment A to 4 rows
int32_t sgp_spmm_tri_max_union(void) { return 2 * 192 - 4; }
int32_t sgp_spmm_tri_max_quads(void) { return 1 << 20; }   // the stream is never copied to LDS
int sgp_spmm_tri_tune(int32_t cfg) { if (cfg >= 0) g_tri_cfg = cfg; return 0; }

int sgp_spmm_tri_f32(const int32_t* uptr, const int32_t* ucol, const int32_t* usplit,
                     const int32_t* gptr, const int32_t* gsup, const int32_t* gidx, const float* gw,
                     const int32_t* rowmap,
                     int32_t n_tiles, int32_t max_union, int32_t max_tile_quads,
                     const float* X, int64_t xrs, int64_t xbs,
                     const float* Xh, int64_t xhrs, int64_t xhbs, int32_t n_own,
                     float* Y, int64_t yrs, int64_t ybs,
                     int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                     sgp_stream_t stream) {
    SGP_REQUIRE(uptr && ucol && usplit && gptr && gsup && gidx && gw && rowmap && X && Y,
                "sgp_spmm_tri_f32: null pointer");
    SGP_REQUIRE(n_tiles >= 0 && n_rows >= 0 && batch >= 0 && max_union >= 0 && max_tile_quads >= 0,
                "sgp_spmm_tri_f32: bad size");
    {
        const long long own = Xh ? n_own : n_cols, far = Xh ? n_cols - n_own : 0;
        SGP_REQUIRE(n_cols >= 0 && own >= 0 && far >= 0 && own * xrs < (1ll << 30) && far * xhrs < (1ll << 30) &&
                    (long long)n_rows * yrs < (1ll << 30),
                    "sgp_spmm_tri_f32: row offsets exceed 32 bits (use sgp_spmm_csr_f32)");
    }
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    if (feat % 64 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_tri_f32: feat=%d is not a multiple of 64", feat);
    // (max_union counts the padding of segment A: 380 distinct rows stage as at most 192 + 190)
    if (max_union > 2 * 192 || max_tile_quads > sgp_spmm_tri_max_quads())
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_tri_f32: tile working set (%d rows, %d quads) exceeds LDS (%d, %d)",
                         max_union, max_tile_quads, sgp_spmm_tri_max_union(), sgp_spmm_tri_max_quads());
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 && sgp::aligned16(X) &&
                sgp::aligned16(Y) && (!Xh || (xhrs % 4 == 0 && xhbs % 4 == 0 && sgp::aligned16(Xh))),
                "sgp_spmm_tri_f32: strides/pointers must be 16-byte aligned");
    TriArgs a;
    a.uptr = uptr; a.ucol = ucol; a.usplit = usplit; a.gptr = gptr; a.gsup = gsup; a.gidx = gidx; a.gw = gw;
    a.rowmap = rowmap;
    a.n_tiles = n_tiles;
    a.src = Src2{X, xrs, xbs, Xh ? Xh : X, xhrs, xhbs, Xh ? n_own : 0x7fffffff};
    a.Y = Y; a.yrs = yrs; a.ybs = ybs;
    a.n_rows = n_rows; a.batch = batch; a.feat = feat;
    const int nft = feat / 64;
    long long want = (long long)batch * n_tiles * nft / 4096;
    int tc = (int)(want < 16 ? 16 : (want > tri_chunk_cap() ? tri_chunk_cap() : want));
    if (tc > batch) tc = batch;
    a.t_chunk = tc;
    a.n_tchunks = (batch + tc - 1) / tc;
    hipStream_t s = (hipStream_t)stream;
    return Xh ? launch_tri<true, 16, 1, 4, 3>(a, s) : launch_tri<false, 16, 1, 4, 3>(a, s);
}

}  // extern "C"
