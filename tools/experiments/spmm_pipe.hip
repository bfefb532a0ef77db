// Pipelined matrix-core row-group SpMM (reference call site: lib/sgp_preprocessing.py:202,
// x = adj @ x).  gfx950 / wave64 only.  Same arithmetic as spmm_mfma (spmm.hip): a wave owns 4
// output rows, walks the sorted union of their columns 4 column classes at a time and feeds
// v_mfma_f32_4x4x1_16b_f32 (exact fp32).  What differs is how a time step moves through the CU:
//
//   * the tile's staged source rows never pass through registers: every wave issues
//     global_load_lds_dwordx4 (16 B per lane, 4 staged rows = 1 KiB per wave instruction) and the
//     data lands in LDS behind the compute of the previous phase;
//   * the tile's distinct-column list is cut in two SEGMENTS (A = its even positions, B = the
//     odd ones, so every group finds half of its columns in either), every group's stream is
//     stored A-part first; a time step is two phases
//         barrier, DMA B(t)   -> region B | MFMAs on region A
//         barrier, DMA A(t+1) -> region A | MFMAs on region B, fold, store
//     so one half of the stage is being refilled while the other is consumed and the DMA has a
//     whole phase (~2 us) to land: 2 barriers per step, no ds_write pass, no staging VGPRs;
//   * the operand reads are software-pipelined TWO quads deep inside a wave (a quad's registers
//     are reloaded as soon as its MFMAs have issued), so the LDS latency sits under the wave's
//     own MFMAs instead of being exposed every time the 4 waves of a SIMD fall into step at a
//     barrier, and a wave that runs alone at the tail of a phase still streams;
//   * the padding of a range is skipped at the grain of one super-step (4 columns).
// Measured structure of a step (tools/timeline_pipe.py, tools/abl_pipe.sh, tools/ubench/
// quad_loop.hip; MI355X, T = 256): the quad loop alone sustains 75-80 ns per 16 MFMAs per SIMD
// (the matrix pipe's 56 ns is not reachable next to 6 LDS returns per quad); staging alone needs
// 60 % of the step (L2 misses); the two overlap to 4.3 us per step.
#include "common.h"
#include <stdlib.h>

using sgp::f32x4;
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

struct Src2 {
    const float* x;  long long xrs, xbs;
    const float* xh; long long xhrs, xhbs;
    int n_own;
};

struct PipeArgs {
    const int* uptr; const int* ucol; const int* usplit;
    const int* gptr;                       // [32 * n_tiles + 1]: (A, B) quad ranges per group
    const int* gsup;                       // [32 * n_tiles]: super-steps (columns / 4, rounded up) per range
    const int* gidx; const float* gw; const int* rowmap;
    int n_tiles;
    Src2 src;
    float* Y; long long yrs, ybs;
    int n_rows, batch, feat;
    int t_chunk, n_tchunks;
    unsigned* sync;                        // persistent mode: arrival counters [grid.y][8] (zeroed per launch) or null
    unsigned* dbg;                         // timeline stamps (ablation builds only)
};

constexpr int kPasses = 7;                               // 448 staged rows
constexpr int kStageBytes = kPasses * 64 * 256;
constexpr int kQuadBytes = 256 + 64;
// the offset prefetch runs up to 3 quads past a wave's last quad (the offsets sit last in LDS;
// what is fetched there is never used)
constexpr int kSlackBytes = 3 * 64;
constexpr int kMaxQuads = (160 * 1024 - kStageBytes - kSlackBytes) / kQuadBytes;

// LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at `lds_off`
// (wave-uniform byte address, via M0).  Issued from asm so that hipcc's s_waitcnt bookkeeping does
// not know about it: the compiler would otherwise drain vmcnt(0) in front of every ds_read that
// follows.  Completion is counted by hand (s_waitcnt vmcnt(0) before the phase barrier).
// M0 is written here and nowhere else in this kernel (the compiler has no use for it: no
// LDS-DMA builtin, no movrel, no GWS), so it is neither saved nor restored.
__device__ __forceinline__ void dma16_saddr(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
__device__ __forceinline__ void dma16_vaddr(const void* vaddr, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(vaddr), "s"(lds_off) : "memory");
}

// PERSIST: one workgroup per CU for the whole launch.  The workgroups of an XCD (block ids b with
// the same b % 8 -- a speed assumption only, any placement computes the same result) walk "units"
// together: unit = (32 consecutive tiles, one time chunk), workgroup rank r takes tile 32 s + r.
// Neighbouring tiles stage overlapping source rows, so when the XCD's CUs sit on the same time
// steps a row crosses the fabric once per XCD instead of once per tile that uses it.
template <bool HALO, int ABL = 0, bool PERSIST = false>
__global__ __launch_bounds__(1024) void spmm_pipe(PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int f_base = blockIdx.y * 64;
    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3, n_ranks = gridDim.x >> 3;
    const int n_super = (a.n_tiles + n_ranks - 1) / n_ranks;
    const int n_units = PERSIST ? n_super * a.n_tchunks : 1;
    const int unit0 = PERSIST ? (int)((long long)n_units * xcd / 8) : 0;
    const int unit1 = PERSIST ? (int)((long long)n_units * (xcd + 1) / 8) : 1;
  for (int unit = unit0; unit < unit1; ++unit) {
    int wg, tile, tchunk;
    if constexpr (PERSIST) {
        wg = unit;
        tile = (unit / a.n_tchunks) * n_ranks + rank;
        tchunk = unit % a.n_tchunks;
        __syncthreads();                                  // the previous unit's stream is no longer read
        if (a.sync) {
            // soft rendezvous of the XCD's workgroups at the start of a unit (bounded: a late or
            // not yet resident workgroup only costs locality, never progress)
            if (threadIdx.x == 0) {
                unsigned* c = a.sync + blockIdx.y * 8 + xcd;
                const unsigned want = (unsigned)(unit - unit0 + 1) * (unsigned)n_ranks;
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; spin < 400; ++spin) {
                    if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
        }
        if (tile >= a.n_tiles) continue;
    } else {
        // XCD-aware decode (workgroup ids are dealt round-robin to the 8 XCDs -- a speed assumption
        // only): consecutive ids on one XCD = consecutive tiles of one time chunk, so the workgroups
        // that share an L2 stage overlapping source rows of the same time steps.
        const int nwg = a.n_tiles * a.n_tchunks;
        const int orig = blockIdx.x;
        const int qq = nwg >> 3, rr = nwg & 7, x8 = orig & 7;
        wg = (x8 < rr ? x8 * (qq + 1) : rr * (qq + 1) + (x8 - rr) * qq) + (orig >> 3);
        tile = wg % a.n_tiles;
        tchunk = wg / a.n_tiles;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = tid & 15;
    const int eg = tid >> 4;
    const int q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = a.uptr[tile];
    const int nU = a.uptr[tile + 1] - u0;
    const int uA = a.usplit[tile];                       // rows of segment A (multiple of 4)

    const int t_begin = tchunk * a.t_chunk;
    const int t_end = min(a.batch, t_begin + a.t_chunk);
    if (t_begin >= t_end) continue;

    // per-lane source of every staged row this lane feeds: byte offset from the step base
    unsigned voff[kPasses];
    unsigned halo_mask = 0;
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        const int u = p * 64 + eg;
        const int c = (u < nU) ? a.ucol[u0 + u] : (nU > 0 ? a.ucol[u0] : 0);
        if (HALO && c >= a.src.n_own) {
            halo_mask |= 1u << p;
            voff[p] = (unsigned)((c - a.src.n_own) * (int)a.src.xhrs + f_base + li * 4) * 4u;
        } else {
            voff[p] = (unsigned)(c * (int)a.src.xrs + f_base + li * 4) * 4u;
        }
    }

    // the tile's stream -> LDS (once per workgroup): weights then indices
    const int tile_q0 = a.gptr[tile * 32], tile_q1 = a.gptr[tile * 32 + 32];
    const int tile_quads = tile_q1 - tile_q0;
    char* wlds = lds + kStageBytes;
    char* ilds = wlds + tile_quads * 256;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.gw) + (long long)tile_q0 * 16;
        f32x4* dst = reinterpret_cast<f32x4*>(wlds);
        for (int i = tid; i < tile_quads * 16; i += 1024) dst[i] = src[i];
        const f32x4* isrc = reinterpret_cast<const f32x4*>(a.gidx) + (long long)tile_q0 * 4;
        f32x4* idst = reinterpret_cast<f32x4*>(ilds);
        for (int i = tid; i < tile_quads * 4; i += 1024) idst[i] = isrc[i];
    }

    const int grp = (tile * 16 + wave) * 2;
    const int gA = __builtin_amdgcn_readfirstlane(a.gptr[grp]) - tile_q0;
    const int gB = __builtin_amdgcn_readfirstlane(a.gptr[grp + 1]) - tile_q0;
    const int gE = __builtin_amdgcn_readfirstlane(a.gptr[grp + 2]) - tile_q0;
    const int nA = gB - gA, nB = gE - gB;
    // super-steps of the LAST quad of either range (1..4): the padding of a range is skipped in
    // units of 4 columns instead of 16
    const int lastA = __builtin_amdgcn_readfirstlane(a.gsup[grp]) - 4 * (nA - 1);
    const int lastB = __builtin_amdgcn_readfirstlane(a.gsup[grp + 1]) - 4 * (nB - 1);
    const int my_row = a.rowmap[tile * 64 + wave * 4 + q];   // output row of class q (-1: none)
    // per-lane LDS byte addresses of the wave's stream (quad 0 of either range).  They are kept
    // as opaque integers: a read is "address register + small immediate", the loop advances the
    // registers -- otherwise the compiler rebuilds `lds + 0x1c000 + ...` with a VALU add per read.
    typedef const __attribute__((address_space(3))) f32x4* lds_f4_t;
    typedef const __attribute__((address_space(3))) float* lds_f1_t;
    typedef const __attribute__((address_space(3))) unsigned* lds_u1_t;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // weights: one float per lane and quad -- lane (q, b = li >> 2, i = li & 3) holds the weight of
    // row i for class q's column in super-step b; the MFMA of super-step s takes it from block b = s
    // of the class (cbsz = 2, abid = s).  Offsets: lane (q, s = li & 3) holds class q's staged-row
    // offset of super-step s; the address add takes it from lane s of its quad (DPP quad_perm).
    unsigned wA = lds0 + kStageBytes + gA * 256 + lane * 4;
    unsigned iA = lds0 + kStageBytes + tile_quads * 256 + gA * 64 + q * 16 + (lane & 3) * 4;
    unsigned wB = wA + nA * 256;
    unsigned iB = iA + nA * 64;
    asm volatile("" : "+v"(wA), "+v"(iA), "+v"(wB), "+v"(iB));
    const unsigned xmine = lds0 + li * 16;
    // debug timeline (ABL & 32): workgroup `dbg[0]` records s_memtime at 8 points of 4 steps
    auto stamp = [&](int t, int point) {
        if constexpr (ABL & 32) {
            const int ts = t - t_begin - 8;
            if (wg == 777 && ts >= 0 && ts < 4) {
                const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
                if (lane == 0) a.dbg[((ts * 16 + wave) * 8 + point)] = now;
            }
        }
    };

    // DMA of one segment of step t: wave w moves staged rows 64 p + 4 w .. + 3 (one 1-KiB piece)
    const unsigned lds_base = lds0;
    const char* x_step = reinterpret_cast<const char*>(a.src.x + (long long)t_begin * a.src.xbs);
    const char* h_step = reinterpret_cast<const char*>(a.src.xh + (long long)t_begin * a.src.xhbs);
    const long long x_inc = a.src.xbs * 4, h_inc = a.src.xhbs * 4;
    const char* const x_step0 = x_step;
    const char* const h_step0 = h_step;
    // which of the (up to 7) pieces of this wave belong to segment A / B: one bit per piece, kept
    // in two scalars (14 separate booleans cost 28 SGPRs and push the compiler into spilling
    // SGPRs to VGPR lanes, i.e. v_readlane -- VALU work -- inside the time loop)
    unsigned piecesA = 0, piecesB = 0;
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        const int r0 = p * 64 + wave * 4;
        if (r0 < uA) piecesA |= 1u << p;
        else if (r0 < nU) piecesB |= 1u << p;
    }
    piecesA = __builtin_amdgcn_readfirstlane(piecesA);
    piecesB = __builtin_amdgcn_readfirstlane(piecesB);
    const unsigned piece0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);
    // (xt, ht) = base of the step whose rows are fetched
    auto dma_segment = [&](const char* xt, const char* ht, unsigned pieces) {
        if constexpr (ABL & 1) return;                    // ablation: no staging traffic
        if constexpr (ABL & 4) { xt = x_step0; ht = h_step0; }   // ablation: staging hits L2
#pragma unroll
        for (int p = 0; p < kPasses; ++p) {
            if (pieces & (1u << p)) {                     // scalar
                const unsigned dst = piece0 + (unsigned)p * 16384u;
                if constexpr (HALO) {
                    const char* b = ((halo_mask >> p) & 1u) ? ht : xt;
                    dma16_vaddr(b + voff[p], __builtin_amdgcn_readfirstlane(dst));
                } else {
                    dma16_saddr(voff[p], xt, dst);
                }
            }
        }
    };
    __syncthreads();                                      // stream visible to every wave
    dma_segment(x_step, h_step, piecesA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // Operand pipeline of a wave, two quads deep.  Quad c uses register set c & 1:
    //   Xa/Xb  staged operands of quads c, c+1     Wa/Wb  their weights
    //   Ia/Ib  LDS offsets of quads c+2, c+3 (already fetched)
    // "Long" body of quad c: after the 4 MFMAs of super-step s have been ISSUED (an MFMA reads its
    // sources at issue) the same registers are reloaded with super-step s of quad c+2, so every
    // wave keeps two quads of LDS reads in flight and a wave that runs alone on its SIMD (the
    // tail of a phase) is not bound by the LDS latency.  The last two quads use the "tail" body.
    float Wa, Wb;
    f32x4 Xa[4], Xb[4];
    unsigned Ia, Ib;
#define SGP_LDW(DST, WP, C) DST = *(lds_f1_t)((WP) + (C) * 256)
#define SGP_LDI(DST, IP, C) DST = *(lds_u1_t)((IP) + (C) * 64)
#define SGP_QP(S) ((S) | ((S) << 2) | ((S) << 4) | ((S) << 6))
#define SGP_LD1(DST, I, S) DST = *(lds_f4_t)(xmine + (unsigned)__builtin_amdgcn_update_dpp(0, (int)(I), SGP_QP(S), 0xf, 0xf, true))
#define SGP_LDX(X, I) SGP_LD1(X[0], I, 0); SGP_LD1(X[1], I, 1); SGP_LD1(X[2], I, 2); SGP_LD1(X[3], I, 3);
#define SGP_SUPER(W, XV, S)                                                                     \
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.x, acc0, 2, S, 0);                          \
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.y, acc1, 2, S, 0);                          \
    acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.z, acc2, 2, S, 0);                          \
    acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.w, acc3, 2, S, 0);
#define SGP_SG(MASK, N) __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
    // tail body: 16 MFMAs, nothing to fetch
#define SGP_BODY_T(W, X) SGP_SUPER(W, X[0], 0) SGP_SUPER(W, X[1], 1) SGP_SUPER(W, X[2], 2) SGP_SUPER(W, X[3], 3)
    // long body of quad C: MFMAs of C, operand reloads for C+2, then W(C+2) and I(C+4)
#define SGP_BODY_L(W, X, I, WP, IP, C)  /* C = 0 or 1: quad relative to the running pointers */ \
    SGP_SUPER(W, X[0], 0) SGP_LD1(X[0], I, 0);                                                  \
    SGP_SUPER(W, X[1], 1) SGP_LD1(X[1], I, 1);                                                  \
    SGP_SUPER(W, X[2], 2) SGP_LD1(X[2], I, 2);                                                  \
    SGP_SUPER(W, X[3], 3) SGP_LD1(X[3], I, 3);                                                  \
    SGP_LDW(W, WP, (C) + 2); SGP_LDI(I, IP, (C) + 4);                                           \
    SGP_SG(0x008, 4) SGP_SG(0x100, 1) SGP_SG(0x008, 4) SGP_SG(0x100, 1)                         \
    SGP_SG(0x008, 4) SGP_SG(0x100, 1) SGP_SG(0x008, 4) SGP_SG(0x100, 3)
    // last quad of a range: only its first NS super-steps carry columns
#define SGP_BODY_E(W, X, NS)                                                                    \
    SGP_SUPER(W, X[0], 0)                                                                       \
    if ((NS) > 1) { SGP_SUPER(W, X[1], 1)                                                       \
        if ((NS) > 2) { SGP_SUPER(W, X[2], 2)                                                   \
            if ((NS) > 3) { SGP_SUPER(W, X[3], 3) } } }
    // before the barrier (the stream is static): offsets of quads 0, 1
#define SGP_PRE(WP, IP) SGP_LDI(Ia, IP, 0); SGP_LDI(Ib, IP, 1);
    // one phase: quads 0 .. NQ-1 of the stream at (WP, IP), the last one NS super-steps long
#define SGP_PHASE(WP0, IP0, NQ, NS, MID)                                                        \
    if (!((NQ) > 0 && !(ABL & 2))) { MID } else {                                               \
        unsigned wq = (WP0), iq = (IP0);                                                        \
        /* issued in exactly the order of the steady state (X0-3, W, I per quad): hipcc's     */ \
        /* s_waitcnt pass merges the loop entry with the back edge and takes the stricter     */ \
        /* count -- with a different order here every iteration drained the LDS queue to one  */ \
        /* outstanding read at its top.  (Quad 1's reads are issued even if the range has a   */ \
        /* single quad: what they fetch is never used.)                                       */ \
        SGP_LDX(Xa, Ia) SGP_LDW(Wa, wq, 0); SGP_LDI(Ia, iq, 2);                                 \
        SGP_LDX(Xb, Ib) SGP_LDW(Wb, wq, 1); SGP_LDI(Ib, iq, 3);                                 \
        MID                                                                                     \
        int c = 0;                                                                              \
        for (; c + 3 < (NQ); c += 2) {                                                          \
            /* priority toggles per quad (measured +2 %: the waves of a SIMD fall out of step, */ \
            /* one is in its MFMA burst while another waits for operands); static priorities   */ \
            /* by age and "most work left first" were 3-10 % SLOWER than the default           */ \
            __builtin_amdgcn_s_setprio(2);                                                      \
            SGP_BODY_L(Wa, Xa, Ia, wq, iq, 0)                                                   \
            __builtin_amdgcn_s_setprio(0);                                                      \
            SGP_BODY_L(Wb, Xb, Ib, wq, iq, 1)                                                   \
            wq += 512; iq += 128;                                                               \
            asm volatile("" : "+v"(wq), "+v"(iq));                                              \
        }                                                                                       \
        const int left = (NQ) - c;                                                              \
        if (left == 3) {                                                                        \
            SGP_BODY_L(Wa, Xa, Ia, wq, iq, 0)                                                   \
            SGP_BODY_T(Wb, Xb)                                                                  \
            SGP_BODY_E(Wa, Xa, NS)                                                              \
        } else if (left == 2) {                                                                 \
            SGP_BODY_T(Wa, Xa)                                                                  \
            SGP_BODY_E(Wb, Xb, NS)                                                              \
        } else {                                                                                \
            SGP_BODY_E(Wa, Xa, NS)                                                              \
        }                                                                                       \
    }

    float* y_row = a.Y + (long long)t_begin * a.ybs + (long long)(my_row < 0 ? 0 : my_row) * a.yrs + f_base + li * 4;
    // sum the 4 column classes; class q keeps row q (see spmm.hip)
#define SGP_FOLD(ACC, DST)                                                                       \
    {                                                                                           \
        auto p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ACC.x), __float_as_uint(ACC.y), false, false); \
        auto p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ACC.z), __float_as_uint(ACC.w), false, false); \
        const float r01 = __uint_as_float(p01[0]) + __uint_as_float(p01[1]);                    \
        const float r23 = __uint_as_float(p23[0]) + __uint_as_float(p23[1]);                    \
        auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(r01), __float_as_uint(r23), false, false); \
        DST = __uint_as_float(h[0]) + __uint_as_float(h[1]);                                    \
    }
    // result of the step held in the accumulators -> its row (streamed: nontemporal, so that it
    // does not displace the staged rows other tiles of this XCD are about to re-read from L2)
#define SGP_EMIT                                                                                 \
    {                                                                                           \
        f32x4 out;                                                                              \
        SGP_FOLD(acc0, out.x) SGP_FOLD(acc1, out.y) SGP_FOLD(acc2, out.z) SGP_FOLD(acc3, out.w) \
        if (my_row >= 0) __builtin_nontemporal_store(out, reinterpret_cast<f32x4*>(y_row));     \
        y_row += a.ybs;                                                                         \
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0; acc2 = acc0; acc3 = acc0;                \
    }
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int t = t_begin; t < t_end; ++t) {
        // ---- phase A: region A holds step t once every wave's pieces have landed
        SGP_PRE(wA, iA)
        stamp(t, 0);
        asm volatile("s_barrier" ::: "memory");
        stamp(t, 1);
        // the refill of the other region is issued first by the younger half of the waves (the
        // hardware serves the oldest wave of a SIMD first, they would wait for the matrix pipe
        // anyway) and after their quads by the older half: an LDS-DMA piece costs its wave
        // 150+ cycles of issue, which must not idle the matrix pipe at the start of the phase
        const bool dma_first = wave >= 8;
        if (dma_first) dma_segment(x_step, h_step, piecesB);
        stamp(t, 2);
        // the fold + store of step t-1 sit UNDER the first operand reads of step t (the wave
        // would wait for them anyway) instead of in front of the barrier every wave waits at
        SGP_PHASE(wA, iA, nA, lastA, if (t > t_begin) SGP_EMIT)
        if (!dma_first) dma_segment(x_step, h_step, piecesB);
        // ---- phase B
        SGP_PRE(wB, iB)
        stamp(t, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(t, 4);
        asm volatile("s_barrier" ::: "memory");
        stamp(t, 5);
        if (dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);
        SGP_PHASE(wB, iB, nB, lastB, )
        if (!dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, h_step + h_inc, piecesA);
        stamp(t, 6);
        // this wave's pieces of A(t+1) (and the store of step t-1) retired before the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(t, 7);
        x_step += x_inc; h_step += h_inc;
    }
    SGP_EMIT
  }   // unit
#undef SGP_EMIT
#undef SGP_FOLD
#undef SGP_PRE
#undef SGP_PHASE
#undef SGP_BODY_L
#undef SGP_BODY_T
#undef SGP_BODY_E
#undef SGP_SG
#undef SGP_SUPER
#undef SGP_LDX
#undef SGP_QP
#undef SGP_LD1
#undef SGP_LDI
#undef SGP_LDW
}

int pipe_chunk_cap() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SGP_SPMM_CHUNK"); v = e ? atoi(e) : 32; if (v < 1) v = 32; }
    return v;
}

unsigned pipe_grid(const PipeArgs& a) { return (unsigned)(a.n_tiles * a.n_tchunks); }

int g_persist = -1, g_unit = -1;
int pipe_persist_mode() {                                  // 0 off, 1 persistent, 2 + rendezvous per unit
    if (g_persist < 0) { const char* e = getenv("SGP_SPMM_PERSIST"); g_persist = e ? atoi(e) : 0; }
    return g_persist;
}
int pipe_unit_steps() {
    if (g_unit < 0) { const char* e = getenv("SGP_SPMM_UNIT"); g_unit = e ? atoi(e) : 128; if (g_unit < 1) g_unit = 128; }
    return g_unit;
}
unsigned* pipe_sync_buffer() {
    static unsigned* p = nullptr;
    if (!p) { if (hipMalloc(&p, 64 * 8 * sizeof(unsigned)) != hipSuccess) p = nullptr; }
    return p;
}

template <bool HALO>
int launch_pipe(PipeArgs a, hipStream_t s) {
    const size_t lds_bytes = 160 * 1024;
    if (pipe_persist_mode() > 0) {
        auto kern = spmm_pipe<HALO, 0, true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return sgp::fail((int)e, "spmm_pipe: LDS opt-in: %s", hipGetErrorString(e));
        const int tc = a.batch < pipe_unit_steps() ? a.batch : pipe_unit_steps();
        a.t_chunk = tc;
        a.n_tchunks = (a.batch + tc - 1) / tc;
        a.sync = nullptr;
        const int ny = a.feat / 64;
        if (pipe_persist_mode() > 1 && ny <= 64 && pipe_sync_buffer()) {
            a.sync = pipe_sync_buffer();
            (void)hipMemsetAsync(a.sync, 0, (size_t)ny * 8 * sizeof(unsigned), s);
        }
        hipLaunchKernelGGL(kern, dim3(256, ny), dim3(1024), lds_bytes, s, a);
        return sgp::check_launch("spmm_pipe");
    }
#ifdef SGP_ABLATION
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("SGP_PIPE_ABL"); abl = e ? atoi(e) : 0; }
#define SGP_ABL(V)                                                                                 \
    if (abl == V) {                                                                                \
        auto k4 = spmm_pipe<HALO, V>;                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL(k4, dim3(pipe_grid(a), a.feat / 64), dim3(1024), lds_bytes, s, a); \
        return sgp::check_launch("spmm_pipe");                                                     \
    }
    SGP_ABL(1) SGP_ABL(2) SGP_ABL(3) SGP_ABL(4) SGP_ABL(6) SGP_ABL(32) SGP_ABL(33) SGP_ABL(35)
#undef SGP_ABL
#endif
    auto kern = spmm_pipe<HALO>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_pipe: LDS opt-in: %s", hipGetErrorString(e));
    dim3 grid(pipe_grid(a), a.feat / 64);
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds_bytes, s, a);
    return sgp::check_launch("spmm_pipe");
}

}  // namespace

#ifdef SGP_ABLATION
static unsigned* pipe_dbg_buffer() {
    static unsigned* p = nullptr;
    if (!p) { (void)hipMalloc(&p, 4 * 16 * 8 * sizeof(unsigned)); (void)hipMemset(p, 0, 4 * 16 * 8 * sizeof(unsigned)); }
    return p;
}
extern "C" int sgp_spmm_pipe_debug_read(unsigned* host) {
    return (int)hipMemcpy(host, pipe_dbg_buffer(), 4 * 16 * 8 * sizeof(unsigned), hipMemcpyDeviceToHost);
}
#endif

extern "C" {

int32_t sgp_spmm_pipe_max_union(void) { return kPasses * 64; }
int sgp_spmm_pipe_tune(int32_t persist, int32_t unit_steps) {
    if (persist >= 0) g_persist = persist;
    if (unit_steps > 0) g_unit = unit_steps;
    return 0;
}
int32_t sgp_spmm_pipe_max_quads(void) { return kMaxQuads; }

int sgp_spmm_pipe_f32(const int32_t* uptr, const int32_t* ucol, const int32_t* usplit,
                      const int32_t* gptr, const int32_t* gsup, const int32_t* gidx, const float* gw,
                      const int32_t* rowmap,
                      int32_t n_tiles, int32_t max_union, int32_t max_tile_quads,
                      const float* X, int64_t xrs, int64_t xbs,
                      const float* Xh, int64_t xhrs, int64_t xhbs, int32_t n_own,
                      float* Y, int64_t yrs, int64_t ybs,
                      int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                      sgp_stream_t stream) {
    SGP_REQUIRE(uptr && ucol && usplit && gptr && gsup && gidx && gw && rowmap && X && Y,
                "sgp_spmm_pipe_f32: null pointer");
    SGP_REQUIRE(n_tiles >= 0 && n_rows >= 0 && batch >= 0 && max_union >= 0 && max_tile_quads >= 0,
                "sgp_spmm_pipe_f32: bad size");
    {
        // the DMA addresses a staged row as a 32-bit BYTE offset from the step base
        const long long own = Xh ? n_own : n_cols, far = Xh ? n_cols - n_own : 0;
        SGP_REQUIRE(n_cols >= 0 && own >= 0 && far >= 0 && own * xrs < (1ll << 30) && far * xhrs < (1ll << 30),
                    "sgp_spmm_pipe_f32: row offsets exceed 32 bits (use sgp_spmm_csr_f32)");
    }
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    if (feat % 64 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_pipe_f32: feat=%d is not a multiple of 64", feat);
    if (max_union > kPasses * 64 || max_tile_quads > kMaxQuads)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_pipe_f32: tile working set (%d rows, %d quads) exceeds LDS (%d, %d)",
                         max_union, max_tile_quads, kPasses * 64, kMaxQuads);
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 && sgp::aligned16(X) &&
                sgp::aligned16(Y) && (!Xh || (xhrs % 4 == 0 && xhbs % 4 == 0 && sgp::aligned16(Xh))) &&
                sgp::aligned16(gidx) && sgp::aligned16(gw),
                "sgp_spmm_pipe_f32: strides/pointers must be 16-byte aligned");
    PipeArgs a;
    a.uptr = uptr; a.ucol = ucol; a.usplit = usplit; a.gptr = gptr; a.gsup = gsup; a.gidx = gidx; a.gw = gw;
    a.rowmap = rowmap;
    a.n_tiles = n_tiles;
    a.src = Src2{X, xrs, xbs, Xh ? Xh : X, xhrs, xhbs, Xh ? n_own : 0x7fffffff};
    a.Y = Y; a.yrs = yrs; a.ybs = ybs;
    a.n_rows = n_rows; a.batch = batch; a.feat = feat;
    // Time chunk per workgroup: long enough to amortise the per-tile setup, short enough that
    // neighbouring tiles (which share source rows through L2 / Infinity Cache) stay in step.
    const int nft = feat / 64;
    long long want = (long long)batch * n_tiles * nft / 4096;
    int tc = (int)(want < 16 ? 16 : (want > pipe_chunk_cap() ? pipe_chunk_cap() : want));
    if (tc > batch) tc = batch;
    a.t_chunk = tc;
    a.n_tchunks = (batch + tc - 1) / tc;
    a.sync = nullptr;
    a.dbg = nullptr;
#ifdef SGP_ABLATION
    a.dbg = pipe_dbg_buffer();
#endif
    hipStream_t s = (hipStream_t)stream;
    return Xh ? launch_pipe<true>(a, s) : launch_pipe<false>(a, s);
}

}  // extern "C"
