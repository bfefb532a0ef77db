// Two-team pipelined row-group SpMM (reference call site: lib/sgp_preprocessing.py:202,
// x = adj @ x).  gfx950 / wave64 only.  Same tiles, column segments A | B, LDS-DMA staging and
// v_mfma_f32_4x4x1_16b_f32 arithmetic (exact fp32) as spmm_pipe.hip; what differs is who waits
// for whom.  In spmm_pipe all 16 waves of a workgroup meet at two barriers per time step, and a
// phase between two barriers is short (the LDS holds one step of the tile's source rows: 4-5
// quads per wave), so the ramp after a barrier and the tail before the next one -- when one wave
// per SIMD is left -- cost as much as a third of the step.  Here the 16 waves are two TEAMS of 8
// (team = bit 2 of the wave id, so either team has two waves on every SIMD).  A team owns one 32-feature half
// of the 64-feature slice: its own stage (rows of 128 B: 56 KB, regions A | B), its own DMA pieces
// (8 staged rows = 1 KiB), its own barrier (an LDS counter its 8 waves poll).  The teams share the
// tile's stream (weights / offsets) and nothing else, so they drift apart, and while one team
// sits at its barrier or ramps up the other one keeps the matrix pipes of all four SIMDs busy.
// A wave owns a PAIR of row groups (8 output rows) and walks both at once: of the 16 lanes of a
// class, li < 8 work on the first group, li >= 8 on the second (8 lanes x 16 B = the 128-byte
// row of the team's feature half).  Block b of the MFMA = lanes 4 b .. 4 b + 3; the two blocks of
// a (class, group) share their A values through cbsz:1, so one weight VGPR carries two
// super-steps (abid selects) and a quad of 4 super-steps needs one ds_read_b64 of weights.
#include "common.h"
#include <stdlib.h>

using sgp::f32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

struct DuoArgs {
    const int* uptr; const int* ucol; const int* usplit;
    const int* gptr;                       // [16 * n_tiles + 1]: (A, B) wave-quad ranges per pair slot
    const int* gsup;                       // [16 * n_tiles]: super-steps per range
    const unsigned short* gidx; const float* gw; const int* rowmap;
    int n_tiles;
    const float* x; long long xrs, xbs;
    float* Y; long long yrs, ybs;
    int n_rows, batch, feat;
    int t_chunk, n_tchunks;
};

constexpr int kPasses = 7;                               // 448 staged rows
constexpr int kTeamStage = kPasses * 64 * 128;           // 56 KB per team
constexpr int kStageBytes = 2 * kTeamStage;
constexpr int kQuadBytes = 512 + 64;                     // weights [64 lanes][2], offsets [4][2][4] u16
constexpr int kSlackBytes = 3 * 64;                      // offset prefetch past the last wave-quad
constexpr int kSyncBytes = 16;                           // barrier counters of the two teams
constexpr int kMaxQuads = (160 * 1024 - kStageBytes - kSlackBytes - kSyncBytes) / kQuadBytes;

// LDS-DMA piece: 64 lanes x 16 B (8 staged rows of 128 B) to 1 KiB of LDS at `lds_off` (M0).
// Issued from asm and counted by hand, see spmm_pipe.hip.
__device__ __forceinline__ void dma16(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}

// a wave-uniform pointer the compiler may have placed in VGPRs -> SGPR pair
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// Barrier of one team: every wave adds 1 (LDS executes a wave's operations in order, so the add
// lands behind its reads of the stage) and polls until all 8 arrivals of barrier number `k` are
// in.  A lost arrival must not hang the GPU.
__device__ __forceinline__ void team_barrier(unsigned addr, unsigned target, int lane) {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(1u) : "memory");
    int spins = 0;
    for (;;) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 20)) __builtin_trap();
    }
}

__global__ __launch_bounds__(1024) void spmm_duo(DuoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];

    // XCD-aware decode, see spmm_pipe.hip
    const int nwg = a.n_tiles * a.n_tchunks;
    const int orig = blockIdx.x;
    const int qq = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
    const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    const int tile = wg % a.n_tiles;
    const int tchunk = wg / a.n_tiles;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = tid & 15;
    const int q = lane >> 4;
    const int grp = (li >> 3) & 1;                        // group of the pair this lane works on
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves are dealt to the SIMDs round-robin (a speed assumption only): SIMD s hosts waves s,
    // s + 4, s + 8, s + 12 = teams 0, 1, 0, 1, so either team has two waves on every SIMD
    const int team = (wave >> 2) & 1;
    const int tw = (wave & 3) + 4 * (wave >> 3);          // pair slot = wave of the team
    const int f_team = blockIdx.y * 64 + team * 32;
    const int u0 = a.uptr[tile];
    const int nU = a.uptr[tile + 1] - u0;
    const int uA = a.usplit[tile];                        // rows of segment A (multiple of 8)

    const int t_begin = tchunk * a.t_chunk;
    const int t_end = min(a.batch, t_begin + a.t_chunk);
    if (t_begin >= t_end) return;

    // DMA pieces of this wave: piece p = staged rows 64 p + 8 tw .. + 7 of its team's stage
    unsigned voff[kPasses];
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        const int u = p * 64 + tw * 8 + (lane >> 3);
        const int c = (u < nU) ? a.ucol[u0 + u] : (nU > 0 ? a.ucol[u0] : 0);
        voff[p] = (unsigned)(c * (int)a.xrs + f_team + (lane & 7) * 4) * 4u;
    }
    unsigned piecesA = 0, piecesB = 0;
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        const int r0 = p * 64 + tw * 8;
        if (r0 < uA) piecesA |= 1u << p;
        else if (r0 < nU) piecesB |= 1u << p;
    }
    piecesA = __builtin_amdgcn_readfirstlane(piecesA);
    piecesB = __builtin_amdgcn_readfirstlane(piecesB);

    // the tile's stream -> LDS (once per workgroup, shared by the teams): weights then offsets
    const int tile_q0 = a.gptr[tile * 16], tile_q1 = a.gptr[tile * 16 + 16];
    const int tile_quads = tile_q1 - tile_q0;
    char* wlds = lds + kStageBytes;
    char* ilds = wlds + tile_quads * 512;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.gw) + (long long)tile_q0 * 32;
        f32x4* dst = reinterpret_cast<f32x4*>(wlds);
        for (int i = tid; i < tile_quads * 32; i += 1024) dst[i] = src[i];
        const f32x4* isrc = reinterpret_cast<const f32x4*>(a.gidx) + (long long)tile_q0 * 4;
        f32x4* idst = reinterpret_cast<f32x4*>(ilds);
        for (int i = tid; i < tile_quads * 4; i += 1024) idst[i] = isrc[i];
        if (tid < kSyncBytes / 4) *reinterpret_cast<unsigned*>(lds + 160 * 1024 - kSyncBytes + tid * 4) = 0u;
    }

    const int slot = (tile * 8 + tw) * 2;
    const int gA = __builtin_amdgcn_readfirstlane(a.gptr[slot]) - tile_q0;
    const int gB = __builtin_amdgcn_readfirstlane(a.gptr[slot + 1]) - tile_q0;
    const int gE = __builtin_amdgcn_readfirstlane(a.gptr[slot + 2]) - tile_q0;
    const int nA = gB - gA, nB = gE - gB;
    const int lastA = __builtin_amdgcn_readfirstlane(a.gsup[slot]) - 4 * (nA - 1);
    const int lastB = __builtin_amdgcn_readfirstlane(a.gsup[slot + 1]) - 4 * (nB - 1);
    const int my_row = a.rowmap[tile * 64 + tw * 8 + grp * 4 + q];   // output row of class q (-1: none)

    typedef const __attribute__((address_space(3))) f32x4* lds_f4_t;
    typedef const __attribute__((address_space(3))) f32x2* lds_f2_t;
    typedef const __attribute__((address_space(3))) unsigned short* lds_u16_t;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned stage0 = lds0 + (unsigned)team * kTeamStage;
    // per-lane LDS addresses of the wave's stream (opaque integers, see spmm_pipe.hip)
    unsigned wA = lds0 + kStageBytes + gA * 512 + lane * 8;
    unsigned iA = lds0 + kStageBytes + tile_quads * 512 + gA * 64 + (q * 8 + grp * 4 + (lane & 3)) * 2;
    unsigned wB = wA + nA * 512;
    unsigned iB = iA + nA * 64;
    asm volatile("" : "+v"(wA), "+v"(iA), "+v"(wB), "+v"(iB));
    const unsigned xmine = stage0 + (li & 7) * 16;
    const unsigned bar = lds0 + 160 * 1024 - kSyncBytes + (unsigned)team * 4u;
    unsigned bar_target = 0;

    const char* x_step = reinterpret_cast<const char*>(a.x + (long long)t_begin * a.xbs);
    const long long x_inc = a.xbs * 4;
    const unsigned piece0 = __builtin_amdgcn_readfirstlane(stage0 + (unsigned)tw * 1024u);
    auto dma_segment = [&](const char* xt, unsigned pieces) {
        xt = uniform_ptr(xt);
#pragma unroll
        for (int p = 0; p < kPasses; ++p)
            if (pieces & (1u << p))                       // scalar
                dma16(voff[p], xt, piece0 + (unsigned)p * 8192u);
    };
    __syncthreads();                                      // stream and counters visible to every wave
    dma_segment(x_step, piecesA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // operand pipeline two wave-quads deep, see spmm_pipe.hip
    f32x2 Wa, Wb;
    f32x4 Xa[4], Xb[4];
    unsigned Ia, Ib;
#define SGP_LDW(DST, WP, C) DST = *(lds_f2_t)((WP) + (C) * 512)
#define SGP_LDI(DST, IP, C) DST = (unsigned)*(lds_u16_t)((IP) + (C) * 64)
#define SGP_QP(S) ((S) | ((S) << 2) | ((S) << 4) | ((S) << 6))
#define SGP_LD1(DST, I, S) DST = *(lds_f4_t)(xmine + (unsigned)__builtin_amdgcn_update_dpp(0, (int)(I), SGP_QP(S), 0xf, 0xf, true))
#define SGP_LDX(X, I) SGP_LD1(X[0], I, 0); SGP_LD1(X[1], I, 1); SGP_LD1(X[2], I, 2); SGP_LD1(X[3], I, 3);
#define SGP_SUPER(WH, XV, AB)                                                                   \
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(WH, XV.x, acc0, 1, AB, 0);                        \
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(WH, XV.y, acc1, 1, AB, 0);                        \
    acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(WH, XV.z, acc2, 1, AB, 0);                        \
    acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(WH, XV.w, acc3, 1, AB, 0);
#define SGP_SG(MASK, N) __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
#define SGP_BODY_T(W, X) SGP_SUPER(W.x, X[0], 0) SGP_SUPER(W.x, X[1], 1) SGP_SUPER(W.y, X[2], 0) SGP_SUPER(W.y, X[3], 1)
#define SGP_BODY_L(W, X, I, WP, IP, C)                                                          \
    SGP_SUPER(W.x, X[0], 0) SGP_LD1(X[0], I, 0);                                                \
    SGP_SUPER(W.x, X[1], 1) SGP_LD1(X[1], I, 1);                                                \
    SGP_SUPER(W.y, X[2], 0) SGP_LD1(X[2], I, 2);                                                \
    SGP_SUPER(W.y, X[3], 1) SGP_LD1(X[3], I, 3);                                                \
    SGP_LDW(W, WP, (C) + 2); SGP_LDI(I, IP, (C) + 4);                                           \
    SGP_SG(0x008, 4) SGP_SG(0x100, 1) SGP_SG(0x008, 4) SGP_SG(0x100, 1)                         \
    SGP_SG(0x008, 4) SGP_SG(0x100, 1) SGP_SG(0x008, 4) SGP_SG(0x100, 3)
#define SGP_BODY_E(W, X, NS)                                                                    \
    SGP_SUPER(W.x, X[0], 0)                                                                     \
    if ((NS) > 1) { SGP_SUPER(W.x, X[1], 1)                                                     \
        if ((NS) > 2) { SGP_SUPER(W.y, X[2], 0)                                                 \
            if ((NS) > 3) { SGP_SUPER(W.y, X[3], 1) } } }
#define SGP_PRE(WP, IP) SGP_LDI(Ia, IP, 0); SGP_LDI(Ib, IP, 1);
#define SGP_PHASE(WP0, IP0, NQ, NS, MID)                                                        \
    if (!((NQ) > 0)) { MID } else {                                                             \
        unsigned wq = (WP0), iq = (IP0);                                                        \
        SGP_LDX(Xa, Ia) SGP_LDW(Wa, wq, 0); SGP_LDI(Ia, iq, 2);                                 \
        SGP_LDX(Xb, Ib) SGP_LDW(Wb, wq, 1); SGP_LDI(Ib, iq, 3);                                 \
        MID                                                                                     \
        int c = 0;                                                                              \
        for (; c + 3 < (NQ); c += 2) {                                                          \
            __builtin_amdgcn_s_setprio(2);                                                      \
            SGP_BODY_L(Wa, Xa, Ia, wq, iq, 0)                                                   \
            __builtin_amdgcn_s_setprio(0);                                                      \
            SGP_BODY_L(Wb, Xb, Ib, wq, iq, 1)                                                   \
            wq += 1024; iq += 128;                                                              \
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
    // sum the 4 column classes; class q keeps row q of the lane's group (see spmm.hip)
#define SGP_FOLD(ACC, DST)                                                                       \
    {                                                                                           \
        auto p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ACC.x), __float_as_uint(ACC.y), false, false); \
        auto p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ACC.z), __float_as_uint(ACC.w), false, false); \
        const float r01 = __uint_as_float(p01[0]) + __uint_as_float(p01[1]);                    \
        const float r23 = __uint_as_float(p23[0]) + __uint_as_float(p23[1]);                    \
        auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(r01), __float_as_uint(r23), false, false); \
        DST = __uint_as_float(h[0]) + __uint_as_float(h[1]);                                    \
    }
#define SGP_EMIT                                                                                 \
    {                                                                                           \
        f32x4 out;                                                                              \
        SGP_FOLD(acc0, out.x) SGP_FOLD(acc1, out.y) SGP_FOLD(acc2, out.z) SGP_FOLD(acc3, out.w) \
        if (my_row >= 0) __builtin_nontemporal_store(out, reinterpret_cast<f32x4*>(y_row));     \
        y_row += a.ybs;                                                                         \
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0; acc2 = acc0; acc3 = acc0;                \
    }

    float* y_row = a.Y + (long long)t_begin * a.ybs + (long long)(my_row < 0 ? 0 : my_row) * a.yrs + f_team + (li & 7) * 4;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const bool dma_first = tw >= 4;                       // younger half of the team: before its quads
    for (int t = t_begin; t < t_end; ++t) {
        // ---- phase A: region A of the team's stage holds step t once the team's pieces have landed
        SGP_PRE(wA, iA)
        bar_target += 8; team_barrier(bar, bar_target, lane);
        if (dma_first) dma_segment(x_step, piecesB);
        SGP_PHASE(wA, iA, nA, lastA, if (t > t_begin) SGP_EMIT)
        if (!dma_first) dma_segment(x_step, piecesB);
        // ---- phase B
        SGP_PRE(wB, iB)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar_target += 8; team_barrier(bar, bar_target, lane);
        if (dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, piecesA);
        SGP_PHASE(wB, iB, nB, lastB, )
        if (!dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, piecesA);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        x_step += x_inc;
    }
    SGP_EMIT
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

int duo_chunk_cap() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SGP_SPMM_CHUNK"); v = e ? atoi(e) : 32; if (v < 1) v = 32; }
    return v;
}

}  // namespace

extern "C" {

int32_t sgp_spmm_duo_max_quads(void) { return kMaxQuads; }

int sgp_spmm_duo_f32(const int32_t* uptr, const int32_t* ucol, const int32_t* usplit,
                     const int32_t* gptr, const int32_t* gsup, const uint16_t* gidx, const float* gw,
                     const int32_t* rowmap,
                     int32_t n_tiles, int32_t max_union, int32_t max_tile_quads,
                     const float* X, int64_t xrs, int64_t xbs,
                     float* Y, int64_t yrs, int64_t ybs,
                     int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                     sgp_stream_t stream) {
    SGP_REQUIRE(uptr && ucol && usplit && gptr && gsup && gidx && gw && rowmap && X && Y,
                "sgp_spmm_duo_f32: null pointer");
    SGP_REQUIRE(n_tiles >= 0 && n_rows >= 0 && batch >= 0 && max_union >= 0 && max_tile_quads >= 0,
                "sgp_spmm_duo_f32: bad size");
    // the DMA addresses a staged row as a 32-bit BYTE offset from the step base
    SGP_REQUIRE(n_cols >= 0 && (long long)n_cols * xrs < (1ll << 30),
                "sgp_spmm_duo_f32: row offsets exceed 32 bits (use sgp_spmm_csr_f32)");
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    if (feat % 64 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_duo_f32: feat=%d is not a multiple of 64", feat);
    if (max_union > kPasses * 64 || max_tile_quads > kMaxQuads)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_duo_f32: tile working set (%d rows, %d wave-quads) exceeds LDS (%d, %d)",
                         max_union, max_tile_quads, kPasses * 64, kMaxQuads);
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 && sgp::aligned16(X) &&
                sgp::aligned16(Y) && sgp::aligned16(gidx) && sgp::aligned16(gw),
                "sgp_spmm_duo_f32: strides/pointers must be 16-byte aligned");
    DuoArgs a;
    a.uptr = uptr; a.ucol = ucol; a.usplit = usplit; a.gptr = gptr; a.gsup = gsup; a.gidx = gidx; a.gw = gw;
    a.rowmap = rowmap;
    a.n_tiles = n_tiles;
    a.x = X; a.xrs = xrs; a.xbs = xbs;
    a.Y = Y; a.yrs = yrs; a.ybs = ybs;
    a.n_rows = n_rows; a.batch = batch; a.feat = feat;
    const int nft = feat / 64;
    long long want = (long long)batch * n_tiles * nft / 4096;
    int tc = (int)(want < 16 ? 16 : (want > duo_chunk_cap() ? duo_chunk_cap() : want));
    if (tc > batch) tc = batch;
    a.t_chunk = tc;
    a.n_tchunks = (batch + tc - 1) / tc;
    const size_t lds_bytes = 160 * 1024;
    auto kern = spmm_duo;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_duo: LDS opt-in: %s", hipGetErrorString(e));
    dim3 grid((unsigned)(a.n_tiles * a.n_tchunks), feat / 64);
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds_bytes, (hipStream_t)stream, a);
    return sgp::check_launch("spmm_duo");
}

}  // extern "C"
