// All-dense block form of the two-phase hop (reference call site: lib/sgp_preprocessing.py:200-203,
// x = adj @ x).  gfx950 / wave64 only.  Round-3 experiment behind `force="dense"`.
//
// Tiles, staged rows, the parity cut into segments A | B and the LDS-DMA staging are those of
// spmm_res / spmm_mix.  The arithmetic is the dense part of spmm_mix taken to its end: the 64 rows of a
// tile form 4 blocks of 16 rows and EVERY column a block uses goes through v_mfma_f32_16x16x4_f32 --
// 16 rows x 4 columns x 16 features per instruction, rows that lack a column carry weight 0 (a block of
// a 100-NN graph uses ~187 columns for 100 per row: 47 % of the products multiply zeros).  There is no
// 4x4x1 stream, no cross-lane fold and no exchange between waves:
//   * 8 waves per workgroup; wave w owns block w >> 1 and feature half w & 1 (16 rows x 32 features, two
//     accumulators).  A column quad costs it ONE ds_read2_b32 (the two 16-feature quarters of its half
//     sit 64 bytes apart in a staged row) and two MFMAs that share the weight operand;
//   * per quad the wave holds 2 registers (per-lane LDS address, weight operand: lane 16 k + i = weight
//     of (row i of the block, column k)), the first DHR quads of either phase for the whole time chunk;
//     longer lists read the rest from the plan arrays (L2) every step;
//   * results leave as 4-byte stores: lane (g, j), register r = row 4 g + r, features 32 fh + j and
//     32 fh + 16 + j -- 64-byte row pieces that the L2 merges with the neighbouring wave's.
// Why try it: the hop is clocked by power: the 4x4x1 form reads and writes 4
// accumulator registers per 256 FMAs and moves 16 bytes of LDS operand per lane and super-step; this
// form issues ~60 % fewer matrix instructions, moves ~60 % fewer LDS bytes and a quarter of the
// accumulator traffic for 1.46x the matrix-pipe cycles.
#include "common.h"
#include <stdlib.h>

using sgp::f32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

struct DenseArgs {
    const int* uptr; const int* ucol; const int* usplit; const int* rowmap;
    const int* dptr; const int* didx; const float* dw;
    int n_tiles;
    const float* x; long long xrs, xbs;
    float* Y; long long yrs, ybs;
    int n_rows, batch, feat;
    int t_chunk, n_tchunks;
};

constexpr int NW = 8;                                     // waves per workgroup
constexpr int RPP = NW * 4;                               // staged rows per DMA pass (one 1 KiB piece per wave)
constexpr int PASSES = 14;                                // x 32 staged rows = 448
constexpr int kStageBytes = PASSES * RPP * 256;           // 114 688

__device__ __forceinline__ void dma16_saddr(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}

// DHR: column quads per phase held in registers; DD: operand reads in flight
template <int DHR, int DD, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void spmm_dense(DenseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
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
    const int eg = tid >> 4;                              // 0 .. 31: staged row inside a pass
    const int k = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = wave >> 1, fh = wave & 1;
    const int u0 = a.uptr[tile];
    const int nU = a.uptr[tile + 1] - u0;
    const int uA = a.usplit[tile];
    const int t_begin = tchunk * a.t_chunk;
    const int t_end = min(a.batch, t_begin + a.t_chunk);
    if (t_begin >= t_end) return;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // ---- staging bookkeeping (as spmm_res, 8 waves: 14 passes of 32 rows)
    unsigned voff[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int u = p * RPP + eg;
        const int c = (u < nU) ? a.ucol[u0 + u] : (nU > 0 ? a.ucol[u0] : 0);
        voff[p] = (unsigned)(c * (int)a.xrs + f_base + li * 4) * 4u;
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
    const char* x_step = reinterpret_cast<const char*>(a.x + (long long)t_begin * a.xbs);
    const long long x_inc = a.xbs * 4;
    auto dma_segment = [&](const char* xt, unsigned pieces) {
        if constexpr (ABL & 1) return;
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
            if (pieces & (1u << p)) dma16_saddr(voff[p], xt, piece0 + (unsigned)p * (unsigned)(RPP * 256));
    };

    // ---- this wave's lists -> registers (once per workgroup)
    unsigned AD[2][DHR];
    float WT[2][DHR];
    int nd[2], first[2];
    const unsigned lane_off = lds0 + (unsigned)(fh * 128 + li * 4);
    {
        const int db = (tile * 4 + rb) * 2;
        const int d0 = __builtin_amdgcn_readfirstlane(a.dptr[db]);
        const int d1 = __builtin_amdgcn_readfirstlane(a.dptr[db + 1]);
        const int d2 = __builtin_amdgcn_readfirstlane(a.dptr[db + 2]);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int b = ph ? d1 : d0, e = ph ? d2 : d1;
            first[ph] = b;
            nd[ph] = e - b;
#pragma unroll
            for (int m = 0; m < DHR; ++m) {
                const bool in = b + m < e;
                AD[ph][m] = lane_off + (in ? (unsigned)a.didx[(long long)(b + m) * 4 + k] : 0u);
                WT[ph][m] = in ? a.dw[(long long)(b + m) * 64 + lane] : 0.f;
            }
        }
    }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int m = 0; m < DHR; ++m) asm volatile("" : "+v"(AD[ph][m]), "+v"(WT[ph][m]));
    }
    // result rows of this lane: slot 16 rb + 4 g + r, g = lane >> 4 (= k)
    unsigned yoff[4];
    unsigned row_ok = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = a.rowmap[tile * 64 + 16 * rb + 4 * k + r];
        if (row >= 0) row_ok |= 1u << r;
        yoff[r] = (unsigned)((long long)(row < 0 ? 0 : row) * a.yrs + f_base + 32 * fh + li) * 4u;
    }
    char* y_step = reinterpret_cast<char*>(a.Y + (long long)t_begin * a.ybs);
    const long long y_inc = a.ybs * 4;

    f32x2 ring[DD];
#pragma unroll
    for (int i = 0; i < DD; ++i) ring[i] = f32x2{0.f, 0.f};
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;

    // LDS traffic of a phase, in issue order (inline asm; LDS returns in order):
    //   R(0) .. R(min(DD, nd) - 1) | loop m: wait R(m) | 2 mfma | issue R(m + DD) if m + DD < min(nd, DHR).
    // Only reads that will be used are issued, so when iteration m starts exactly min(DD - 1, nd' - 1 - m)
    // reads younger than R(m) are in flight (nd' = min(nd, DHR)): the static count DD - 1 while m + DD <= nd',
    // a full drain in the last DD - 1 iterations.  Nothing is in flight when a phase ends -- a read that
    // returned after the compiler had moved on would land in whatever lives in its register by then.
#define SGP_RD(P_, M_) asm volatile("ds_read2_b32 %0, %1 offset1:16" : "+v"(ring[(M_) % DD]) : "v"(AD[P_][M_]))
#define SGP_WT(M_, N_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[(M_) % DD]) : "n"(N_))
    typedef const __attribute__((address_space(3))) float* lds_f1_t;
    auto overflow = [&](int ph) {                         // quads DHR .. nd - 1 straight from the plan arrays
        for (int m = DHR; m < nd[ph]; ++m) {
            const long long q = first[ph] + m;
            const unsigned ad = lane_off + (unsigned)a.didx[q * 4 + k];
            const float w = a.dw[q * 64 + lane];
            const float b0 = *(lds_f1_t)(size_t)ad, b1 = *(lds_f1_t)(size_t)(ad + 64);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b1, acc1, 0, 0, 0);
        }
    };
#define SGP_PHASE(P_)                                                                              \
    {                                                                                              \
        const int ndr = nd[P_] < DHR ? nd[P_] : DHR;                                               \
        { _Pragma("unroll") for (int m = 0; m < DD; ++m) { if (m < ndr) SGP_RD(P_, m); } }         \
        if (ndr > 0) {                                                                             \
            _Pragma("unroll") for (int m = 0; m < DHR; ++m) {                                      \
                if (m + DD <= ndr) { SGP_WT(m, DD - 1); } else { SGP_WT(m, 0); }                   \
                if ((P_) == 0 && m == 0) {                                                         \
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};                                          \
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(WT[P_][m], ring[m % DD].x, z, 0, 0, 0); \
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(WT[P_][m], ring[m % DD].y, z, 0, 0, 0); \
                } else {                                                                           \
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(WT[P_][m], ring[m % DD].x, acc0, 0, 0, 0); \
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(WT[P_][m], ring[m % DD].y, acc1, 0, 0, 0); \
                }                                                                                  \
                if (m + DD < DHR) { if (m + DD < ndr) SGP_RD(P_, m + DD); }                        \
                if (m + 1 == ndr) break;                                                           \
            }                                                                                      \
            if (nd[P_] > DHR) overflow(P_);                                                        \
        } else if ((P_) == 0) {                                                                    \
            acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0;                                         \
        }                                                                                          \
    }

    __syncthreads();
    dma_segment(x_step, piecesA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool dma_first = (wave & 1) != 0;               // one wave of every SIMD pair refills before, one after its quads
    for (int t = t_begin; t < t_end; ++t) {
        asm volatile("" : "+s"(nd[0]), "+s"(nd[1]));
        // ---- phase A: segment A holds step t once every wave's pieces have landed
        asm volatile("s_barrier" ::: "memory");
        if (dma_first) dma_segment(x_step, piecesB);
        SGP_PHASE(0)
        if (!dma_first) dma_segment(x_step, piecesB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- phase B
        asm volatile("s_barrier" ::: "memory");
        if (dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, piecesA);
        SGP_PHASE(1)
        if (!dma_first && t + 1 < t_end) dma_segment(x_step + x_inc, piecesA);
        // rows out: lane (g, j), register r = (row 4 g + r, features 32 fh + j | + 16)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (row_ok & (1u << r)) {
                float* yp = reinterpret_cast<float*>(y_step + yoff[r]);
                yp[0] = acc0[r];
                yp[16] = acc1[r];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        x_step += x_inc; y_step += y_inc;
    }
#undef SGP_PHASE
#undef SGP_WT
#undef SGP_RD
}

int dense_chunk_cap() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SGP_SPMM_CHUNK"); v = e ? atoi(e) : 32; if (v < 1) v = 32; }
    return v;
}

constexpr int kDHR = 34;

}  // namespace

extern "C" {

int32_t sgp_spmm_dense_max_union(void) { return PASSES * RPP; }

int sgp_spmm_dense_f32(const int32_t* uptr, const int32_t* ucol, const int32_t* usplit, const int32_t* rowmap,
                       const int32_t* dptr, const int32_t* didx, const float* dw,
                       int32_t n_tiles, int32_t max_union,
                       const float* X, int64_t xrs, int64_t xbs,
                       float* Y, int64_t yrs, int64_t ybs,
                       int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                       sgp_stream_t stream) {
    SGP_REQUIRE(uptr && ucol && usplit && rowmap && dptr && didx && dw && X && Y, "sgp_spmm_dense_f32: null pointer");
    SGP_REQUIRE(n_tiles >= 0 && n_rows >= 0 && n_cols >= 0 && batch >= 0 && max_union >= 0, "sgp_spmm_dense_f32: bad size");
    SGP_REQUIRE((long long)n_cols * xrs < (1ll << 30) && (long long)n_rows * yrs < (1ll << 30),
                "sgp_spmm_dense_f32: row offsets exceed 32 bits (use sgp_spmm_csr_f32)");
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    if (feat % 64 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_dense_f32: feat=%d is not a multiple of 64", feat);
    if (max_union > sgp_spmm_dense_max_union())
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_dense_f32: %d staged rows per tile exceed %d", max_union,
                         sgp_spmm_dense_max_union());
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 && sgp::aligned16(X) && sgp::aligned16(Y),
                "sgp_spmm_dense_f32: strides/pointers must be 16-byte aligned");
    DenseArgs a;
    a.uptr = uptr; a.ucol = ucol; a.usplit = usplit; a.rowmap = rowmap; a.dptr = dptr; a.didx = didx; a.dw = dw;
    a.n_tiles = n_tiles; a.x = X; a.xrs = xrs; a.xbs = xbs; a.Y = Y; a.yrs = yrs; a.ybs = ybs;
    a.n_rows = n_rows; a.batch = batch; a.feat = feat;
    const int nft = feat / 64;
    long long want = (long long)batch * n_tiles * nft / 4096;
    int tc = (int)(want < 16 ? 16 : (want > dense_chunk_cap() ? dense_chunk_cap() : want));
    if (tc > batch) tc = batch;
    a.t_chunk = tc;
    a.n_tchunks = (batch + tc - 1) / tc;
    const size_t lds_bytes = kStageBytes;
    dim3 grid((unsigned)(a.n_tiles * a.n_tchunks), nft);
    hipStream_t s = (hipStream_t)stream;
#ifdef SGP_ABLATION
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("SGP_PIPE_ABL"); abl = e ? atoi(e) : 0; }
    if (abl == 1) {
        auto k1 = spmm_dense<kDHR, 4, 1>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k1, grid, dim3(NW * 64), lds_bytes, s, a);
        return sgp::check_launch("spmm_dense");
    }
#endif
    auto kern = spmm_dense<kDHR, 4>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_dense: LDS opt-in: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, s, a);
    return sgp::check_launch("spmm_dense");
}

}  // extern "C"
