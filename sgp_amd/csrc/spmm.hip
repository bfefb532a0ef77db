// Batched CSR SpMM for the K-hop propagation  x <- A_hat . x  (reference call site:
// lib/sgp_preprocessing.py:202, torch_sparse spmm_sum).  gfx950 / wave64 only.
//
//   Y[b, i, :] = sum_e val[e] * X[b, col[e], :]       b = time step, i = node
//
// Three kernels:
//   spmm_csr_rows   generic: one wave per (row, block of TB steps); 16-byte gathers straight
//                   from L1/L2; any graph, feat % 4 == 0.
//   spmm_csr_scalar fallback for feat % 4 != 0 or unaligned strides (lane = feature).
//   spmm_tiled      locality path: a workgroup owns a tile of consecutive rows, stages the
//                   distinct source rows of that tile for one time step in LDS (register
//                   prefetch of step t+1 under the compute of step t), and every 16-lane DPP
//                   row walks 16 edges per ds-free rotation: edge records are fetched once
//                   per 16 edges and handed round the row with row_ror, so the only LDS
//                   traffic in the inner loop is one ds_read_b128 per (edge, 4 features).
#include "common.h"
#include <stdlib.h>

using sgp::f32x4;

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

struct Src {                      // where column c of step b lives
    const float* x;  long long xrs, xbs;
    const float* xh; long long xhrs, xhbs;
    int n_own;
    const int* pred; int pred_want;   // launch predicate (common.h)
    __device__ __forceinline__ bool skip() const { return pred != nullptr && pred[0] != pred_want; }
    __device__ __forceinline__ const float* row(int b, int c) const {
        return (c < n_own) ? x + (long long)b * xbs + (long long)c * xrs
                           : xh + (long long)b * xhbs + (long long)(c - n_own) * xhrs;
    }
};

// ---------------------------------------------------------------- generic rows kernel
// LPR lanes cover one source row chunk of 4*LPR floats; G = 64/LPR edges are in flight per
// wave instruction; TB time steps share every (col, val) fetch.
template <int LPR, int TB>
__device__ __forceinline__ void csr_rows_body(
        const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
        const Src& src, float* __restrict__ Y, long long yrs, long long ybs,
        int batch, int feat, int row, int b0) {
    constexpr int G = kWave / LPR;
    const int lane = threadIdx.x & 63;
    const int f0 = blockIdx.z * (4 * LPR) + (lane % LPR) * 4;
    const int g = lane / LPR;
    const bool fok = f0 < feat;

    f32x4 acc[TB];
#pragma unroll
    for (int i = 0; i < TB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int e1 = rowptr[row + 1];
    for (int e = rowptr[row] + g; e < e1; e += G) {
        const int c = col[e];
        const float v = val[e];
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            if (fok && b0 + i < batch) {
                const f32x4 xv = ld4(src.row(b0 + i, c) + f0);
                acc[i] += v * xv;
            }
        }
    }
    // fold the G edge-groups together
#pragma unroll
    for (int i = 0; i < TB; ++i) {
#pragma unroll
        for (int off = LPR; off < kWave; off <<= 1) {
            acc[i].x += __shfl_xor(acc[i].x, off);
            acc[i].y += __shfl_xor(acc[i].y, off);
            acc[i].z += __shfl_xor(acc[i].z, off);
            acc[i].w += __shfl_xor(acc[i].w, off);
        }
    }
    // group i % G writes step i (spreads the stores over the wave)
#pragma unroll
    for (int i = 0; i < TB; ++i) {
        if (fok && b0 + i < batch && g == (i % G))
            st4(Y + (long long)(b0 + i) * ybs + (long long)row * yrs + f0, acc[i]);
    }
}

template <int LPR, int TB>
__global__ __launch_bounds__(256) void spmm_csr_rows(
        const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
        Src src, float* __restrict__ Y, long long yrs, long long ybs,
        int n_rows, int batch, int feat) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows || src.skip()) return;
    csr_rows_body<LPR, TB>(rowptr, col, val, src, Y, yrs, ybs, batch, feat, row, blockIdx.y * TB);
}

// The same product under a launch predicate: a BOUNDED grid that strides over (row block, batch block).  A predicated
// launch that does not run still dispatches every workgroup to its first instruction -- 6.4 million of them for the
// direct form on the target line (N = 100 000, T = 1024): 1.43 ms per skipped launch, measured behind every split hop
// when this kernel became the default fallback (round 6).  2048 x 8 workgroups exit in microseconds.
template <int LPR, int TB>
__global__ __launch_bounds__(256) void spmm_csr_rows_strided(
        const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
        Src src, float* __restrict__ Y, long long yrs, long long ybs,
        int n_rows, int batch, int feat) {
    if (src.skip()) return;
    const int n_rb = (n_rows + 3) / 4, n_bb = (batch + TB - 1) / TB;
    for (int bb = blockIdx.y; bb < n_bb; bb += gridDim.y)
        for (int rb = blockIdx.x; rb < n_rb; rb += gridDim.x) {
            const int row = rb * 4 + (threadIdx.x >> 6);
            if (row < n_rows) csr_rows_body<LPR, TB>(rowptr, col, val, src, Y, yrs, ybs, batch, feat, row, bb * TB);
        }
}

__global__ __launch_bounds__(256) void spmm_csr_scalar(
        const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
        Src src, float* __restrict__ Y, long long yrs, long long ybs,
        int n_rows, int batch, int feat) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (row >= n_rows || src.skip()) return;
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    for (int f = lane; f < feat; f += kWave) {
        float acc = 0.f;
        for (int e = e0; e < e1; ++e) acc += val[e] * src.row(b, col[e])[f];
        Y[(long long)b * ybs + (long long)row * yrs + f] = acc;
    }
}

// ---------------------------------------------------------------- tiled kernel
template <int S>
__device__ __forceinline__ int ror_i(int v) {
    if constexpr (S == 0) return v;
    else return __builtin_amdgcn_update_dpp(0, v, 0x120 + S, 0xf, 0xf, true);    // row_ror:S
}
template <int S>
__device__ __forceinline__ float ror_f(float v) {
    return __int_as_float(ror_i<S>(__float_as_int(v)));
}

// 8 rotations of one 16-edge batch: addresses first, then 8 LDS reads in flight, then FMAs.
template <int S0>
__device__ __forceinline__ void edge_half(const char* lds, int off, float w, int li16, f32x4& acc) {
    int ad[8];
    float wv[8];
    f32x4 xv[8];
    // the row offset travels round the DPP row; every lane adds its OWN 16-byte column slot
#define SGP_ROT(i) ad[i] = ror_i<S0 + i>(off) + li16; wv[i] = ror_f<S0 + i>(w);
    SGP_ROT(0) SGP_ROT(1) SGP_ROT(2) SGP_ROT(3) SGP_ROT(4) SGP_ROT(5) SGP_ROT(6) SGP_ROT(7)
#undef SGP_ROT
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const f32x4*>(lds + ad[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += wv[i] * xv[i];
}

// Same 8 rotations with the weight rotation folded into the FMA: v_fmac_f32_dpp takes its
// src0 (the edge weight) through row_ror, so a step is add_dpp + ds_read_b128 + 4 fmac_dpp.
#define SGP_FMAC_ROR(S, ACC, W, X)                                                              \
    asm("v_fmac_f32_dpp %0, %1, %2 row_ror:" #S " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(W), "v"(X))
#define SGP_STEP_ROR(S, I)                                                                      \
    SGP_FMAC_ROR(S, acc.x, w, xv[I].x); SGP_FMAC_ROR(S, acc.y, w, xv[I].y);                     \
    SGP_FMAC_ROR(S, acc.z, w, xv[I].z); SGP_FMAC_ROR(S, acc.w, w, xv[I].w);
template <int S0, int ABL = 0>
__device__ __forceinline__ void edge_half_dpp(const char* lds, int off, float w, int li16, f32x4& acc) {
    int ad[8];
    f32x4 xv[8];
#define SGP_ROT(i) ad[i] = ror_i<S0 + i>(off) + li16;
    SGP_ROT(0) SGP_ROT(1) SGP_ROT(2) SGP_ROT(3) SGP_ROT(4) SGP_ROT(5) SGP_ROT(6) SGP_ROT(7)
#undef SGP_ROT
    if constexpr (ABL == 3) {          // ablation: no LDS reads (operands faked from the address)
#pragma unroll
        for (int i = 0; i < 8; ++i) { float f = __int_as_float(ad[i]); xv[i] = f32x4{f, f, f, f}; }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const f32x4*>(lds + ad[i]);
    }
    if constexpr (ABL == 2) {          // ablation: no FMAs (reads kept alive)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(xv[i].x), "v"(xv[i].y), "v"(xv[i].z), "v"(xv[i].w));
        return;
    }
    if constexpr (S0 == 0) {
        acc += w * xv[0];
        SGP_STEP_ROR(1, 1) SGP_STEP_ROR(2, 2) SGP_STEP_ROR(3, 3) SGP_STEP_ROR(4, 4)
        SGP_STEP_ROR(5, 5) SGP_STEP_ROR(6, 6) SGP_STEP_ROR(7, 7)
    } else {
        SGP_STEP_ROR(8, 0) SGP_STEP_ROR(9, 1) SGP_STEP_ROR(10, 2) SGP_STEP_ROR(11, 3)
        SGP_STEP_ROR(12, 4) SGP_STEP_ROR(13, 5) SGP_STEP_ROR(14, 6) SGP_STEP_ROR(15, 7)
    }
}

// Quarter batch (4 rotations) for the tall-tile form, where registers are short: 20 temporaries
// instead of 40.
template <int S0>
__device__ __forceinline__ void edge_quarter_dpp(const char* lds, int off, float w, int li16, f32x4& acc) {
    int ad[4];
    f32x4 xv[4];
#define SGP_ROT(i) ad[i] = ror_i<S0 + i>(off) + li16;
    SGP_ROT(0) SGP_ROT(1) SGP_ROT(2) SGP_ROT(3)
#undef SGP_ROT
#pragma unroll
    for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const f32x4*>(lds + ad[i]);
    if constexpr (S0 == 0) {
        acc += w * xv[0];
        SGP_STEP_ROR(1, 1) SGP_STEP_ROR(2, 2) SGP_STEP_ROR(3, 3)
    } else if constexpr (S0 == 4) {
        SGP_STEP_ROR(4, 0) SGP_STEP_ROR(5, 1) SGP_STEP_ROR(6, 2) SGP_STEP_ROR(7, 3)
    } else if constexpr (S0 == 8) {
        SGP_STEP_ROR(8, 0) SGP_STEP_ROR(9, 1) SGP_STEP_ROR(10, 2) SGP_STEP_ROR(11, 3)
    } else {
        SGP_STEP_ROR(12, 0) SGP_STEP_ROR(13, 1) SGP_STEP_ROR(14, 2) SGP_STEP_ROR(15, 3)
    }
}

struct TiledArgs {
    const int* trow; const int* uptr; const int* ucol; const int* erow; const unsigned short* ecol; const float* eval;
    int tile_rows, n_tiles;
    Src src;
    float* Y; long long yrs, ybs;
    int n_rows, batch, feat;
    int t_chunk, n_tchunks;
    int stage_bytes;                   // tall tiles: the edge records live in LDS behind the stage
};

// Feature tile FT = 64 floats (16 lanes x 16 B per staged row).  NTHR threads = NTHR/16 edge
// groups (DPP rows).  PASSES: staged rows per thread (register prefetch depth), capacity
// U_MAX = PASSES * NTHR / 16 rows.  Each edge group owns RPG rows of the tile and keeps their
// edge records (NB batches of 16 per row) in registers for the whole time chunk.
template <int NTHR, int PASSES, int RPG, int NB, bool HALO, int VARIANT>
__global__ __launch_bounds__(NTHR) void spmm_tiled(TiledArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int FT = 64;
    constexpr int RPP = NTHR / 16;                // rows staged per pass
    constexpr int NEG = NTHR / 16;                // edge groups per workgroup

    // XCD-aware decode: consecutive ids on one XCD = consecutive tiles of one time chunk
    if (a.src.skip()) return;
    const int nwg = a.n_tiles * a.n_tchunks;
    const int orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tile = w % a.n_tiles;
    const int tchunk = w / a.n_tiles;
    const int f_base = blockIdx.y * FT;

    const int tid = threadIdx.x;
    const int li = tid & 15;                      // 16-byte slot inside a row; lane of the DPP row
    const int eg = tid >> 4;                      // staged row in a pass == edge group id
    const int u0 = a.uptr[tile];
    const int nU = a.uptr[tile + 1] - u0;

    // per-thread source rows: element offset from the step base, fixed for the time chunk.
    // Slots past the tile's list read row 0 (harmless) so the loads stay branch-free.
    // (32-bit element offsets: the host checks n_cols * row_stride < 2^31.)
    int soff[PASSES];
    unsigned halo_mask = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int u = p * RPP + eg;
        const int c = (u < nU) ? a.ucol[u0 + u] : 0;
        if (HALO && c >= a.src.n_own) {
            halo_mask |= 1u << p;
            soff[p] = (c - a.src.n_own) * (int)a.src.xhrs;
        } else {
            soff[p] = c * (int)a.src.xrs;
        }
    }
    const int n_pass = (nU + RPP - 1) / RPP;      // workgroup-uniform

    const int t_begin = tchunk * a.t_chunk;
    const int t_end = min(a.batch, t_begin + a.t_chunk);
    if (t_begin >= t_end) return;

    // my rows' edge records -> registers (LDS byte offset of the source row + weight).  Tall tiles
    // (RPG > 2: small sparse graphs staged whole, see graph.ShiftOperator.tile_plan) keep them in LDS
    // behind the stage instead -- 2 RPG NB registers would spill -- as u16 staged-row index + weight.
    constexpr bool TALL = RPG > 2;
    const int row0 = a.trow[tile];
    const int rows_here = a.trow[tile + 1] - row0;
    int eoff[TALL ? 1 : RPG][NB];
    float ewv[TALL ? 1 : RPG][NB];
    int nbat[RPG];
    float* rec_w = reinterpret_cast<float*>(lds + a.stage_bytes);                         // [RPG * NEG][NB][16]
    unsigned short* rec_o = reinterpret_cast<unsigned short*>(rec_w + RPG * NEG * NB * 16);
#pragma unroll
    for (int g = 0; g < RPG; ++g) {
        const int rr = eg + g * NEG;
        int eb = 0, ee = 0;
        if (rr < rows_here) { eb = a.erow[row0 + rr]; ee = a.erow[row0 + rr + 1]; }
        nbat[g] = (ee - eb) >> 4;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const bool on = n < nbat[g];
            if constexpr (TALL) {
                rec_w[(rr * NB + n) * 16 + li] = on ? a.eval[eb + n * 16 + li] : 0.f;
                rec_o[(rr * NB + n) * 16 + li] = on ? a.ecol[eb + n * 16 + li] : (unsigned short)0;
            } else {
                eoff[g][n] = on ? (int)a.ecol[eb + n * 16 + li] * (FT * 4) : 0;
                ewv[g][n] = on ? a.eval[eb + n * 16 + li] : 0.f;
            }
        }
    }

    constexpr bool kStage = VARIANT != 4 && VARIANT != 5;
    constexpr bool kBarrier = VARIANT != 5;
    f32x4 stage[PASSES];
    auto issue = [&](int t) {
        if constexpr (!kStage) return;
        const float* xt = a.src.x + (long long)t * a.src.xbs + f_base + li * 4;
        const float* ht = a.src.xh + (long long)t * a.src.xhbs + f_base + li * 4;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if (p < n_pass) {
                const float* b = (HALO && ((halo_mask >> p) & 1u)) ? ht : xt;
                stage[p] = ld4(b + soff[p]);
            }
        }
    };
    issue(t_begin);

    // The result rows of step t are kept in registers and stored at the top of step t+1, after the
    // wait that retires the staged loads: loads and stores share vmcnt, and a store issued at the
    // end of step t would make that wait sit out a full store round trip every step.
    f32x4 res[RPG];
#pragma unroll
    for (int g = 0; g < RPG; ++g) res[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto store_rows = [&](int t) {
#pragma unroll
        for (int g = 0; g < RPG; ++g) {
            const int rr = eg + g * NEG;
            if (rr < rows_here)
                st4(a.Y + (long long)t * a.ybs + (long long)(row0 + rr) * a.yrs + f_base + li * 4, res[g]);
        }
    };

    for (int t = t_begin; t < t_end; ++t) {
        if constexpr (kStage) {
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
                if (p < n_pass)
                    *reinterpret_cast<f32x4*>(lds + ((p * RPP + eg) * FT + li * 4) * 4) = stage[p];
        }
        if (t > t_begin) store_rows(t - 1);
        // bare barrier: hipcc's __syncthreads also drains vmcnt, i.e. it would sit out the round
        // trip of the stores issued one line above on every step
        if constexpr (kBarrier) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 1 < t_end) issue(t + 1);          // in flight under the compute below

#pragma unroll
        for (int g = 0; g < RPG; ++g) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                if (n < nbat[g]) {
                    int eo; float ew;
                    if constexpr (TALL) {
                        const int rr = eg + g * NEG;
                        eo = (int)rec_o[(rr * NB + n) * 16 + li] * (FT * 4);
                        ew = rec_w[(rr * NB + n) * 16 + li];
                    } else {
                        eo = eoff[g][n]; ew = ewv[g][n];
                    }
                    if constexpr (TALL) {
                        edge_quarter_dpp<0>(lds, eo, ew, li * 16, acc);
                        edge_quarter_dpp<4>(lds, eo, ew, li * 16, acc);
                        edge_quarter_dpp<8>(lds, eo, ew, li * 16, acc);
                        edge_quarter_dpp<12>(lds, eo, ew, li * 16, acc);
                    } else if constexpr (VARIANT >= 1) {
                        constexpr int ABL = (VARIANT == 2 || VARIANT == 3) ? VARIANT : 0;
                        edge_half_dpp<0, ABL>(lds, eo, ew, li * 16, acc);
                        edge_half_dpp<8, ABL>(lds, eo, ew, li * 16, acc);
                    } else {
                        edge_half<0>(lds, eo, ew, li * 16, acc);
                        edge_half<8>(lds, eo, ew, li * 16, acc);
                    }
                }
            }
            res[g] = acc;
        }
        // all reads done before the next overwrite (the staged loads of step t+1 are waited for by
        // the ds_write that consumes them, not here)
        if constexpr (kBarrier) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    store_rows(t_end - 1);
}

int tiled_variant();
int chunk_cap();

constexpr int kTiledThreads = 1024;
constexpr int kTiledPasses = 8;
constexpr int kTiledCapacity = kTiledPasses * kTiledThreads / 16;   // staged rows per tile
constexpr int kTiledGroups = kTiledThreads / 16;                    // rows in flight per workgroup

int chunk_cap() {
    static int v = -1;
    if (v < 0) { v = (int)sgp::tune("spmm_chunk", 32); if (v < 1) v = 32; }
    return v;
}

int tiled_variant() {
    static int v = -1;
    if (v < 0) v = (int)sgp::tune("spmm_variant", 1);
    return v;
}

template <int RPG, int NB, bool HALO, int VARIANT>
int launch_tiled_v(const TiledArgs& a, hipStream_t s) {
    const size_t lds_bytes = RPG > 2 ? (size_t)a.stage_bytes + (size_t)RPG * kTiledGroups * NB * 16 * 6
                                     : (size_t)kTiledCapacity * 64 * 4;
    if (lds_bytes > 160 * 1024)
        return sgp::fail(SGP_EUNSUP, "spmm_tiled: tall tile needs %zu bytes of LDS (stage + edge records)", lds_bytes);
    auto kern = spmm_tiled<kTiledThreads, kTiledPasses, RPG, NB, HALO, VARIANT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_tiled: LDS opt-in: %s", hipGetErrorString(e));
    dim3 grid(a.n_tiles * a.n_tchunks, a.feat / 64);
    hipLaunchKernelGGL(kern, grid, dim3(kTiledThreads), lds_bytes, s, a);
    return sgp::check_launch("spmm_tiled");
}

template <int RPG, int NB, bool HALO>
int launch_tiled(const TiledArgs& a, hipStream_t s) {
    switch (tiled_variant()) {
        case 0: return launch_tiled_v<RPG, NB, HALO, 0>(a, s);
#ifdef SGP_ABLATION
        case 2: return launch_tiled_v<RPG, NB, HALO, 2>(a, s);
        case 3: return launch_tiled_v<RPG, NB, HALO, 3>(a, s);
        case 4: return launch_tiled_v<RPG, NB, HALO, 4>(a, s);
        case 5: return launch_tiled_v<RPG, NB, HALO, 5>(a, s);
#endif
        default: return launch_tiled_v<RPG, NB, HALO, 1>(a, s);
    }
}

template <bool HALO>
int dispatch_tiled(const TiledArgs& a, int rpg, int nb, hipStream_t s) {
#define SGP_T(R, B) if (rpg == R && nb == B) return launch_tiled<R, B, HALO>(a, s);
    SGP_T(1, 2) SGP_T(2, 2) SGP_T(1, 8) SGP_T(4, 1) SGP_T(6, 1) SGP_T(4, 2) SGP_T(6, 2)
#undef SGP_T
    return sgp::fail(SGP_EUNSUP, "spmm_tiled: no kernel for rows/group=%d batches=%d", rpg, nb);
}

}  // namespace

extern "C" {

int32_t sgp_spmm_tiled_max_union(int32_t feat) {
    return (feat > 0 && feat % 64 == 0) ? kTiledCapacity : 0;
}
int32_t sgp_spmm_tiled_max_tile_rows(void) { return 6 * kTiledGroups; }   // (rows / group 1, 2, 4, 6)
int32_t sgp_spmm_tiled_max_row_edges(void) { return 8 * 16; }

int sgp_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                     const float* X, int64_t xrs, int64_t xbs,
                     const float* Xh, int64_t xhrs, int64_t xhbs, int32_t n_own,
                     float* Y, int64_t yrs, int64_t ybs,
                     int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                     const int32_t* pred, int32_t run_if, sgp_stream_t stream) {
    const sgp::Predicate pr{pred, run_if};
    SGP_REQUIRE(rowptr && col && val && X && Y, "sgp_spmm_csr_f32: null pointer");
    SGP_REQUIRE(n_rows >= 0 && n_cols >= 0 && batch >= 0 && feat >= 0, "sgp_spmm_csr_f32: negative size");
    SGP_REQUIRE(Xh != nullptr || n_own >= n_cols, "sgp_spmm_csr_f32: n_own < n_cols needs X_halo");
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    Src src{X, xrs, xbs, Xh ? Xh : X, xhrs, xhbs, Xh ? n_own : 0x7fffffff, pr.flag, pr.want};
    const bool vec = feat % 4 == 0 && xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 &&
                     sgp::aligned16(X) && sgp::aligned16(Y) &&
                     (!Xh || (xhrs % 4 == 0 && xhbs % 4 == 0 && sgp::aligned16(Xh)));
    const unsigned gx = (n_rows + 3) / 4;
    if (!vec) {
        SGP_REQUIRE(batch <= 65535, "sgp_spmm_csr_f32: scalar path supports batch <= 65535");
        hipLaunchKernelGGL(spmm_csr_scalar, dim3(gx, batch), dim3(256), 0, s,
                           rowptr, col, val, src, Y, yrs, ybs, n_rows, batch, feat);
        return sgp::check_launch("spmm_csr_scalar");
    }
    constexpr int TB = 4;
    const unsigned gy = (batch + TB - 1) / TB;
    SGP_REQUIRE(gy <= 65535, "sgp_spmm_csr_f32: batch too large for one launch (chunk it)");
    const bool bounded = pr.flag != nullptr;       // predicated: the grid a skipped launch must retire stays small
#define SGP_ROWS(LPR)                                                                          \
    do {                                                                                       \
        if (bounded)                                                                           \
            hipLaunchKernelGGL((spmm_csr_rows_strided<LPR, TB>),                               \
                               dim3(gx < 2048u ? gx : 2048u, gy < 8u ? gy : 8u, (feat + 4 * LPR - 1) / (4 * LPR)), \
                               dim3(256), 0, s, rowptr, col, val, src, Y, yrs, ybs, n_rows, batch, feat);          \
        else                                                                                   \
            hipLaunchKernelGGL((spmm_csr_rows<LPR, TB>), dim3(gx, gy, (feat + 4 * LPR - 1) / (4 * LPR)), \
                               dim3(256), 0, s, rowptr, col, val, src, Y, yrs, ybs, n_rows, batch, feat);  \
    } while (0)
    if (feat >= 256 || feat % 256 == 0) SGP_ROWS(64);
    else if (feat > 64) SGP_ROWS(32);
    else if (feat > 32) SGP_ROWS(16);
    else if (feat > 16) SGP_ROWS(8);
    else SGP_ROWS(4);
#undef SGP_ROWS
    return sgp::check_launch("spmm_csr_rows");
}

int sgp_spmm_tiled_f32(const int32_t* trow, const int32_t* uptr, const int32_t* ucol,
                       const int32_t* erow, const uint16_t* ecol, const float* eval,
                       int32_t tile_rows, int32_t n_tiles, int32_t max_union, int32_t max_row_edges,
                       const float* X, int64_t xrs, int64_t xbs,
                       const float* Xh, int64_t xhrs, int64_t xhbs, int32_t n_own,
                       float* Y, int64_t yrs, int64_t ybs,
                       int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                       const int32_t* pred, int32_t run_if, sgp_stream_t stream) {
    const sgp::Predicate pr{pred, run_if};
    SGP_REQUIRE(trow && uptr && ucol && erow && ecol && eval && X && Y, "sgp_spmm_tiled_f32: null pointer");
    SGP_REQUIRE(tile_rows > 0 && n_tiles >= 0 && n_rows >= 0 && batch >= 0 && max_union >= 0 &&
                max_row_edges >= 0 && max_row_edges % 16 == 0, "sgp_spmm_tiled_f32: bad size");
    SGP_REQUIRE((long long)tile_rows * n_tiles >= n_rows, "sgp_spmm_tiled_f32: tiles do not cover rows");
    {
        const long long own = Xh ? n_own : n_cols, far = Xh ? n_cols - n_own : 0;
        SGP_REQUIRE(n_cols >= 0 && own >= 0 && far >= 0 && own * xrs < (1ll << 31) && far * xhrs < (1ll << 31),
                    "sgp_spmm_tiled_f32: row offsets exceed 32 bits (use sgp_spmm_csr_f32)");
    }
    if (n_rows == 0 || batch == 0 || feat == 0) return 0;
    if (feat % 64 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_tiled_f32: feat=%d is not a multiple of 64", feat);
    if (max_union > kTiledCapacity)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_tiled_f32: a tile references %d distinct rows, LDS stage holds %d",
                         max_union, kTiledCapacity);
    if (tile_rows > sgp_spmm_tiled_max_tile_rows() || max_row_edges > sgp_spmm_tiled_max_row_edges())
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_tiled_f32: tile_rows=%d / max_row_edges=%d out of range",
                         tile_rows, max_row_edges);
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 &&
                sgp::aligned16(X) && sgp::aligned16(Y) &&
                (!Xh || (xhrs % 4 == 0 && xhbs % 4 == 0 && sgp::aligned16(Xh))),
                "sgp_spmm_tiled_f32: strides/pointers must be 16-byte aligned");
    TiledArgs a;
    a.trow = trow; a.uptr = uptr; a.ucol = ucol; a.erow = erow; a.ecol = ecol; a.eval = eval;
    a.tile_rows = tile_rows; a.n_tiles = n_tiles;
    a.src = Src{X, xrs, xbs, Xh ? Xh : X, xhrs, xhbs, Xh ? n_own : 0x7fffffff, pr.flag, pr.want};
    a.Y = Y; a.yrs = yrs; a.ybs = ybs;
    a.n_rows = n_rows; a.batch = batch; a.feat = feat;
    // enough workgroups to balance 256 CUs, long enough chunks to amortise the per-tile setup
    // Time chunk per workgroup: long enough to amortise the per-tile setup, short enough that
    // neighbouring tiles (which share source rows through L2 / Infinity Cache) cannot drift
    // far apart in t -- with 431-step chunks rocprofv3 showed 2.4x the algorithmic HBM reads.
    const int nft = feat / 64;
    long long want = (long long)batch * n_tiles * nft / 4096;
    int tc = (int)(want < 16 ? 16 : (want > chunk_cap() ? chunk_cap() : want));
    if (tc > batch) tc = batch;
    a.t_chunk = tc;
    a.n_tchunks = (batch + tc - 1) / tc;
    a.stage_bytes = ((max_union + 63) / 64 * 64) * 256;   // whole passes of 64 staged rows
    int rpg = (tile_rows + kTiledGroups - 1) / kTiledGroups;
    rpg = rpg <= 2 ? rpg : (rpg <= 4 ? 4 : 6);          // (the kernel masks the rows a group does not have)
    const int nb = (rpg > 2 && max_row_edges <= 16) ? 1 : (max_row_edges <= 32 ? 2 : 8);
    hipStream_t s = (hipStream_t)stream;
    return Xh ? dispatch_tiled<true>(a, rpg, nb, s) : dispatch_tiled<false>(a, rpg, nb, s);
}

}  // extern "C"
