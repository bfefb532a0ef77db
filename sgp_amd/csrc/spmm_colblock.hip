// Column-blocked hop for graphs WITHOUT locality (reference call site: lib/sgp_preprocessing.py:202,
// x = adj @ x; SURVEY.md 8d's adversarial "random sparse A").  gfx950 / wave64 only.
//
// A tile of such a graph references as many distinct source rows as it has edges, so there is
// nothing to stage: every (edge, step) pair gathers its own 256-byte row, and with all rows of a time
// step in play (N = 100k: 25.6 MB) the gathers miss the 4 MB L2 of their XCD and are served by the
// Infinity Cache (generic CSR kernel: 7.5 TB/s of gathers, 1.9 % of the HBM roofline).  Here the
// COLUMNS are cut into blocks whose source rows fit an L2 (plan: sgp_amd/colblock.py) and every
// workgroup of the chip sweeps the blocks in the same order, once per time step: while block b is
// being read, all 32 CUs of an XCD gather from the same 2.5 MB, i.e. from their L2.
//   * one persistent workgroup (16 waves) per CU owns a contiguous range of <= 512 rows of equal edge
//     count for ALL time steps; its partial sums live in LDS ([row][64 floats], 128 KB), so nothing is
//     read-modified-written in HBM between blocks;
//   * the 64 (wave, lane group) slots of a workgroup each own the rows r = slot (mod 64) of the range
//     and nobody else touches them: a slot walks its own edge list of the block, sorted by row, keeps
//     the running sum of a row in registers and adds it to the row's LDS sum when the row's run ends
//     (plain ds_read_b128 / add / ds_write_b128 under the slot's own exec mask -- no atomics: LDS
//     float atomics measured ~100x slower here, 840 ms per hop instead of 88 for the generic kernel);
//   * an edge is 8 bytes of plan {column | local row << 22 | last-of-run << 31, weight}; a wave
//     instruction gathers 4 edges (16 lanes x 16 bytes each).  The slots' lists of a (workgroup, block)
//     segment are padded to one length with weight-0 entries that "end a run" of local row 511, a row
//     nobody owns: whatever a padding entry read (0 * NaN) is dropped there and cannot leak into the
//     next row's sum;
//   * end of a step: barrier, rows out (16-byte stores), sums cleared, barrier -- and, when every
//     workgroup of the launch is resident (one per CU), a rendezvous of all of them: a workgroup's
//     stream is a few per cent longer or shorter than its neighbour's, and without it the workgroups
//     of an XCD drift blocks apart within a few hundred steps -- their working set then spans several
//     blocks and the gathers miss again (measured: 36 % L2 hits, no faster than the generic kernel).
//     The wait is bounded (~1 ms, then the workgroup stops taking part): a launch that shares the chip
//     with another kernel loses the pacing, never the result, and cannot hang.
// Sums are formed in plan order: results are reproducible run to run.
// Bound: every edge moves 256 bytes L2 -> CU: 64 B/clk/CU = ~34 TB/s on the chip.
#include "common.h"

using sgp::f32x4;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kRowsCap = 511;                 // rows per workgroup (LDS: 512 rows x 256 B, the last one takes the padding)
constexpr int kU = 4;                         // rounds (of 64 edges) per loop iteration; segments are padded to it
constexpr int kWaves = 16;

struct CbArgs {
    const uint2* plan;                        // {col | row_local << 22 | last << 31, weight bits}
    const int* segptr;                        // [n_wg * nb + 1], in rounds of 64 entries
    const int* wg_row0;                       // [n_wg + 1]
    const float* x; long long xrs, xbs;
    const float* xh; long long xhrs, xhbs;    // halo source: columns >= n_own (local block of a node partition)
    int n_own;
    float* y; long long yrs, ybs;
    int nb, batch;
    unsigned* pace;                           // arrival counter of the per-step rendezvous, or null
    unsigned n_arrive;                        // workgroups of the launch
    const int* pred; int pred_want;           // launch predicate (common.h)
};

template <bool HALO>
__global__ __launch_bounds__(kWaves * 64) void spmm_colblock(CbArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (a.pred != nullptr && a.pred[0] != a.pred_want) return;
    const int wg = blockIdx.x;
    const int f_base = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = a.wg_row0[wg];
    const int nrows = a.wg_row0[wg + 1] - row0;

    f32x4* rows = reinterpret_cast<f32x4*>(lds);          // [512][16] float4
    for (int i = tid; i < 512 * 16; i += kWaves * 64) rows[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // one stream of rounds per time step (the blocks follow each other in it), software-pipelined two
    // deep: plan entries of iteration i + 2 and source rows of iteration i + 1 are in flight while
    // iteration i is summed.  Reads past the end of the stream fetch the next workgroup's entries (the
    // plan ends with 2 kU spare rounds) and their rows for nothing.
    const int s_begin = __builtin_amdgcn_readfirstlane(a.segptr[(long long)wg * a.nb]);
    const int s_end = __builtin_amdgcn_readfirstlane(a.segptr[(long long)wg * a.nb + a.nb]);
    const int slot = wave * 4 + g;                        // this lane group's list inside a round
    const long long xrs_b = a.xrs * 4;
    const long long xhrs_b = a.xhrs * 4;
    const uint2* pl = a.plan + slot;
    bool paced = a.pace != nullptr;
    for (int t = 0; t < a.batch; ++t) {
        const char* xt = reinterpret_cast<const char*>(a.x + (long long)t * a.xbs + f_base) + li * 16;
        const char* xht = HALO ? reinterpret_cast<const char*>(a.xh + (long long)t * a.xhbs + f_base) + li * 16 : nullptr;
        // source row of a plan entry: own rows from x, columns >= n_own from the halo rows the partition received
        auto src = [&](unsigned e) -> const f32x4* {
            const int c = (int)(e & 0x3fffffu);
            if (HALO && c >= a.n_own) return reinterpret_cast<const f32x4*>(xht + (long long)(c - a.n_own) * xhrs_b);
            return reinterpret_cast<const f32x4*>(xt + (long long)c * xrs_b);
        };
        uint2 e0[kU], e1[kU], e2[kU];
        f32x4 x0[kU], x1[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) e0[u] = pl[(long long)(s_begin + u) * 64];
#pragma unroll
        for (int u = 0; u < kU; ++u) e1[u] = pl[(long long)(s_begin + kU + u) * 64];
#pragma unroll
        for (int u = 0; u < kU; ++u) x0[u] = *src(e0[u].x);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = s_begin; r < s_end; r += kU) {
#pragma unroll
            for (int u = 0; u < kU; ++u) {                  // (streamed: the plan must not displace the block from the L2)
                const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(pl + (long long)(r + 2 * kU + u) * 64));
                e2[u] = make_uint2(v.x, v.y);
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) x1[u] = *src(e1[u].x);
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                acc += __uint_as_float(e0[u].y) * x0[u];
                if (e0[u].x >> 31) {                      // last edge of this row's run: sum -> the row
                    f32x4* p = rows + ((e0[u].x >> 22) & 511u) * 16u + li;
                    *p += acc;
                    acc = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) { e0[u] = e1[u]; e1[u] = e2[u]; x0[u] = x1[u]; }
        }
        __syncthreads();
        // rows out, sums cleared
        float* yt = a.y + (long long)t * a.ybs + f_base;
        for (int i = tid; i < nrows * 16; i += kWaves * 64) {
            const f32x4 v = rows[i];
            rows[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(yt + (long long)(row0 + (i >> 4)) * a.yrs + (i & 15) * 4));
        }
        if (paced && tid == 0 && t + 1 < a.batch) {
            __hip_atomic_fetch_add(a.pace, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(t + 1) * a.n_arrive;
            int spins = 0;
            while ((int)(__hip_atomic_load(a.pace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > 2048) { paced = false; break; }
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int32_t sgp_spmm_colblock_rows_cap(void) { return kRowsCap; }
int32_t sgp_spmm_colblock_round_pad(void) { return kU; }

int sgp_spmm_colblock_f32(const int32_t* plan, const int32_t* segptr, const int32_t* wg_row0,
                          int32_t n_wg, int32_t n_blocks,
                          const float* X, int64_t xrs, int64_t xbs,
                          const float* X_halo, int64_t xhrs, int64_t xhbs, int32_t n_own,
                          float* Y, int64_t yrs, int64_t ybs,
                          int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                          const int32_t* pred, int32_t run_if, sgp_stream_t stream) {
    const sgp::Predicate pr{pred, run_if};
    SGP_REQUIRE(plan && segptr && wg_row0 && X && Y, "sgp_spmm_colblock_f32: null pointer");
    SGP_REQUIRE(n_wg >= 0 && n_blocks >= 0 && n_rows >= 0 && n_cols >= 0 && batch >= 0 && feat >= 0,
                "sgp_spmm_colblock_f32: bad size");
    SGP_REQUIRE(n_cols < (1 << 22), "sgp_spmm_colblock_f32: more than 2^22 columns (use sgp_spmm_csr_f32)");
    if (n_rows == 0 || batch == 0 || feat == 0 || n_wg == 0) return 0;
    if (feat % 64 != 0)
        return sgp::fail(SGP_EUNSUP, "sgp_spmm_colblock_f32: feat=%d is not a multiple of 64", feat);
    SGP_REQUIRE(xrs % 4 == 0 && xbs % 4 == 0 && yrs % 4 == 0 && ybs % 4 == 0 && sgp::aligned16(X) && sgp::aligned16(Y),
                "sgp_spmm_colblock_f32: strides/pointers must be 16-byte aligned");
    SGP_REQUIRE((long long)n_cols * xrs < (1ll << 40) && feat / 64 <= 65535, "sgp_spmm_colblock_f32: operand too large");
    SGP_REQUIRE(!X_halo || (n_own >= 0 && n_own <= n_cols && xhrs % 4 == 0 && xhbs % 4 == 0 && sgp::aligned16(X_halo) &&
                            (long long)(n_cols - n_own) * xhrs < (1ll << 40)),
                "sgp_spmm_colblock_f32: halo rows must be 16-byte aligned, n_own within the columns");
    CbArgs a;
    a.plan = reinterpret_cast<const uint2*>(plan); a.segptr = segptr; a.wg_row0 = wg_row0;
    a.x = X; a.xrs = xrs; a.xbs = xbs; a.y = Y; a.yrs = yrs; a.ybs = ybs;
    a.xh = X_halo; a.xhrs = xhrs; a.xhbs = xhbs; a.n_own = X_halo ? n_own : 0x7fffffff;
    a.nb = n_blocks; a.batch = batch;
    a.pred = pr.flag; a.pred_want = pr.want;
    hipStream_t s = (hipStream_t)stream;
    // per-step rendezvous only when all workgroups are resident at once (one per CU: 128 KB of LDS each)
    a.pace = nullptr; a.n_arrive = (unsigned)n_wg * (unsigned)(feat / 64);
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            (long long)a.n_arrive <= cus && batch > 1)
            a.pace = sgp::sync_slot(s);              // this launch's own counter (two launches in flight never share one)
    }
    const size_t lds_bytes = 512 * 256;
    auto kern = X_halo ? spmm_colblock<true> : spmm_colblock<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_colblock: LDS opt-in: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg, (unsigned)(feat / 64)), dim3(kWaves * 64), lds_bytes, s, a);
    return sgp::check_launch("spmm_colblock");
}

}  // extern "C"
