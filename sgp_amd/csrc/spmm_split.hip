// Split-fp16 hop: y[b] = A x[b] on the 16-bit matrix cores with fp32-equivalent results
// (reference call site: lib/sgp_preprocessing.py:200-203, `x = adj @ x` per hop).  gfx950 / wave64 only.
//
// Why another hop kernel.  The exact-fp32 row-group kernels (spmm_res / spmm_mix) are bound by
// the fp32 matrix rate (1 / 16 of the 16-bit rate): compute alone caps them at 0.34 of the HBM roofline, and
// their 64-row tiles pull 5.8 staged rows per result row through the CU.  Here a value is carried as TWO fp16
// pieces of its scaled self, v * s = hi + lo (22 significant bits; s a power of two that puts the data in the
// middle of the fp16 range), and a product as hi*hi + hi*lo + lo*hi accumulated in fp32 by
// v_mfma_f32_16x16x32_f16 -- measured against fp64 its error is BELOW that of an fp32 fma chain
// (tools/ubench/f16split_test.hip: 0.9e-7 vs 1.6e-7 of the input scale over 128-term rows).  Dense 16 x 32
// blocks of A at 16x the fp32 rate make 256-row tiles affordable: 3.1 staged rows per result row.
//
// Structure (plan: sgp_amd/splitplan.py):
//   * a workgroup of 8 waves owns a tile of up to 8 x 32 rows for a chunk of time steps; wave w owns up to 32 rows
//     (two 16-row halves) and NCH chunks of 32 columns -- its A fragments (hi / lo piece, both halves, 16 VGPRs per
//     chunk) are loaded once and stay in registers for the whole chunk;
//   * a unit = (time step, 16-feature slice).  The tile's distinct source rows (<= SMAX) arrive as 64-byte pieces by
//     LDS-DMA (global_load_lds_dwordx4, 4 lanes per row, optional second "halo" source) in one of THREE LDS buffers,
//     two units ahead of the multiply; the wave that requested a piece scales and splits it IN PLACE (v_fma_mixlo /
//     mixhi_f16) into the operand layout: 8 staged rows = 512 B, hi pieces of row r at 32 r, lo pieces at 256 + 32 r;
//   * B operands come straight out of those fp16 rows with ds_read_b64_tr_b16 (per-lane ROW addresses: the 4 rows a
//     16-lane group reads may lie anywhere), 4 reads + 6 MFMAs per chunk, the reads issued from asm two chunks ahead
//     with counted lgkmcnt waits;
//   * results: 4 x 4 transpose inside every quad of lanes (DPP) -> 16-byte row pieces; an even slice waits for its odd
//     neighbour so that whole 128-byte lines leave together; stores are issued behind the conversion, the next unit's
//     staging requests behind them (stores and loads share vmcnt: a wait for "all but the newest nld" then covers the
//     pieces it is meant for whatever the stores do);
//   * one barrier per unit.
// Limits checked by the planner: a wave's rows touch <= 32 NCH distinct columns, a tile <= SMAX; feat % 16 == 0;
// |x| * x_scale and |a| * w_scale must stay below 65504 (the host picks the scales from bounds).
#include "common.h"
#include <stdlib.h>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
using sgp::f32x4;

#ifndef SGP_SPLIT_NW
#define SGP_SPLIT_NW 16
#endif
constexpr int NW = SGP_SPLIT_NW;             // waves per workgroup: 16 x 16 rows (default), 12 x 16, or 8 x 32 (two 16-row halves)
constexpr int NH = NW == 8 ? 2 : 1;          // 16-row halves per wave
static_assert(NW == 8 || NW == 12 || NW == 16, "8 waves x 32 rows, or 12 / 16 waves x 16 rows");
#ifndef SGP_SPLIT_NCH
#define SGP_SPLIT_NCH (SGP_SPLIT_NW == 8 ? 9 : 8)
#endif
constexpr int NCH = SGP_SPLIT_NCH;           // resident 32-column chunks per wave (experiment builds: -DSGP_SPLIT_NCH=..)
#ifndef SGP_SPLIT_SMAX
#define SGP_SPLIT_SMAX 768
#endif
constexpr int SMAX = SGP_SPLIT_SMAX;         // staged rows per tile (3 x 64 x SMAX bytes of LDS: at most 832)
constexpr int NLD = (SMAX + 16 * NW - 1) / (16 * NW);   // LDS-DMA instructions per wave and unit (16 rows each)
constexpr int BUF = SMAX * 64;               // one staged unit: 64 B per row (fp32 in flight, then hi | lo fp16)
constexpr int NBUF = 3;                      // landing | being converted | being multiplied
constexpr int HDR = 64;                      // ints per tile header: [NW : 2 NW] rows of every wave, [2 NW] staged rows U

// ablation / timeline switches (SGP_TUNE=split_abl=..) exist only in builds with -DSGP_ABLATION (tools/build_variant.sh):
// the product kernel carries none of their tests
#ifdef SGP_ABLATION
#define ABL(bit) ((a.mode & (bit)) != 0)
#else
#define ABL(bit) false
#endif

struct SplitArgs {
    const int* hdr; const int* rowid; const int* ucol; const h8* afr; const int* adr;
    int n_tiles, tiles_per_xcd;
    const float* X; long long xrs, xbs;
    const float* XH; long long xhrs, xhbs;   // halo source (columns >= n_own): local block of a node partition
    int n_own;
    float* Y; long long yrs, ybs;
    int batch, nslice, t_chunk;
    float x_scale, inv_scale;
    unsigned long long* dbg;                 // mode 256: per-wave s_memtime stamps of workgroup 0
    int mode;                                // ablations (SGP_TUNE=split_abl=..): 1 no loads, 2 no MFMAs, 4 no stores, 8 no conversion, 16 all tiles of an XCD stage the same rows (upper bound of L2 sharing), 32 unpaired stores
};

// B operand of one chunk: four transpose reads (hi / lo piece x rows k = 0..3 / 4..7 of every lane group).  Issued
// from asm so that the wait in front of the MFMAs can be COUNTED (LDS returns in order): hipcc waits lgkmcnt(0),
// i.e. also for the reads of the next chunk that were just issued.
struct BOp { s4v h0, h1, l0, l1; };
__device__ __forceinline__ void tr_issue(BOp& b, unsigned a0, unsigned a1) {
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:256\n\tds_read_b64_tr_b16 %3, %5 offset:256"
                 : "=&v"(b.h0), "=&v"(b.h1), "=&v"(b.l0), "=&v"(b.l1) : "v"(a0), "v"(a1) : "memory");
}
template <int N> __device__ __forceinline__ void tr_wait(BOp& b) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b.h0), "+v"(b.h1), "+v"(b.l0), "+v"(b.l1) : "n"(N));
}
__device__ __forceinline__ h8 cat8(s4v x, s4v y) {
    return __builtin_bit_cast(h8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
}

// 64 lanes x 16 B from per-lane global addresses (sbase + voff) straight into LDS at lds_off + 16 * lane
__device__ __forceinline__ void dma16(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_off) : "memory");   // m0 is a reserved register: hipcc rejects it in a clobber list and never keeps a value in it across an asm
}

// the same with a full per-lane address (two sources: own rows and halo rows)
__device__ __forceinline__ void dma16_vaddr(const void* vaddr, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(vaddr), "s"(lds_off) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_n(int n) {       // n is wave-uniform, 0 .. NLD
    switch (n) {
        case 0: wait_vm<0>(); break; case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break;
        case 3: wait_vm<3>(); break; case 4: wait_vm<4>(); break; case 5: wait_vm<5>(); break;
        case 6: wait_vm<6>(); break; default: wait_vm<7>(); break;
    }
}
static_assert(NCH >= 2, "the operand ring is primed with two chunks");
static_assert(NLD <= NCH, "one staging piece per chunk of the MFMA phase");
static_assert(NLD <= 7 && NBUF * BUF <= 160 * 1024, "wait_vm_n covers 0 .. 7 outstanding pieces; three buffers in 160 KB");

// v * s = hi + lo in 8 instructions per 4 values: hi = fp16(v * s), lo = fp16(v * s - hi) as ONE fused operation each
// (v_fma_mixlo / mixhi_f16: fp32 fma of (fp32 v, fp32 s, fp16 half of a register), rounded once to fp16 into the low
// / high half of the destination) -- the remainder of an 11-bit rounding of a 24-bit value is exact in the fma.
// (hipcc's own sequence for the same split is 12: pk_mul, cvt_pkrtz, 2 cvt_f32_f16, pk_add, cvt_pk per pair.)
__device__ __forceinline__ void split4(const f32x4 v, const float s, uint2& hi, uint2& lo) {
    unsigned h01 = 0, h23 = 0, l01 = 0, l23 = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(v[0]), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(v[1]), "s"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(v[2]), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(v[3]), "s"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(v[0]), "s"(s), "v"(h01));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(v[1]), "s"(s), "v"(h01));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(v[2]), "s"(s), "v"(h23));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(v[3]), "s"(s), "v"(h23));
    hi.x = h01; hi.y = h23; lo.x = l01; lo.y = l23;
}

template <bool HALO>
__global__ __launch_bounds__(NW * 64, NW / 4) void spmm_split(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];

    // XCD x (= blockIdx % 8) walks its own contiguous range of tiles, time chunk by time chunk, so the 32
    // workgroups an XCD runs side by side are neighbouring tiles of the same steps (their staged rows overlap)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = xcd * a.tiles_per_xcd + j % a.tiles_per_xcd;
    const int tchunk = j / a.tiles_per_xcd;
    if (tile >= a.n_tiles) return;
    const int t_begin = tchunk * a.t_chunk;
    const int t_end = min(a.batch, t_begin + a.t_chunk);
    if (t_begin >= t_end) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int src_tile = ABL(16) ? xcd * a.tiles_per_xcd : tile;   // 16: every tile of an XCD stages the same rows
    const int nU = __builtin_amdgcn_readfirstlane(a.hdr[(size_t)src_tile * HDR + 2 * NW]);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // ---- resident plan: A fragments and per-lane row addresses of the transpose reads
    h8 af[NCH][2 * NH];
    int ad[NCH][2];
    {
        const h8* ap = a.afr + ((size_t)(tile * NW + wave) * NCH * 2 * NH) * 64 + lane;
        const int* dp = a.adr + ((size_t)(tile * NW + wave) * NCH * 2) * 64 + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int q = 0; q < 2 * NH; ++q) af[c][q] = ap[(c * 2 * NH + q) * 64];
            ad[c][0] = dp[(c * 2 + 0) * 64];
            ad[c][1] = dp[(c * 2 + 1) * 64];
        }
    }
    // ---- pieces this wave stages: instruction i covers staged rows (i NW + wave) 16 .. + 15, 4 lanes per row;
    // lanes past the tile's last row re-load row 0 (a valid address; their LDS rows are never read)
    unsigned xoff[NLD];
    const int* uc = a.ucol + (size_t)src_tile * SMAX;
    int nld = 0;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int s = (i * NW + wave) * 16 + (lane >> 2);
        const int c = uc[s < nU ? s : 0];
        if (HALO && c >= a.n_own)                              // bit 31 marks a halo row (offsets stay below 2^31)
            xoff[i] = 0x80000000u | (unsigned)((c - a.n_own) * a.xhrs * 4 + (lane & 3) * 16);
        else
            xoff[i] = (unsigned)(c * a.xrs * 4 + (lane & 3) * 16);
        if ((i * NW + wave) * 16 < nU) nld = i + 1;
    }
    wait_vm<0>();                                             // the plan loads above: from here on vmcnt is counted by hand
    // own rows inside a buffer: row s = (i NW + wave) 16 + lane / 4 at s * 64, this lane's 16 B at + (lane & 3) * 16
    const int own = wave * 1024 + lane * 16;
    // converted layout, in place: a group of 8 staged rows (512 B) keeps the hi pieces of row r at 32 r and the lo
    // pieces at 256 + 32 r -- eight consecutive rows cover all 64 banks with either piece, and lo = hi + 256 is an
    // immediate offset of the transpose reads.  A wave instruction covers two whole groups, so every read of a
    // group has returned before its first write goes out.
    const int cv_off = wave * 1024 + (lane >> 5) * 512 + ((lane >> 2) & 7) * 32 + (lane & 3) * 8;

    auto piece = [&](unsigned off, const float* xb, const float* xh, unsigned lds_off) {
        if constexpr (HALO) {
            const char* b = (off & 0x80000000u) ? (const char*)xh : (const char*)xb;
            dma16_vaddr(b + (off & 0x7fffffffu), lds_off);
        } else {
            dma16(off, xb, lds_off);
        }
    };
    auto issue_dma = [&](int t, int sl, int buf) {
        const float* xb = ABL(64) ? a.X : a.X + (long long)t * a.xbs + sl * 16;   // 64: always step 0 (all L2 hits)
        const float* xh = HALO ? a.XH + (long long)t * a.xhbs + sl * 16 : nullptr;
        const unsigned base = lds0 + buf * BUF + wave * 1024;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (i < nld) piece(xoff[i], xb, xh, base + i * (NW * 1024));
    };
    auto convert = [&](int buf) {
        char* rb = lds + buf * BUF;
        // all NLD pieces, also those this wave did not request (rows past the tile's last: stale bytes that no
        // transpose read addresses): no branches, so the six reads go out together
        f32x4 v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) v[i] = *(const f32x4*)(rb + own + i * (NW * 1024));
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            uint2 hi, lo;
            split4(v[i], a.x_scale, hi, lo);
            *(uint2*)(rb + cv_off + i * (NW * 1024)) = hi;
            *(uint2*)(rb + cv_off + i * (NW * 1024) + 256) = lo;
        }
    };
    // The products are formed TRANSPOSED -- the staged rows are the MFMA's A operand (M = the slice's 16 features), the
    // plan's fragments its B operand (N = the wave's 16 rows; both operands have the same lane layout) -- so that the
    // accumulator of lane (n, q) is features 4 q .. 4 q + 3 of row n: one 16-byte piece of a result row, stored as it is
    // (before: a 4 x 4 transpose inside every quad of lanes, ~20 VALU instructions per unit and half).
    // This lane stores row slots my_slot (half 0) and 16 + my_slot (half 1); -1 = empty slot
    const int my_slot = lane & 15;
    const int* rid = a.rowid + (size_t)(tile * NW + wave) * (16 * NH);
    const int row_a = rid[my_slot], row_b = NH == 2 ? rid[(NH - 1) * 16 + my_slot] : -1;
    const long long yoff_a = (long long)row_a * a.yrs + 4 * (lane >> 4);
    const long long yoff_b = (long long)row_b * a.yrs + 4 * (lane >> 4);

    const int n_units = (t_end - t_begin) * a.nslice;
    // DMA cursor (two units ahead of the multiply) and multiply cursor
    int dt = t_begin, dsl = 0;
    auto advance = [&](int& t, int& sl) { if (++sl == a.nslice) { sl = 0; ++t; } };

    issue_dma(dt, dsl, 0); advance(dt, dsl);
    if (n_units > 1) { issue_dma(dt, dsl, 1); advance(dt, dsl); wait_vm_n(nld); } else wait_vm<0>();
    convert(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    int t = t_begin, sl = 0, cur = 0;
    // measured (T = 512, target graph): all eight waves in the same order 17.6 ms per hop, the two waves of a SIMD in
    // opposite order (one multiplies while the other converts) 18.3 -- mode 128 selects the opposite order
    // every wave multiplies first, then converts (three same-lease A/B runs: 17.7-18.1 ms per hop against 18.6-19.0 with
    // the two waves of a SIMD in opposite order -- a conversion beside the partner's MFMAs takes twice as long);
    // mode 128 selects the opposite order
    const bool late = wave >= NW / 2 && ABL(128);
    f32x4 h0 = {0, 0, 0, 0}, h1 = {0, 0, 0, 0};
    auto stamp = [&](int u, int k) {
        if (ABL(256) && blockIdx.x == 0 && lane == 0 && u >= 16 && u < 24)
            a.dbg[((u - 16) * NW + wave) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    for (int u = 0; u < n_units; ++u) {
        stamp(u, 0);
        const int nxt = cur == NBUF - 1 ? 0 : cur + 1;
        const int nn = nxt == NBUF - 1 ? 0 : nxt + 1;
        const bool more2 = u + 2 < n_units;
        // the six pieces of unit u + 2 are requested one per chunk INSIDE the MFMA phase: a piece whose issue stalls on
        // a full memory queue then waits under matrix-core work that is already queued, not in front of it
        const bool dma_now = more2 && !ABL(1);
        const float* xb2 = ABL(64) ? a.X : a.X + (long long)dt * a.xbs + dsl * 16;
        const float* xh2 = HALO ? a.XH + (long long)dt * a.xhbs + dsl * 16 : nullptr;
        const unsigned base2 = lds0 + nn * BUF + wave * 1024;
        if (dma_now) advance(dt, dsl);

        stamp(u, 1);
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        auto stage_next = [&]() {
            if (u + 1 < n_units) {
                // the pieces of unit u + 1 were requested a whole unit ago; only the nld newest (unit u + 2, requested
                // earlier in this unit by either kind of wave) may stay in flight
                if (dma_now) wait_vm_n(nld); else wait_vm<0>();
                if (!ABL(8)) convert(nxt);
            }
        };
        // The two waves of a SIMD (w and w + 4) take the unit's two phases in opposite order, so one multiplies while
        // the other converts (both orders are legal: unit u was converted before the last barrier, unit u + 1 landed a
        // unit ago).  A late wave requests its pieces of unit u + 2 in front of its conversion, an early wave one per
        // chunk inside its MFMA phase: either way they have a whole unit to land.
        if (late) {
            if (dma_now) {
#pragma unroll
                for (int i = 0; i < NLD; ++i)
                    if (i < nld) piece(xoff[i], xb2, xh2, base2 + i * (NW * 1024));
            }
            stage_next();
        }
        stamp(u, 2);
        if (!ABL(2)) {
            // operands of chunk c + 1 are requested before the MFMAs of chunk c issue
            const unsigned cbo = lds0 + cur * BUF;
            BOp b[3];                                         // two chunks in flight beside the one being multiplied
            tr_issue(b[0], cbo + ad[0][0], cbo + ad[0][1]);
            tr_issue(b[1], cbo + ad[1][0], cbo + ad[1][1]);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                BOp& x = b[c % 3];
                if (c + 2 < NCH) { tr_issue(b[(c + 2) % 3], cbo + ad[c + 2][0], cbo + ad[c + 2][1]); tr_wait<8>(x); }
                else if (c + 1 < NCH) tr_wait<4>(x);
                else tr_wait<0>(x);
                const h8 bh = cat8(x.h0, x.h1), bl = cat8(x.l0, x.l1);
                if constexpr (NH == 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, af[c][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, af[c][2 * (NH - 1)], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, af[c][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, af[c][2 * (NH - 1)], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, af[c][1], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, af[c][2 * (NH - 1) + 1], acc1, 0, 0, 0);
                } else {
                    // one half per wave: the cross terms go to a second accumulator so that consecutive MFMAs do
                    // not wait for each other's result (four waves per SIMD fill the rest)
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, af[c][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, af[c][0], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, af[c][1], acc1, 0, 0, 0);
                }
                if (c < NLD && dma_now && !late && c < nld) piece(xoff[c], xb2, xh2, base2 + c * (NW * 1024));
            }
            if constexpr (NH == 1) acc0 += acc1;
        } else if (dma_now && !late) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                if (i < nld) piece(xoff[i], xb2, xh2, base2 + i * (NW * 1024));
        }
        stamp(u, 3);
        if (!late) stage_next();
        stamp(u, 4);
        if (!ABL(4)) {
            f32x4 r0 = acc0 * a.inv_scale, r1 = acc1 * a.inv_scale;
            // an even slice waits for its odd neighbour: the two 64-byte halves of a 128-byte line leave together
            if (!(sl & 1) && sl + 1 < a.nslice && !ABL(32)) {
                h0 = r0; h1 = r1;
            } else {
                float* yb = a.Y + (long long)t * a.ybs + sl * 16;
                if ((sl & 1) && !ABL(32)) {
                    if (row_a >= 0) { *(f32x4*)(yb + yoff_a - 16) = h0; *(f32x4*)(yb + yoff_a) = r0; }
                    if (NH == 2 && row_b >= 0) { *(f32x4*)(yb + yoff_b - 16) = h1; *(f32x4*)(yb + yoff_b) = r1; }
                } else {
                    if (row_a >= 0) *(f32x4*)(yb + yoff_a) = r0;
                    if (NH == 2 && row_b >= 0) *(f32x4*)(yb + yoff_b) = r1;
                }
            }
        }
        stamp(u, 5);
        advance(t, sl);
        cur = nxt;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(u, 6);
    }
}

}  // namespace

extern "C" int32_t sgp_spmm_split_chunks(void) { return NCH; }
extern "C" int32_t sgp_spmm_split_max_union(void) { return SMAX; }
extern "C" int32_t sgp_spmm_split_waves(void) { return NW; }
extern "C" int32_t sgp_spmm_split_rows_per_wave(void) { return 16 * NH; }

extern "C" int sgp_spmm_split_f32(const int32_t* hdr, const int32_t* rowid, const int32_t* ucol, const void* afr,
                                  const int32_t* adr,
                                  int32_t n_tiles,
                                  const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                                  const float* X_halo, int64_t xh_row_stride, int64_t xh_batch_stride, int32_t n_own,
                                  float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                                  int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                                  float x_scale, float w_scale, int32_t t_chunk, sgp_stream_t stream) {
    SGP_REQUIRE(n_tiles >= 0 && batch >= 0 && n_rows >= 0 && n_cols >= 0, "spmm_split: negative size");
    if (n_tiles == 0 || batch == 0 || n_rows == 0) return 0;
    SGP_REQUIRE(hdr && rowid && ucol && afr && adr && X && Y, "spmm_split: null pointer");
    SGP_REQUIRE(feat > 0 && feat % 16 == 0, "spmm_split: feat = %d is not a multiple of 16", feat);
    SGP_REQUIRE(sgp::aligned16(X) && x_row_stride % 4 == 0 && x_batch_stride % 4 == 0 &&
                sgp::aligned16(Y) && y_row_stride % 4 == 0 && y_batch_stride % 4 == 0,
                "spmm_split: X and Y rows must be 16-byte aligned");
    {
        const long long own = X_halo ? n_own : n_cols, far = X_halo ? n_cols - n_own : 0;
        SGP_REQUIRE(own >= 0 && far >= 0 && own * x_row_stride < (1ll << 29) && far * xh_row_stride < (1ll << 29) &&
                    (long long)n_rows * y_row_stride < (1ll << 40), "spmm_split: source rows beyond 31-bit byte offsets");
        SGP_REQUIRE(!X_halo || (sgp::aligned16(X_halo) && xh_row_stride % 4 == 0 && xh_batch_stride % 4 == 0),
                    "spmm_split: halo rows must be 16-byte aligned");
    }
    SGP_REQUIRE(x_scale > 0.f && w_scale > 0.f, "spmm_split: scales must be positive");
    SplitArgs a;
    a.hdr = hdr; a.rowid = rowid; a.ucol = ucol; a.afr = (const h8*)afr; a.adr = adr;
    a.n_tiles = n_tiles; a.tiles_per_xcd = (n_tiles + 7) / 8;
    a.X = X; a.xrs = x_row_stride; a.xbs = x_batch_stride;
    a.XH = X_halo; a.xhrs = xh_row_stride; a.xhbs = xh_batch_stride; a.n_own = X_halo ? n_own : 0x7fffffff;
    a.Y = Y; a.yrs = y_row_stride; a.ybs = y_batch_stride;
    a.batch = batch; a.nslice = feat / 16;
    if (t_chunk <= 0) {
        // time steps per workgroup: long chunks amortise the plan load (A fragments: ~1.6 units' worth of staging per
        // workgroup), short ones fill the last round of the chip.  Cost model: rounds taken / rounds of work x
        // (1 + 1.6 / steps); ties go to the longer chunk.
        int cus = 256, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
        double best = 1e30;
        for (int tc = 64; tc >= 8; tc >>= 1) {
            if (tc > batch && tc > 8) continue;
            const double w = (double)n_tiles * ((batch + tc - 1) / tc) / cus;
            const double cost = (w <= 1.0 ? 1.0 / w : (double)(long long)(w + 0.999999) / w) * (1.0 + 1.6 / tc);
            if (cost < best - 1e-9) { best = cost; t_chunk = tc; }
        }
        if (t_chunk <= 0) t_chunk = 8;
    }
    a.t_chunk = t_chunk;
    a.x_scale = x_scale; a.inv_scale = 1.f / (x_scale * w_scale);
#ifdef SGP_ABLATION
    static const int abl = (int)sgp::tune("split_abl", 0);
#else
    constexpr int abl = 0;
#endif
    a.mode = abl;
    a.dbg = nullptr;
    if (abl & 256) { if (hipMalloc(&a.dbg, 8 * NW * 8 * 8) != hipSuccess) return sgp::fail(SGP_EINVAL, "dbg alloc"); (void)hipMemset(a.dbg, 0, 8 * NW * 8 * 8); }
    const int n_tchunks = (batch + t_chunk - 1) / t_chunk;
    auto kern = X_halo ? spmm_split<true> : spmm_split<false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * BUF);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_split: LDS attribute: %s", hipGetErrorString(e));
    const unsigned grid = 8u * (unsigned)a.tiles_per_xcd * (unsigned)n_tchunks;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), NBUF * BUF, (hipStream_t)stream, a);
    if (abl & 256) {
        unsigned long long h[8 * NW * 8];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(a.dbg);
        printf("spmm_split timeline (cycles since the unit's top; columns: top | first phase start | MFMAs start | MFMAs done | staged | stores | barrier)\n");
        for (int u = 0; u < 8; ++u) for (int w = 0; w < NW; ++w) {
            const unsigned long long* r = h + (u * NW + w) * 8;
            printf("  unit %d wave %d: top %llu |", u, w, r[0] - h[0]);
            for (int k = 1; k < 7; ++k) printf(" %6lld", (long long)(r[k] - r[0]));
            printf("\n");
        }
    }
    return sgp::check_launch("spmm_split");
}
