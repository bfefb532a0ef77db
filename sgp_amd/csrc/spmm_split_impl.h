// Split-fp16 hop: y[b] = A x[b] on the 16-bit matrix cores with fp32-equivalent results
// (reference call site: lib/sgp_preprocessing.py:200-203, `x = adj @ x` per hop).  gfx950 / wave64 only.
//
// Why another hop kernel.  The exact-fp32 row-group kernels (spmm_res / spmm_mix) are bound by
// the fp32 matrix rate (1 / 16 of the 16-bit rate): compute alone caps them at 0.34 of the HBM roofline, and
// their 64-row tiles pull 5.8 staged rows per result row through the CU.  Here a value is carried as TWO fp16
// pieces of its scaled self, v * s = hi + lo (22 significant bits; s a power of two that puts the data in the
// upper part of the fp16 range), and a product as hi*hi + hi*lo + lo*hi accumulated in fp32 by
// v_mfma_f32_16x16x32_f16.  Dense 16 x 32 blocks of A at 16x the fp32 rate make 256-row tiles affordable:
// 3.2 staged rows per result row.
//
// Scales and error model.  x is scaled PER FEATURE COLUMN (table x_tab[2][feat]: scale, then its inverse; made by
// sgp_split_prepare_f32 from a bound on every column), A PER ROW (plan: fragments hold a[i, :] * 2^e_i, rinv the
// inverse), so the result is invariant to rescaling a column of x or a row of A, as fp32 is.  With a column's
// bound at 2^13 .. 2^14 a value keeps 22 bits down to 2^-16 of the bound; below that the low piece is an fp16
// subnormal and the error becomes ABSOLUTE, <= 2^-38 x the column's bound -- which is why the default dispatch
// (sgp_split_prepare_f32) admits this kernel only where that is below 2^-22 of the column's RMS, and sends
// everything else to the exact-fp32 kernels (launch predicate: the entry's `pred, run_if` pair).
//
// Structure (plan: sgp_amd/splitplan.py):
//   * a workgroup of NW = 16 waves owns a tile of up to 16 x 16 rows for a chunk of time steps; wave w owns up to 16
//     rows and NCH chunks of 32 columns -- its A fragments (hi / lo piece, 8 VGPRs per chunk) are loaded once and
//     stay in registers for the whole time chunk;
//   * a unit = (time step, 16-feature slice).  The tile's distinct source rows (<= SMAX) arrive as 64-byte pieces by
//     LDS-DMA (global_load_lds_dwordx4, 4 lanes per row, optional second "halo" source) in one of THREE LDS buffers,
//     two units ahead of the multiply; the wave that requested a piece scales and splits it IN PLACE (v_fma_mixlo /
//     mixhi_f16) into the operand layout: 8 staged rows = 512 B, hi pieces of row r at 32 r, lo pieces at 256 + 32 r;
//   * the products are formed TRANSPOSED (staged rows = the MFMA's A operand, M = the slice's 16 features; the plan's
//     fragments = its B operand, N = the wave's 16 rows), so a lane's accumulator is a 16-byte piece of a result row;
//     B^T operands come straight out of the fp16 rows with ds_read_b64_tr_b16 (per-lane ROW addresses);
//   * ONE software-pipelined phase per unit (round 5; before: multiply, then convert, then store, one after the
//     other -- the round-5 ablation table showed the parts of a unit adding up instead of overlapping): inside the
//     chunk loop of unit u a wave also requests its pieces of unit u + 2 (one per chunk), reads its landed pieces of
//     unit u + 1, splits them between the MFMAs of the later chunks and writes them back, all LDS traffic issued
//     from asm with COUNTED lgkmcnt waits (LDS returns in order); the result rows leave right behind the last
//     chunk; an even slice waits for its odd neighbour so that whole 128-byte lines leave together;
//   * one barrier per unit.
// Limits checked by the planner: a wave's rows touch <= 32 NCH distinct columns, a tile <= SMAX; feat % 16 == 0.
// THIS FILE IS AN IMPLEMENTATION HEADER: spmm_split.hip instantiates it as the standard form (16 waves x 7 chunks: one
// wave per 16 rows x 224 columns, four waves per SIMD), spmm_split_wide.hip as the WIDE form (8 waves x 14 chunks: 448
// columns per wave, two waves per SIMD -- operators with long rows, whose passes it halves).  The including file defines
// SGP_SPLIT_NW / _NCH / _SMAX / _CR and SGP_SPLIT_NAME(x) (the exported symbols' names).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
using sgp::f32x4;

#ifndef SGP_SPLIT_NW
#define SGP_SPLIT_NW 16
#endif
constexpr int NW = SGP_SPLIT_NW;             // waves per workgroup, 16 rows each
static_assert(NW == 8 || NW == 12 || NW == 16, "8, 12 or 16 waves x 16 rows");
#ifndef SGP_SPLIT_NCH
#define SGP_SPLIT_NCH 7
#endif
constexpr int NCH = SGP_SPLIT_NCH;           // resident 32-column chunks per wave (experiment builds: -DSGP_SPLIT_NCH=..)
#ifndef SGP_SPLIT_SMAX
#define SGP_SPLIT_SMAX 768
#endif
constexpr int SMAX = SGP_SPLIT_SMAX;         // staged rows per tile (3 x 64 x SMAX bytes of LDS)
constexpr int NLD = (SMAX + 16 * NW - 1) / (16 * NW);   // LDS-DMA instructions per wave and unit (16 rows each)
constexpr int BUF = SMAX * 64;               // one staged unit: 64 B per row (fp32 in flight, then hi | lo fp16)
#ifndef SGP_SPLIT_VSTAGE
#define SGP_SPLIT_VSTAGE 0
#endif
// VSTAGE (wide form): a unit's pieces travel through REGISTERS instead of LDS-DMA -- global_load_dwordx4 into 4 VGPRs per
// piece, split there, written once in the operand layout.  The LDS then moves 48 KB per unit for staging instead of 144
// (DMA in + conversion read + write back) and needs TWO buffers (being written | being multiplied), which leaves room
// for more staged rows per tile.  Needs 4 NLD registers for the pieces in flight: the form with 256 registers per wave.
constexpr bool VSTAGE = SGP_SPLIT_VSTAGE != 0;
constexpr int NBUF = VSTAGE ? 2 : 3;         // (landing |) being converted | being multiplied
constexpr int HDR = 64;                      // ints per tile header: [NW : 2 NW] rows of every wave, [2 NW] staged rows U
constexpr int MAXFEAT = 1024;                // scale table: 2 x feat floats behind the staging buffers
#ifndef SGP_SPLIT_CR
#define SGP_SPLIT_CR 3
#endif
constexpr int CR = SGP_SPLIT_CR;             // chunk behind which the conversion of unit u + 1 starts (>= NLD: after the staging requests)
static_assert(BUF < 65536 - 512, "packed 16-bit transpose-read addresses");
static_assert(VSTAGE ? (CR >= 1 && CR + NLD <= NCH) : (CR >= NLD - 1 && CR + NLD < NCH),
              "conversion sits between the staging requests and the last chunk");
static_assert(NLD <= 7 && NBUF * BUF + 2 * MAXFEAT * 4 <= 160 * 1024, "three buffers and the scale table in 160 KB");

#ifndef SGP_SPLIT_RING
#define SGP_SPLIT_RING 3
#endif
constexpr int RING = SGP_SPLIT_RING;         // operand registers: the chunk being multiplied + RING - 1 requested ahead
static_assert(RING >= 2 && RING <= 4, "one to three chunks of operands in flight");

// LDS operations the conversion issues behind the MFMAs of chunk c (scale + piece 0 | write 2, read 1 | ... | write 2)
// (VSTAGE: the scale quad behind chunk CR - 1, two writes behind each of chunks CR .. CR + NLD - 1)
constexpr int conv_ops(int c) {
    if (VSTAGE) return (c == CR - 1 ? 1 : 0) + (c >= CR && c < CR + NLD ? 2 : 0);
    return c == CR ? 2 : (c > CR && c < CR + NLD ? 3 : (c == CR + NLD ? 2 : 0));
}
// operations issued behind chunk c's operand reads when the wave waits for them: the reads of the chunks requested
// since (4 each) and the conversion steps of the chunks in between
#ifndef SGP_SPLIT_ACC3
#define SGP_SPLIT_ACC3 0
#endif
#ifndef SGP_SPLIT_SPREAD
#define SGP_SPLIT_SPREAD 0
#endif
constexpr bool SPREAD = SGP_SPLIT_SPREAD != 0;   // the newest chunk's lo reads are issued behind the first MFMA instead of in front of the wait
constexpr int lgkm_behind(int c, bool conv) {
    int n = 4 * ((c + RING - 1 < NCH ? c + RING - 1 : NCH - 1) - c);
    if (SPREAD && c + RING - 1 < NCH) n -= 2;
    if (conv) for (int j = (c - (RING - 1) > 0 ? c - (RING - 1) : 0); j < c; ++j) n += conv_ops(j);
    return n;
}
// lgkmcnt holds 15: a count beyond it is clamped -- the wait is then stricter than needed (correct, less overlap)
constexpr int lgkm_wait(int c, bool conv) { return lgkm_behind(c, conv) < 15 ? lgkm_behind(c, conv) : 15; }
static_assert(RING == 4 || (lgkm_behind(NCH - 1, true) <= 15 && lgkm_behind(CR + 1, true) <= 15 && lgkm_behind(CR + 2, true) <= 15),
              "the standard ring's waits are exact");

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// ablation / timeline switches (SGP_TUNE=split_abl=..) exist only in builds with -DSGP_ABLATION (tools/build_variant.sh):
// the product kernel carries none of their tests
#ifdef SGP_ABLATION
#define ABL(bit) ((a.mode & (bit)) != 0)
#else
#define ABL(bit) false
#endif

struct SplitArgs {
    const int* hdr; const int* rowid; const int* ucol; const h8* afr; const int* adr; const float* rinv;
    int n_tiles, tiles_per_xcd;
    int time_major;                          // 1: an XCD walks ITS time chunks (chunk % 8 == xcd) over ALL tiles -- a step's whole slab stays in its L2
    const float* X; long long xrs, xbs;
    const float* XH; long long xhrs, xhbs;   // halo source (columns >= n_own): local block of a node partition
    int n_own;
    float* Y; long long yrs, ybs;
    int batch, nslice, t_chunk;
    const float* xtab;                       // [2][16 nslice]: per-column scale, then its inverse
    const int* pred; int pred_want;          // launch predicate (`pred, run_if` of the entry): run only if *pred == pred_want
    unsigned long long* dbg;                 // mode 256: per-wave s_memtime stamps of workgroup 0
    int mode;                                // ablations (SGP_TUNE=split_abl=..): 1 no loads, 2 no MFMAs, 4 no stores, 8 no conversion, 16 all tiles of an XCD stage the same rows, 32 unpaired stores, 64 all loads hit the L2
};

// B^T operand of one chunk: four transpose reads (hi / lo piece x rows k = 0..3 / 4..7 of every lane group).  Issued
// from asm so that the wait in front of the MFMAs can be COUNTED (LDS returns in order).
struct BOp { s4v h0, h1, l0, l1; };
__device__ __forceinline__ void tr_issue(BOp& b, unsigned a0, unsigned a1) {
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:256\n\tds_read_b64_tr_b16 %3, %5 offset:256"
                 : "=&v"(b.h0), "=&v"(b.h1), "=&v"(b.l0), "=&v"(b.l1) : "v"(a0), "v"(a1) : "memory");
}
// the same in two halves (hi pieces | lo pieces): -DSGP_SPLIT_SPREAD=1 puts an MFMA between them
__device__ __forceinline__ void tr_issue_hi(BOp& b, unsigned a0, unsigned a1) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(b.h0), "=&v"(b.h1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void tr_issue_lo(BOp& b, unsigned a0, unsigned a1) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:256\n\tds_read_b64_tr_b16 %1, %3 offset:256" : "=&v"(b.l0), "=&v"(b.l1) : "v"(a0), "v"(a1) : "memory");
}
template <int N> __device__ __forceinline__ void tr_wait(BOp& b) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b.h0), "+v"(b.h1), "+v"(b.l0), "+v"(b.l1) : "n"(N));
}
__device__ __forceinline__ h8 cat8(s4v x, s4v y) {
    return __builtin_bit_cast(h8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
}
// the two byte addresses of a chunk's transpose reads travel as 16-bit halves of one register
__device__ __forceinline__ unsigned addr_lo(unsigned base, unsigned packed) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(base), "v"(packed));
    return r;
}
__device__ __forceinline__ unsigned addr_hi(unsigned base, unsigned packed) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(base), "v"(packed));
    return r;
}

// LDS traffic of the conversion, in the same counted stream as the transpose reads
__device__ __forceinline__ void lds_read16(f32x4& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=&v"(v) : "v"(addr) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_read16_off(f32x4& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_write8(unsigned addr, uint2 d) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(d), "n"(OFF) : "memory");
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

// VSTAGE: 16 B per lane into registers (scalar base + 32-bit lane offset / full per-lane address); the wait that
// retires them names the registers, so nothing reads them before
__device__ __forceinline__ void vload16(f32x4& d, unsigned voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void vload16_vaddr(f32x4& d, const void* vaddr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(d) : "v"(vaddr) : "memory");
}

#ifndef SGP_SPLIT_STORE_MOD
#define SGP_SPLIT_STORE_MOD ""
#endif
// result rows: scalar base (the step's slice) + per-lane 32-bit byte offset; OFF = -64 reaches the even half of the line
template <int OFF> __device__ __forceinline__ void store16(const void* sbase, unsigned voff, f32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 " SGP_SPLIT_STORE_MOD :: "v"(voff), "v"(d), "s"(sbase), "n"(OFF) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_n(int n) {       // n is wave-uniform, 0 .. NLD
    switch (n) {
        case 0: wait_vm<0>(); break; case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break;
        case 3: wait_vm<3>(); break; case 4: wait_vm<4>(); break; case 5: wait_vm<5>(); break;
        case 6: wait_vm<6>(); break; default: wait_vm<7>(); break;
    }
}

// v * s = hi + lo in 8 instructions per 4 values: hi = fp16(v * s), lo = fp16(v * s - hi) as ONE fused operation each
// (v_fma_mixlo / mixhi_f16: fp32 fma of (fp32 v, fp32 s, fp16 half of a register), rounded once to fp16 into the low
// / high half of the destination) -- the remainder of an 11-bit rounding of a 24-bit value is exact in the fma.
__device__ __forceinline__ void split4(const f32x4 v, const f32x4 s, uint2& hi, uint2& lo) {
    // (the low-half instruction leaves the other half of its destination alone and the high-half one then writes it: no
    // zeroed destinations -- four v_mov per piece less)
    unsigned h01, h23, l01, l23;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h01) : "v"(v[0]), "v"(s[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(v[1]), "v"(s[1]));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h23) : "v"(v[2]), "v"(s[2]));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(v[3]), "v"(s[3]));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(l01) : "v"(v[0]), "v"(s[0]), "v"(h01));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(v[1]), "v"(s[1]), "v"(h01));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(l23) : "v"(v[2]), "v"(s[2]), "v"(h23));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(v[3]), "v"(s[3]), "v"(h23));
    hi.x = h01; hi.y = h23; lo.x = l01; lo.y = l23;
}

template <bool HALO, bool ACC>
__global__ __launch_bounds__(NW * 64, NW / 4) void spmm_split(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (a.pred != nullptr && a.pred[0] != a.pred_want) return;

    // XCD x (= blockIdx % 8) walks its own contiguous range of tiles, time chunk by time chunk, so the 32
    // workgroups an XCD runs side by side are neighbouring tiles of the same steps (their staged rows overlap)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    // small operators (the source rows of one step fit an L2): time-major -- every tile of a time chunk runs on the SAME
    // XCD side by side, so a staged row is fetched from the fabric once per step instead of once per XCD that holds a
    // tile referencing it
    const int tile = a.time_major ? j % a.n_tiles : xcd * a.tiles_per_xcd + j % a.tiles_per_xcd;
    const int tchunk = a.time_major ? (j / a.n_tiles) * 8 + xcd : j / a.tiles_per_xcd;
    if (tile >= a.n_tiles) return;
    const int t_begin = tchunk * a.t_chunk;
    const int t_end = min(a.batch, t_begin + a.t_chunk);
    if (t_begin >= t_end) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int src_tile = ABL(16) ? xcd * a.tiles_per_xcd : tile;   // 16: every tile of an XCD stages the same rows
    const int nU = __builtin_amdgcn_readfirstlane(a.hdr[(size_t)src_tile * HDR + 2 * NW]);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int feat = a.nslice * 16;

    // ---- scale table of the columns (scale | inverse) behind the staging buffers
    {
        float* tab = (float*)(lds + NBUF * BUF);
        for (int i = tid; i < 2 * feat; i += NW * 64) tab[i] = a.xtab[i];
    }
    // ---- resident plan: A fragments and per-lane row addresses of the transpose reads
    h8 af[NCH][2];
    unsigned ad[NCH];
    {
        const h8* ap = a.afr + ((size_t)(tile * NW + wave) * NCH * 2) * 64 + lane;
        const int* dp = a.adr + ((size_t)(tile * NW + wave) * NCH) * 64 + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            af[c][0] = ap[(c * 2 + 0) * 64];
            af[c][1] = ap[(c * 2 + 1) * 64];
            ad[c] = (unsigned)dp[c * 64];
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
        // (32-bit arithmetic: the host checks rows * stride < 2^29 elements)
        if (HALO && c >= a.n_own)                              // bit 31 marks a halo row (offsets stay below 2^31)
            xoff[i] = 0x80000000u | ((unsigned)(c - a.n_own) * (unsigned)a.xhrs * 4u + (unsigned)(lane & 3) * 16u);
        else
            xoff[i] = (unsigned)c * (unsigned)a.xrs * 4u + (unsigned)(lane & 3) * 16u;
        if ((i * NW + wave) * 16 < nU) nld = i + 1;
    }
    // this lane's result row: slot lane & 15 of the wave (-1 = empty), features 4 q .. 4 q + 3 of the slice, q = lane / 16
    const int* rid = a.rowid + (size_t)(tile * NW + wave) * 16;
    const int row_a = rid[lane & 15];
    const float rinv = a.rinv[(size_t)(tile * NW + wave) * 16 + (lane & 15)];
    const unsigned yo = row_a < 0 ? ~0u : (unsigned)row_a * (unsigned)a.yrs * 4u + 16u * (unsigned)(lane >> 4);   // byte offset (< 2^32: host check)
    // the plan loads above must be retired by a wait THE COMPILER CAN SEE (vmcnt(0), other counters untouched): with an
    // asm wait it keeps its own scoreboard open and guards the first use of the fragments -- the first MFMAs of EVERY
    // unit -- with s_waitcnt vmcnt(3), which at run time waits for the staging pieces in flight (measured: the round-5
    // pipelined loop at 18.7 ms per hop with those waits).  From here on vmcnt is counted by hand.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // own rows inside a buffer: row s = (i NW + wave) 16 + lane / 4 at s * 64, this lane's 16 B at + (lane & 3) * 16
    const unsigned own = lds0 + wave * 1024 + lane * 16;
    // converted layout, in place: a group of 8 staged rows (512 B) keeps the hi pieces of row r at 32 r and the lo
    // pieces at 256 + 32 r -- eight consecutive rows cover all 64 banks with either piece, and lo = hi + 256 is an
    // immediate offset of the transpose reads.  A wave instruction covers two whole groups, so every read of a
    // group has returned before its first write goes out.
    const unsigned cv_off = lds0 + wave * 1024 + (lane >> 5) * 512 + ((lane >> 2) & 7) * 32 + (lane & 3) * 8;
    // scale quads: conversion -- features 4 (lane & 3) .. + 3 of the slice; result -- inverse scales of 4 q .. 4 q + 3
    const unsigned tab_cv = lds0 + NBUF * BUF + (lane & 3) * 16;
    const unsigned tab_rs = lds0 + NBUF * BUF + feat * 4 + (lane >> 4) * 16;

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

    const int n_units = (t_end - t_begin) * a.nslice;
    // DMA cursor (two units ahead of the multiply) and multiply cursor
    int dt = t_begin, dsl = 0;
    auto advance = [&](int& t, int& sl) { if (++sl == a.nslice) { sl = 0; ++t; } };

    // VSTAGE: the pieces of the unit AFTER the one being multiplied, in registers from their request (behind that piece's
    // conversion one unit earlier) to their conversion
    f32x4 ld[VSTAGE ? NLD : 1];
#pragma unroll
    for (int i = 0; i < (VSTAGE ? NLD : 1); ++i) ld[i] = f32x4{0, 0, 0, 0};
    auto vpiece = [&](f32x4& d, unsigned off, const float* xb, const float* xh) {
        if constexpr (HALO) {
            const char* b = (off & 0x80000000u) ? (const char*)xh : (const char*)xb;
            vload16_vaddr(d, b + (off & 0x7fffffffu));
        } else {
            vload16(d, off, xb);
        }
    };
    auto issue_loads = [&](int t, int sl) {
        const float* xb = a.X + (long long)t * a.xbs + sl * 16;
        const float* xh = HALO ? a.XH + (long long)t * a.xhbs + sl * 16 : nullptr;
        // (every piece, unconditionally -- pieces past the tile's last staged row re-load row 0, as the DMA form's idle
        // lanes do: a load under a branch makes the compiler merge the loaded registers with their old value, and it did
        // so by COPYING a register whose load was still in flight)
        static_for<0, (VSTAGE ? NLD : 0)>([&](auto I) {
            constexpr int i = decltype(I)::value;
            vpiece(ld[i], xoff[i], xb, xh);
        });
    };
    if constexpr (VSTAGE) {
        // ---- prologue: unit 0 loaded, split and written; unit 1 requested
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // the scale table is in place
        issue_loads(dt, dsl); advance(dt, dsl);
        f32x4 s4;
        lds_read16(s4, tab_cv);
        wait_vm<0>();
        static_for<0, NLD>([&](auto I) {
            constexpr int i = decltype(I)::value;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s4), "+v"(ld[i]) :: "memory");
            uint2 hi, lo;
            split4(ld[i], s4, hi, lo);
            lds_write8<i * NW * 1024>(cv_off, hi); lds_write8<i * NW * 1024 + 256>(cv_off, lo);
        });
        if (n_units <= 1) { dt = t_begin; dsl = 0; }                                 // (a single unit: its own pieces again, unused)
        issue_loads(dt, dsl); advance(dt, dsl);
        if (n_units <= 2) { dt = t_begin; dsl = 0; }                                 // the running pointer below never leaves the chunk
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        // ---- prologue: units 0 and 1 requested, unit 0 converted
        issue_dma(dt, dsl, 0); advance(dt, dsl);
        if (n_units > 1) { issue_dma(dt, dsl, 1); advance(dt, dsl); wait_vm_n(nld); } else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // the scale table is in place
        {
            f32x4 v[NLD], s4;
            lds_read16(s4, tab_cv);
#pragma unroll
            for (int i = 0; i < NLD; ++i) lds_read16(v[i], own + i * (NW * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s4), "+v"(v[0]), "+v"(v[NLD - 1]), "+v"(v[NLD / 2]));
            static_for<0, NLD>([&](auto I) {
                constexpr int i = decltype(I)::value;
                uint2 hi, lo;
                split4(v[i], s4, hi, lo);
                lds_write8<i * NW * 1024>(cv_off, hi); lds_write8<i * NW * 1024 + 256>(cv_off, lo);
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    }

    int t = t_begin, sl = 0;
    unsigned cur = 0, nxt = BUF, nn = 2 * BUF;                                  // byte offsets of the three buffers
    f32x4 h0 = {0, 0, 0, 0};
    auto stamp = [&](int u, int k) {
        if (ABL(256) && blockIdx.x == 0 && lane == 0 && u >= 16 && u < 24)
            a.dbg[((u - 16) * NW + wave) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    // running pointers: the slice of unit u + 2 (staging requests), of unit u (result rows); a step's last slice wraps
    const float* xb2 = ABL(64) ? a.X : a.X + (long long)dt * a.xbs + dsl * 16;
    const float* xh2 = HALO ? a.XH + (long long)dt * a.xhbs + dsl * 16 : nullptr;
    const long long x_wrap = a.xbs - 16 * (a.nslice - 1), xh_wrap = a.xhbs - 16 * (a.nslice - 1);
    float* ys = a.Y + (long long)t * a.ybs;
    const long long y_wrap = a.ybs - 16 * (a.nslice - 1);
    int sl1 = a.nslice > 1 ? 1 : 0;                                             // slice of unit u + 1
    for (int u = 0; u < n_units; ++u) {
        stamp(u, 0);
        // unit u + 2 is requested one piece per chunk, unit u + 1 converted between the chunks, unit u multiplied.
        // (The conversion also runs in a time chunk's last unit, on a buffer nobody reads: no branch around it.)
        const bool dma_now = u + 2 < n_units && !ABL(1);
        const unsigned base2 = lds0 + nn + wave * 1024;
        const unsigned cbo = lds0 + cur;

        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#if SGP_SPLIT_ACC3
        f32x4 acc2 = {0, 0, 0, 0};
#endif
        f32x4 v, s4;
        BOp b[RING];
        stamp(u, 1);
        // LDS operations are issued in program order and return in order: every wait counts the operations issued
        // BEHIND the ones it needs (lgkm_behind).  Per chunk c: request the operands of chunk c + RING - 1, wait for chunk
        // c's, multiply; behind the MFMAs of chunks CR .. CR + NLD sits one step of the conversion of unit u + 1:
        // (CR) read the column scales and piece 0 | (CR + i) split piece i - 1, write it back, read piece i.
#pragma unroll
        for (int c = 0; c < RING - 1; ++c) tr_issue(b[c], addr_lo(cbo, ad[c]), addr_hi(cbo, ad[c]));
        static_for<0, NCH>([&](auto C) {
            constexpr int c = decltype(C)::value;
            BOp& x = b[c % RING];
            unsigned na0 = 0, na1 = 0;
            if constexpr (c + RING - 1 < NCH) {
                na0 = addr_lo(cbo, ad[c + RING - 1]); na1 = addr_hi(cbo, ad[c + RING - 1]);
                if constexpr (SPREAD) tr_issue_hi(b[(c + RING - 1) % RING], na0, na1);
                else tr_issue(b[(c + RING - 1) % RING], na0, na1);
            }
            if (ABL(8)) tr_wait<lgkm_wait(c, false)>(x); else tr_wait<lgkm_wait(c, true)>(x);
            if (!ABL(2)) {
                const h8 bh = cat8(x.h0, x.h1), bl = cat8(x.l0, x.l1);
                // the cross terms go to a second accumulator so that consecutive MFMAs do not wait for each other
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, af[c][0], acc0, 0, 0, 0);
                if constexpr (SPREAD && c + RING - 1 < NCH) tr_issue_lo(b[(c + RING - 1) % RING], na0, na1);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, af[c][0], acc1, 0, 0, 0);
#if SGP_SPLIT_ACC3
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, af[c][1], acc2, 0, 0, 0);
#else
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, af[c][1], acc1, 0, 0, 0);
#endif
            } else if constexpr (SPREAD && c + RING - 1 < NCH) tr_issue_lo(b[(c + RING - 1) % RING], na0, na1);
            if constexpr (VSTAGE) {
                // behind chunk CR - 1: the scale quad of unit u + 1's slice; behind chunk CR + i: piece i of unit u + 1 --
                // requested one unit ago, so at most the nld - 1 requests made since may still be in flight (loads retire
                // in order; stores in between only make the wait stricter) -- is split and written into the other
                // buffer, and piece i of unit u + 2 is requested into the registers it leaves
                if constexpr (c == CR - 1) lds_read16(s4, tab_cv + sl1 * 64);
                if constexpr (c >= CR && c < CR + NLD) {
                    constexpr int i = c - CR;
                    // (no branch around any of it: in a time chunk's last units the conversion writes a buffer nobody
                    // reads and the requests repeat the chunk's last slice)
                    wait_vm<NLD - 1>();
                    if constexpr (i == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(s4), "+v"(ld[i]) : "n"(c + RING - 1 < NCH ? 4 : 0) : "memory");
                    else asm volatile("" : "+v"(ld[i]) :: "memory");
                    uint2 hi, lo;
                    split4(ld[i], s4, hi, lo);
                    const unsigned w = cv_off + nxt;
                    lds_write8<i * NW * 1024>(w, hi); lds_write8<i * NW * 1024 + 256>(w, lo);
                    vpiece(ld[i], xoff[i], xb2, xh2);
                }
            } else {
            if constexpr (c < NLD) { if (dma_now && c < nld) piece(xoff[c], xb2, xh2, base2 + c * (NW * 1024)); }
            }
            if constexpr (!VSTAGE && c >= CR && c <= CR + NLD) {
                if (!ABL(8)) {
                    constexpr int i = c - CR;                  // piece to read now; piece i - 1 is split and written
                    if constexpr (i == 0) {
                        // the pieces of unit u + 1 were requested a whole unit ago; only the nld newest requests (unit u + 2,
                        // made in chunks 0 .. NLD - 1 of this unit) may stay in flight -- loads retire in order, so a count of
                        // nld is reached only with every older load done, whatever the stores in between do
                        if (dma_now) wait_vm_n(nld); else wait_vm<0>();
                        lds_read16(s4, tab_cv + sl1 * 64);
                    } else {
                        // behind piece i - 1's read: the operand reads this chunk requested
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(s4), "+v"(v) : "n"(c + RING - 1 < NCH ? 4 : 0));
                        uint2 hi, lo;
                        split4(v, s4, hi, lo);
                        const unsigned w = cv_off + nxt;
                        lds_write8<(i - 1) * NW * 1024>(w, hi); lds_write8<(i - 1) * NW * 1024 + 256>(w, lo);
                    }
                    if constexpr (i < NLD) lds_read16_off<i * NW * 1024>(v, own + nxt);
                }
            }
        });
        stamp(u, 3);
        if (!ABL(4)) {
            f32x4 iv;
            lds_read16(iv, tab_rs + sl * 64);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(iv));
#if SGP_SPLIT_ACC3
            acc1 += acc2;
#endif
            const f32x4 r0 = (acc0 + acc1) * (iv * rinv);
            // an even slice waits for its odd neighbour: the two 64-byte halves of a 128-byte line leave together
            if (!(sl & 1) && sl + 1 < a.nslice && !ABL(32)) {
                h0 = r0;
            } else if (yo != ~0u) {
                if constexpr (ACC) {
                    char* yb = (char*)ys + yo;
                    if ((sl & 1) && !ABL(32)) { *(f32x4*)(yb - 64) += h0; *(f32x4*)yb += r0; }
                    else *(f32x4*)yb += r0;
                } else {
                    if ((sl & 1) && !ABL(32)) { store16<-64>(ys, yo, h0); store16<0>(ys, yo, r0); }
                    else store16<0>(ys, yo, r0);
                }
            }
        }
        stamp(u, 5);
        // advance the cursors: staging (unit u + 2 -> u + 3), result rows (u -> u + 1), scale row of unit u + 2
        if (VSTAGE ? u + 3 < n_units : dma_now) {             // (VSTAGE requests unconditionally: the pointer stays inside the chunk)
            const bool wrap = ++dsl == a.nslice;
            if (wrap) dsl = 0;
            if (!ABL(64)) xb2 += wrap ? x_wrap : 16;
            if (HALO) xh2 += wrap ? xh_wrap : 16;
        }
        {
            const bool wrap = sl + 1 == a.nslice;
            ys += wrap ? y_wrap : 16;
            sl = sl1;
            sl1 = sl1 + 1 == a.nslice ? 0 : sl1 + 1;
            if (wrap) ++t;
        }
        if constexpr (VSTAGE) { const unsigned f = cur; cur = nxt; nxt = f; }
        else { const unsigned f = cur; cur = nxt; nxt = nn; nn = f; }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(u, 6);
    }
}

}  // namespace

extern "C" int32_t SGP_SPLIT_NAME(chunks)(void) { return NCH; }
extern "C" int32_t SGP_SPLIT_NAME(max_union)(void) { return SMAX; }
extern "C" int32_t SGP_SPLIT_NAME(waves)(void) { return NW; }
extern "C" int32_t SGP_SPLIT_NAME(rows_per_wave)(void) { return 16; }
extern "C" int32_t SGP_SPLIT_NAME(max_feat)(void) { return MAXFEAT; }

extern "C" int SGP_SPLIT_NAME(f32)(const int32_t* hdr, const int32_t* rowid, const int32_t* ucol, const void* afr,
                                  const int32_t* adr, const float* rinv,
                                  int32_t n_tiles,
                                  const float* X, int64_t x_row_stride, int64_t x_batch_stride,
                                  const float* X_halo, int64_t xh_row_stride, int64_t xh_batch_stride, int32_t n_own,
                                  float* Y, int64_t y_row_stride, int64_t y_batch_stride,
                                  int32_t n_rows, int32_t n_cols, int32_t batch, int32_t feat,
                                  const float* x_tab, int32_t accumulate, int32_t t_chunk, const int32_t* pred, int32_t run_if, sgp_stream_t stream) {
    const sgp::Predicate pr{pred, run_if};
    SGP_REQUIRE(n_tiles >= 0 && batch >= 0 && n_rows >= 0 && n_cols >= 0, "spmm_split: negative size");
    if (n_tiles == 0 || batch == 0 || n_rows == 0) return 0;
    SGP_REQUIRE(hdr && rowid && ucol && afr && adr && rinv && X && Y && x_tab, "spmm_split: null pointer");
    SGP_REQUIRE(feat > 0 && feat % 16 == 0 && feat <= MAXFEAT, "spmm_split: feat = %d is not a multiple of 16 up to %d", feat, MAXFEAT);
    SGP_REQUIRE(sgp::aligned16(X) && x_row_stride % 4 == 0 && x_batch_stride % 4 == 0 &&
                sgp::aligned16(Y) && y_row_stride % 4 == 0 && y_batch_stride % 4 == 0,
                "spmm_split: X and Y rows must be 16-byte aligned");
    {
        const long long own = X_halo ? n_own : n_cols, far = X_halo ? n_cols - n_own : 0;
        SGP_REQUIRE(own >= 0 && far >= 0 && own * x_row_stride < (1ll << 29) && far * xh_row_stride < (1ll << 29) &&
                    (long long)n_rows * y_row_stride < (1ll << 30), "spmm_split: rows beyond 32-bit byte offsets");
        SGP_REQUIRE(!X_halo || (sgp::aligned16(X_halo) && xh_row_stride % 4 == 0 && xh_batch_stride % 4 == 0),
                    "spmm_split: halo rows must be 16-byte aligned");
    }
    SplitArgs a;
    a.hdr = hdr; a.rowid = rowid; a.ucol = ucol; a.afr = (const h8*)afr; a.adr = adr; a.rinv = rinv;
    a.n_tiles = n_tiles; a.tiles_per_xcd = (n_tiles + 7) / 8;
    a.X = X; a.xrs = x_row_stride; a.xbs = x_batch_stride;
    a.XH = X_halo; a.xhrs = xh_row_stride; a.xhbs = xh_batch_stride; a.n_own = X_halo ? n_own : 0x7fffffff;
    a.Y = Y; a.yrs = y_row_stride; a.ybs = y_batch_stride;
    a.batch = batch; a.nslice = feat / 16;
    a.xtab = x_tab; a.pred = pr.flag; a.pred_want = pr.want;
    const bool auto_tc = t_chunk <= 0;
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
    // time-major mapping where one step's source rows (n_cols x feat floats) fit an XCD's L2 beside the result rows
    // streaming out (SGP_TUNE=split_time_major=0|1 overrides; split_tc = time steps per workgroup there)
    // Measured (profiles/r6/ab_time_major.log, ms per hop, tile-major -> time-major at 16 steps per workgroup): PV-US shape
    // 100-NN 17.0 -> 15.4, its full graph (8 passes) 110.9 -> 92.8, N = 10 000 3.81 -> 3.62 (64 steps).
    static const long tm_tune = sgp::tune("split_time_major", -1);
    static const long tm_tc = sgp::tune("split_tc", 0);
    a.time_major = tm_tune >= 0 ? (int)(tm_tune != 0) : (int)((long long)n_cols * feat * 4 <= (4ll << 20) && n_tiles >= 8);
    if (a.time_major) {
        if (tm_tc > 0 && auto_tc) t_chunk = (int)tm_tc;
        else if (auto_tc && n_tiles <= 32 && t_chunk > 16) t_chunk = 16;       // (few tiles: short chunks keep an XCD's tiles on the same steps)
    }
    a.t_chunk = t_chunk;
#ifdef SGP_ABLATION
    static const int abl = (int)sgp::tune("split_abl", 0);
#else
    constexpr int abl = 0;
#endif
    a.mode = abl;
    a.dbg = nullptr;
    if (abl & 256) { if (hipMalloc(&a.dbg, 8 * NW * 8 * 8) != hipSuccess) return sgp::fail(SGP_EINVAL, "dbg alloc"); (void)hipMemset(a.dbg, 0, 8 * NW * 8 * 8); }
    const int n_tchunks = (batch + t_chunk - 1) / t_chunk;
    auto kern = X_halo ? (accumulate ? spmm_split<true, true> : spmm_split<true, false>)
                       : (accumulate ? spmm_split<false, true> : spmm_split<false, false>);
    const int lds_bytes = NBUF * BUF + 2 * feat * 4;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * BUF + 2 * MAXFEAT * 4);
    if (e != hipSuccess) return sgp::fail((int)e, "spmm_split: LDS attribute: %s", hipGetErrorString(e));
    const unsigned grid = a.time_major ? 8u * (unsigned)n_tiles * (unsigned)((n_tchunks + 7) / 8)
                                       : 8u * (unsigned)a.tiles_per_xcd * (unsigned)n_tchunks;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds_bytes, (hipStream_t)stream, a);
    if (abl & 256) {
        unsigned long long h[8 * NW * 8];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(a.dbg);
        printf("spmm_split timeline (cycles since the unit's top; columns: top | chunk loop start | - | chunk loop done | - | stores | barrier)\n");
        for (int u = 0; u < 8; ++u) for (int w = 0; w < NW; ++w) {
            const unsigned long long* r = h + (u * NW + w) * 8;
            printf("  unit %d wave %d: top %llu |", u, w, r[0] - h[0]);
            for (int k = 1; k < 7; ++k) printf(" %6lld", (long long)(r[k] - r[0]));
            printf("\n");
        }
    }
    return sgp::check_launch("spmm_split");
}
