// Persistent DynGESN sequence kernel (reference: lib/nn/reservoir/graph_reservoir.py:85-93 stepped
// by tsl/nn/blocks/encoders/gcrnn.py:67-93; SURVEY.md 8f row f2: "fused persistent GEMM -> SpMM ->
// tanh step").  gfx950 / wave64 only.
//
//     h_l(t) = (1 - a_l) h_l(t-1) + a_l act( p_l(t) + A_hat z_l(t) ),   z_l(t) = h_l(t-1) W_hh,l^T,
//     p_0(t) = x(t) W_ih,0^T + b_0,   p_l(t) = h_{l-1}(t) W_ih,l^T + b_l
//
// The graph product sits inside the recurrence, so every step is a chip-wide dependency; the graphs
// of this baseline are small (207 / 325 nodes x 320 units x 3 layers) and the step is latency-bound.
// gesn.hip issues two launches per (step, layer): six dependent kernels, ~40 us per step.  Here ONE
// cooperative launch runs a chunk of steps:
//  * layers form a wavefront: in tick k layer l works on step k - l, so all layers are busy at once
//    and a tick needs ONE grid barrier (the sequence is tc + L - 1 ticks);
//  * a workgroup owns (layer l, a group of 64 output columns of [W_hh,l ; W_ih,l+1], RT row tiles of
//    16 nodes) for the whole launch: its slice of the weights lives in REGISTERS as MFMA operands
//    (R / 16 VGPRs per wave), its nodes' states and adjacency rows in LDS;
//  * per tick: "update" -- a wave serves two nodes (one per 32-lane half): gathers the neighbours' z
//    rows, adds p, applies activation + leak, writes h' to LDS (every column group of a row tile
//    repeats this: it needs the whole h' row as its GEMM operand, and repeating a 32-node gather is
//    cheaper than another barrier) -- then "GEMM" -- C[16 nodes, 64 cols] = h' W^T on
//    v_mfma_f32_16x16x4_f32, K split over the 4 wave quarters and reduced through LDS -- into the
//    tick's parity of a double buffer that holds z_l for the next step and p_{l+1} for this one.
// The first GEMM of a launch runs on the initial states with the same code, so a sequence cut into
// several calls is bit-identical to one call.  The XCDs' L2s are not coherent with each other: the C
// buffers are read and written with `sc1` (agent-scope) accesses only (ld4_agent / st4_agent),
// everything static stays in registers / LDS.
// A barrier that does not complete (it cannot, under a cooperative launch) raises a flag after ~1 s
// instead of hanging the device; the host falls back to gesn.hip's path on any launch problem.
#include "common.h"
#include <stdlib.h>

using sgp::f32x4;

namespace sgp_gesn {

constexpr int kMaxLayers = 8;
constexpr int kMaxRT = 4;                  // row tiles per workgroup
constexpr int kEdgeCap = 1024;             // adjacency entries cached in LDS per workgroup (else read from global)

struct PArgs {
    const int* rowptr; const int* col; const float* val;
    const float* p0;                       // [tc][N][R] layer-0 input term of this chunk
    const float* wcat;                     // [L][2R][R]  rows 0..R-1 = W_hh,l; R..2R-1 = W_ih,l+1
    const float* wpk;                      // the same in MFMA fragment order, see pack_weights
    const float* bcat;                     // [L][2R]     zeros | b_{l+1}
    float* cbuf;                           // [2 parity][L][N][2R]   z_l | p_{l+1}
    float* h_state;                        // [L][N][R] in / out
    float* out; long long ors, oss;        // rows of this chunk's first step
    unsigned* bar;                         // kBarWords words: [0] top counter, [1] failure flag, group counters
    float alpha[kMaxLayers], om[kMaxLayers];   // leaking rate, (float)(1 - rate)
    int act, tc, N, R, L;
    int dbg;                               // SGP_GESN_DBG timing ablations (results invalid): 1 no gathers, 2 no GEMM, 4 no barrier wait
    int n_rt, rt_per_wg, n_rtg;
    int item_base[kMaxLayers + 1];         // first workgroup of layer l
    int n_cg[kMaxLayers];                  // column groups (64 wide) of layer l
};

// Everything the workgroups exchange (the C buffers) is read and written at agent scope
// (global_load / global_store ... sc1 -- what an agent-scope atomic compiles to: coherent across the
// XCDs' L2s on its own), so the barrier needs no L2 write-back / invalidate: with __threadfence() on
// both sides (buffer_wbl2 sc1 + buffer_inv sc1 from every wave) a tick cost 160 us, 2.5x the six-launch
// step it replaces.  16-byte form (no 128-bit atomic exists): issued from inline asm, the caller waits
// (s_waitcnt vmcnt) before it reads `r`; scalar base + one 32-bit lane offset + immediate = one
// address VGPR per row instead of two per load.
template <int IMM>
__device__ __forceinline__ void ld4_agent(f32x4& r, const float* sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 sc1" : "=v"(r) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
// tanh(x) = 1 - 2 / (1 + e^{2x}) on v_exp_f32 / v_rcp_f32 (the reservoir kernels' form, absolute error
// < 3e-7): libdevice's tanhf is ~60 instructions per value, 16 values per lane and step -- it was a
// third of the tick
__device__ __forceinline__ float tanh_fast(float x) {
    const float big = fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * 2.885390081777927f)), 1.f);
    const float x2 = x * x;                               // below 0.25 the odd polynomial keeps RELATIVE accuracy (reservoir_impl.h)
    const float p = fmaf(x2, fmaf(x2, fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
    return fabsf(x) < 0.25f ? x * p : big;
}
__device__ __forceinline__ void st4_agent(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

// Two levels: workgroup b arrives at group counter b % kBarGroups (own cache line each); the last
// arrival of a group bumps the top counter, which everybody polls.  One counter for all ~200
// workgroups serialises their read-modify-writes at the coherence point (6.5 us per tick measured);
// per-workgroup flags polled by one wave (4 coalesced loads per poll) came out the same as this (16.6 vs 16.0 us per tick).
constexpr int kBarGroups = 16;
constexpr int kBarSticky = 400;                    // failure flag of the whole call (gesn.hip clears it once per call)
constexpr int kBarWords = 16 + 16 * kBarGroups;   // [0] top, [1] failure flag, [16 + 16 g] group g
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned round, int* lds_flag) {
    // my stores have been acknowledged by the coherence point before anyone can see my arrival
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned nb = gridDim.x, g = blockIdx.x % kBarGroups;
        const unsigned n_groups = nb < (unsigned)kBarGroups ? nb : (unsigned)kBarGroups;
        const unsigned gsize = (nb - g + kBarGroups - 1) / kBarGroups;
        const unsigned old = __hip_atomic_fetch_add(&bar[16 + 16 * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == round * gsize)
            __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = round * n_groups;
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0 &&
                (spins > (1u << 22) || __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                ok = 0;
                break;
            }
        }
        if (!ok) {
            __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // sticky copy outside the words that every launch clears: a time-out in an early chunk of a
            // call must still be visible when the host looks after the last one
            __hip_atomic_store(&bar[kBarSticky], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *lds_flag = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return *lds_flag != 0;
}

// Weight fragments in the order the GEMM waves read them: wpk[l][col tile c][K quarter ks][b][lane][u] =
// W_l[16 c + (lane & 15)][ks R/4 + 4 (4 b + u) + (lane >> 4)] (zero past the quarter / past the layer's
// output columns), so a wave's fragment is 4 KSB consecutive 16-byte reads per lane, 1 KiB per
// instruction.  (Read straight from the row-major matrix, one instruction touched 16 cache lines
// and the 24 loads of a wave took ~5 us of the CU's address path per tick.)
__global__ __launch_bounds__(64) void pack_weights(const float* wcat, float* wpk, int R, int L, int ksb) {
    const int lane = threadIdx.x, j = lane & 15, kq = lane >> 4;
    const int n_ct = 2 * R / 16;
    int id = blockIdx.x;
    const int ks = id & 3; id >>= 2;
    const int c = id % n_ct, l = id / n_ct;
    const int n_out = l + 1 < L ? 2 * R : R, KS = R >> 4;
    const float* wrow = wcat + (long long)l * 2 * R * R + (long long)min(16 * c + j, 2 * R - 1) * R + ks * (R >> 2) + kq;
    float* dst = wpk + ((long long)(l * n_ct + c) * 4 + ks) * ksb * 256 + lane * 4;
    for (int s = 0; s < 4 * ksb; ++s)
        dst[(s >> 2) * 256 + (s & 3)] = (16 * c + j < n_out && s < KS) ? wrow[4 * s] : 0.f;
}

// KSB = 4-step blocks of MFMA k-steps per wave: the wave's weight fragment is 4 KSB VGPRs (R <= 64 KSB)
template <int KSB>
__global__ __launch_bounds__(1024) void gesn_persistent(PArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = a.R, RP = R + 4, RT = a.rt_per_wg;
    // ---- which (layer, column group, row-tile group) is mine
    int layer = 0;
    while (layer + 1 < a.L && (int)blockIdx.x >= a.item_base[layer + 1]) ++layer;
    const int item = (int)blockIdx.x - a.item_base[layer];
    const int cg = item % a.n_cg[layer], rtg = item / a.n_cg[layer];
    const int n_out = layer + 1 < a.L ? 2 * R : R;
    const int node_lo = rtg * RT * 16, node_hi = min(a.N, (rtg * RT + RT) * 16);

    // ---- LDS carve-up
    // (row padding and a 16-float tail, all zero: the GEMM's padded k-steps read up to 11 floats past
    // a K quarter and multiply them by zero weights)
    float* hb = reinterpret_cast<float*>(lds_raw);                                   // [RT][16][RP] + 16
    f32x4* red = reinterpret_cast<f32x4*>(hb + RT * 16 * RP + 16);                   // [RT][4 ks][4 ct][64]
    int* ecol = reinterpret_cast<int*>(red + RT * 16 * 64);                          // [kEdgeCap]
    float* eval = reinterpret_cast<float*>(ecol + kEdgeCap);                         // [kEdgeCap]
    int* erow = reinterpret_cast<int*>(eval + kEdgeCap);                             // [RT * 16 + 1] my rows of rowptr
    int* flag = erow + kMaxRT * 16 + 1;

    // ---- static data -> registers / LDS
    const int e_lo = a.rowptr[node_lo], e_hi = a.rowptr[node_hi];
    const bool e_cached = e_hi - e_lo <= kEdgeCap;
    if (e_cached)
        for (int e = e_lo + tid; e < e_hi; e += 1024) { ecol[e - e_lo] = a.col[e]; eval[e - e_lo] = a.val[e]; }
    if (tid <= RT * 16) erow[tid] = a.rowptr[min(node_lo + tid, a.N)];
    const float* hs = a.h_state + (long long)layer * a.N * R;
    for (int i = tid; i < RT * 16 * RP + 16; i += 1024) {
        const int n = node_lo + i / RP, f = i % RP;
        hb[i] = (i < RT * 16 * RP && f < R && n < a.N) ? hs[(long long)n * R + f] : 0.f;
    }
    // GEMM role: wave (ct, ks) -> column tile cg * 4 + ct, K quarter ks
    const int ct = wave & 3, ks = wave >> 2;
    const int j = lane & 15, kq = lane >> 4;
    const int col0 = (cg * 4 + ct) * 16;
    const bool has_cols = col0 < n_out;                           // n_out is a multiple of 16
    const int KS = R >> 4;                                        // k-steps per wave
    const int k0 = ks * (R >> 2);
    // my weight fragment: 4 KSB values per lane, lane (kq, j) <-> W[col0 + j][k0 + 4 s + kq], read from
    // the packed copy.  Resident in registers across the whole launch when the update phase leaves
    // room (KEEP_W); for wide layers it is re-read from L2 at the top of every GEMM (KSB 16-byte loads
    // in one batch) -- kept "resident" there, hipcc spilled it to scratch and reloaded it one value per
    // MFMA (8 us per tick).
    constexpr bool KEEP_W = KSB <= 4;
    const unsigned wfrag_off = (unsigned)(((layer * (2 * R / 16) + min(cg * 4 + ct, 2 * R / 16 - 1)) * 4 + ks) * KSB * 64 + lane);
    const f32x4* wfrag = reinterpret_cast<const f32x4*>(a.wpk) + wfrag_off;
    f32x4 wkeep[KEEP_W ? KSB : 1];
    if constexpr (KEEP_W) {
#pragma unroll
        for (int b = 0; b < KSB; ++b) wkeep[b] = wfrag[b * 64];
    }
    const float alpha = a.alpha[layer], om = a.om[layer];
    const long long plane = (long long)a.L * a.N * 2 * R;        // one parity of cbuf
    float* const c_mine = a.cbuf + (long long)layer * a.N * 2 * R;
    const float* const c_below = a.cbuf + (long long)(layer > 0 ? layer - 1 : 0) * a.N * 2 * R;
    __syncthreads();

    // C[parity][node j, col0 + 4 kq .. + 3] = W[col0 + 4 kq + i, :] . h'[node j, :] + bias: the tile is
    // computed TRANSPOSED (A = my weight fragment, B = h' rows from LDS), so a lane ends up with four
    // consecutive columns of ONE node and stores 16 bytes; the h operands of 4 k-steps are requested
    // together (k-steps beyond R / 16 multiply zero weights).  The four K quarters of a tile meet in
    // LDS; the wave of quarter r sums and stores row tile r.
    auto gemm = [&](int parity) {
        if (has_cols) {
            f32x4 wreg[KSB];
            // (the fragment's address is re-made from one 32-bit offset here: kept as ready-made 64-bit
            // pointers across the tick loop -- one per 4 KiB of immediate range -- they cost the registers
            // that decide between spilling and not)
            unsigned wo = wfrag_off;
            asm volatile("" : "+v"(wo));
            const f32x4* wf = reinterpret_cast<const f32x4*>(a.wpk) + wo;
#pragma unroll
            for (int b = 0; b < KSB; ++b) {
                if constexpr (KEEP_W) wreg[b] = wkeep[b];
                else wreg[b] = wf[b * 64];
            }
            for (int r = 0; r < RT; ++r) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* hrow = hb + (r * 16 + j) * RP + k0 + kq;
#pragma unroll
                for (int b = 0; b < KSB; ++b) {
                    if (4 * b < KS) {
                        float hv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) hv[u] = hrow[4 * (4 * b + u)];   // (<= 11 floats past my K quarter: finite, see hb)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[b][u], hv[u], acc, 0, 0, 0);
                    }
                }
                red[((r * 4 + ks) * 4 + ct) * 64 + lane] = acc;
            }
        }
        __syncthreads();
        if (has_cols && ks < RT) {
            const int r = ks;
            f32x4 v = *reinterpret_cast<const f32x4*>(a.bcat + layer * 2 * R + col0 + 4 * kq);   // (L2-resident; not worth 4 VGPRs)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 o = red[((r * 4 + k) * 4 + ct) * 64 + lane];
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
            const int n = node_lo + r * 16 + j;
            if (n < a.N) st4_agent(c_mine + parity * plane + (long long)n * 2 * R + col0 + 4 * kq, v);
        }
    };

    // z of the first step comes from the initial states, by the same code as every later one
    gemm((layer + 1) & 1);
    unsigned round = 1;
    if (!grid_barrier(a.bar, round, flag)) return;

    // update role: a wave serves TWO nodes at a time, one per 32-lane half (flat index 2 (wave + 16 q)
    // + half over the RT x 16 nodes); lane l32 of a half owns features 4 (32 s + l32) .. + 3, s < S
    constexpr int S = KSB / 2;                                    // 16-byte slices per lane (R <= 128 S)
    constexpr int EB = KSB <= 2 ? 8 : (KSB == 4 ? 6 : 4);   // edges requested per round trip (VGPR budget)
    const int half = lane >> 5, l32 = lane & 31;
    const int n_ticks = a.tc + a.L - 1;
    for (int tick = 0; tick < n_ticks; ++tick) {
        const int t = tick - layer;
        if (t >= 0 && t < a.tc) {
            const int rd = (tick + 1) & 1, wr = tick & 1;
            const float* zbuf = c_mine + rd * plane;
            // ---- update.  All rows of a batch of EB edges (of both nodes) are requested before the
            // first is used -- 16-byte sc1 loads from inline asm, one hand-placed wait per batch:
            // written as one dword atomic per (edge, 64 features) every edge was its own round trip to
            // the coherence point (44 us per tick), one node per wave at a time 21 us.
            for (int q = 0; 32 * q < RT * 16; ++q) {
                const int idx = 2 * (wave + 16 * q) + half;       // my node among the workgroup's
                const int n = node_lo + idx;
                const bool live = idx < RT * 16 && n < a.N;
                const float* p0row = a.p0 + ((long long)t * a.N + n) * R;
                const float* pbase = c_below + rd * plane;                    // wave-uniform
                const unsigned poff = (unsigned)((n * 2 * R + R + 4 * l32) * 4);
                const int eb0 = live ? erow[idx] : 0, eb1 = live ? erow[idx + 1] : 0;
                const int deg = eb1 - eb0;
                const int deg_max = max(__builtin_amdgcn_readlane(deg, 0), __builtin_amdgcn_readlane(deg, 32));
                f32x4 v[S];
                bool ok[S];
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2) {
                    v[s2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    ok[s2] = live && 4 * (32 * s2 + l32) < R;
                    if (ok[s2]) {
                        if (layer == 0) v[s2] = *reinterpret_cast<const f32x4*>(p0row + 4 * (32 * s2 + l32));
                        else if (s2 == 0) ld4_agent<0>(v[s2], pbase, poff);
                        else if (s2 == 1) ld4_agent<512>(v[s2], pbase, poff);
                        else if (s2 == 2) ld4_agent<1024>(v[s2], pbase, poff);
                        else ld4_agent<1536>(v[s2], pbase, poff);
                    }
                }
                for (int e = 0; e < deg_max; e += EB) {
                    f32x4 z[EB][S];
                    float w[EB];
#pragma unroll
                    for (int i = 0; i < EB; ++i) {
                        w[i] = 0.f;
#pragma unroll
                        for (int s2 = 0; s2 < S; ++s2) z[i][s2] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (e + i < deg && !(a.dbg & 1)) {
                            const int ee = eb0 + e + i;
                            const int cj = e_cached ? ecol[ee - e_lo] : a.col[ee];
                            w[i] = e_cached ? eval[ee - e_lo] : a.val[ee];
                            const unsigned zoff = (unsigned)((cj * 2 * R + 4 * l32) * 4);
#pragma unroll
                            for (int s2 = 0; s2 < S; ++s2) {
                                if (ok[s2]) {
                                    if (s2 == 0) ld4_agent<0>(z[i][s2], zbuf, zoff);
                                    else if (s2 == 1) ld4_agent<512>(z[i][s2], zbuf, zoff);
                                    else if (s2 == 2) ld4_agent<1024>(z[i][s2], zbuf, zoff);
                                    else ld4_agent<1536>(z[i][s2], zbuf, zoff);
                                }
                            }
                        }
                    }
                    // (one asm per register set: the operand list of a single statement is limited)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int i = 0; i < EB; ++i)
#pragma unroll
                        for (int s2 = 0; s2 < S; ++s2) asm volatile("" : "+v"(z[i][s2]));
#pragma unroll
                    for (int s2 = 0; s2 < S; ++s2) asm volatile("" : "+v"(v[s2]));
#pragma unroll
                    for (int i = 0; i < EB; ++i) {
                        // (w = 0 and z = 0 past the end of the shorter list)
#pragma unroll
                        for (int s2 = 0; s2 < S; ++s2)
#pragma unroll
                            for (int m = 0; m < 4; ++m) v[s2][m] = fmaf(w[i], z[i][s2][m], v[s2][m]);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (nodes without edges)
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2) asm volatile("" : "+v"(v[s2]));
                float ss = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (a.act == SGP_ACT_TANH || a.act == SGP_ACT_TANH_REL) v[s2][m] = tanh_fast(v[s2][m]);
                        else if (a.act == SGP_ACT_RELU) v[s2][m] = fmaxf(v[s2][m], 0.f);
                        ss = fmaf(v[s2][m], v[s2][m], ss);        // lanes beyond R hold 0
                    }
                float inv = 1.f;
                if (a.act == SGP_ACT_SELF_NORM) {
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) ss += __shfl_xor(ss, off);   // within my half
                    inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
                }
                float* hrow = hb + idx * RP;
                float* orow = a.out + (long long)t * a.oss + (long long)n * a.ors + (long long)layer * R;
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2) {
                    const int f = 4 * (32 * s2 + l32);
                    if (ok[s2]) {
                        const f32x4 hp = *reinterpret_cast<const f32x4*>(hrow + f);
                        f32x4 hn;
#pragma unroll
                        for (int m = 0; m < 4; ++m) hn[m] = om * hp[m] + alpha * (v[s2][m] * inv);
                        *reinterpret_cast<f32x4*>(hrow + f) = hn;
                        if (cg == 0) *reinterpret_cast<f32x4*>(orow + f) = hn;
                    }
                }
            }
            __syncthreads();
            if (!(a.dbg & 2)) gemm(wr);
        }
        ++round;
        if (a.dbg & 4) { __syncthreads(); continue; }
        if (!grid_barrier(a.bar, round, flag)) return;
    }
    if (cg == 0) {
        float* hd = a.h_state + (long long)layer * a.N * R;
        for (int i = tid; i < RT * 16 * R; i += 1024) {
            const int n = node_lo + i / R, f = i % R;
            if (n < a.N) hd[(long long)n * R + f] = hb[(i / R) * RP + f];
        }
    }
}

int ksb_of(int R) { const int k = ((R >> 4) + 3) / 4; return k <= 2 ? 2 : (k <= 4 ? 4 : 6); }

int g_mode = -1;                           // -1: read SGP_TUNE=gesn_persistent=.. once; 0 off; 1 on
int mode() {
    if (g_mode < 0) g_mode = sgp::tune("gesn_persistent", 1) != 0;
    return g_mode;
}

size_t lds_bytes(int R, int RT) {
    return (size_t)(RT * 16 * (R + 4) + 16) * 4 + (size_t)RT * 16 * 64 * 16 + (size_t)kEdgeCap * 8 + (kMaxRT * 16 + 1 + 3) * 4;
}

// Plans the launch; returns false when the shape is not served (the caller then uses gesn.hip's path).
bool plan(int N, int R, int L, PArgs& a, int& n_blocks, size_t& lds) {
    // (R <= 384: the 512-wide instantiation needs more than the 128 VGPRs a 1024-thread workgroup gets)
    if (R % 16 != 0 || R > 384 || L > kMaxLayers || N <= 0) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return false;
    a.n_rt = (N + 15) / 16;
    for (int rt = 1; rt <= kMaxRT; ++rt) {
        lds = lds_bytes(R, rt);
        if (lds > 160 * 1024) break;
        const int n_rtg = (a.n_rt + rt - 1) / rt;
        int items = 0;
        for (int l = 0; l < L; ++l) {
            a.n_cg[l] = ((l + 1 < L ? 2 * R : R) + 63) / 64;
            a.item_base[l] = items;
            items += a.n_cg[l] * n_rtg;
        }
        a.item_base[L] = items;
        if (items <= cus) {                // one 1024-thread workgroup per CU
            a.rt_per_wg = rt; a.n_rtg = n_rtg; n_blocks = items;
            return true;
        }
    }
    return false;
}

// One chunk of tc steps.  0 = done, > 0 = the launch was refused (caller falls back), < 0 = error.
int launch(PArgs& a, int n_blocks, size_t lds, hipStream_t stream) {
    const int ksb = ksb_of(a.R);                           // R <= 64 ksb
    const void* kern = ksb <= 2 ? reinterpret_cast<const void*>(gesn_persistent<2>)
                     : ksb <= 4 ? reinterpret_cast<const void*>(gesn_persistent<4>)
                                : reinterpret_cast<const void*>(gesn_persistent<6>);
    // The update phase issues asynchronous loads from inline asm: their destination registers must
    // stay put until the hand-placed wait.  A build whose register allocation spills (scratch > 0)
    // could store / reuse such a register early -- refuse it (the stepwise path then serves the call).
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, kern);
    if (e != hipSuccess || fa.localSizeBytes != 0) { (void)hipGetLastError(); return 1; }
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return 1; }
    e = hipMemsetAsync(a.bar, 0, kBarWords * sizeof(unsigned), stream);
    if (e != hipSuccess) { (void)hipGetLastError(); return 1; }
    void* params[] = {&a};
    e = hipLaunchCooperativeKernel(kern, dim3(n_blocks), dim3(1024), params, (unsigned)lds, stream);
    if (e != hipSuccess) { (void)hipGetLastError(); return 1; }
    return 0;
}

int run_chunk(const int32_t* rowptr, const int32_t* col, const float* val, const float* p0,
              const float* wcat, const float* wpk, const float* bcat, float* cbuf, float* h_state, float* out,
              long long ors, long long oss, unsigned* bar, const double* alpha, int act,
              int tc, int N, int R, int L, hipStream_t stream) {
    PArgs a;
    int n_blocks = 0;
    size_t lds = 0;
    if (ors % 4 != 0 || oss % 4 != 0 || !sgp::aligned16(out)) return 1;   // 16-byte row stores
    if (!plan(N, R, L, a, n_blocks, lds)) return 1;
    a.rowptr = rowptr; a.col = col; a.val = val; a.p0 = p0; a.wcat = wcat; a.wpk = wpk; a.bcat = bcat;
    a.cbuf = cbuf; a.h_state = h_state; a.out = out; a.ors = ors; a.oss = oss; a.bar = bar;
    for (int l = 0; l < kMaxLayers; ++l) {
        a.alpha[l] = l < L ? (float)alpha[l] : 0.f;
        a.om[l] = l < L ? (float)(1.0 - alpha[l]) : 0.f;
    }
    a.act = act; a.tc = tc; a.N = N; a.R = R; a.L = L;
    { static int dbg = -1; if (dbg < 0) dbg = (int)sgp::tune("gesn_dbg", 0); a.dbg = dbg; }
    return launch(a, n_blocks, lds, stream);
}

long long packed_floats(int R, int L) { return (long long)L * (2 * R / 16) * 4 * ksb_of(R) * 256; }
int pack(const float* wcat, float* wpk, int R, int L, hipStream_t stream) {
    hipLaunchKernelGGL(pack_weights, dim3(L * (2 * R / 16) * 4), dim3(64), 0, stream, wcat, wpk, R, L, ksb_of(R));
    return sgp::check_launch("gesn pack_weights");
}

}  // namespace sgp_gesn

extern "C" int sgp_gesn_tune(int32_t persistent) {
    if (persistent >= 0) sgp_gesn::g_mode = persistent != 0;
    return sgp_gesn::mode();
}
