// Fused multi-layer leaky echo-state reservoir: ALL layers of a narrow stacked reservoir in one
// launch (reference: lib/nn/reservoir/reservoir.py:170-180 -- the Python loop steps every layer
// inside one time step, layer l consuming layer l-1's NEW state of the same step).
//
//   h_l[t] = (1 - a_l) h_l[t-1] + a_l act( W_ih,l x_l[t] + b_l + W_hh,l h_l[t-1] ),
//   x_0[t] = x[t],  x_l[t] = h_{l-1}[t]
//
// (l, t) depends on (l-1, t) and (l, t-1): a wavefront.  A workgroup owns NTW node tiles of 16
// nodes; wave (tile slot ts, layer l) keeps h_l of its tile in registers for all T steps and, in
// iteration i, computes time step t = i - l: layer l works on step t while layer l+1 works on
// step t-1.  The hand-off h_l[t] -> x_{l+1}[t] goes through a double-buffered LDS slab in the
// accumulator layout of v_mfma_f32_16x16x4_f32 (lane = node + 16 q, register r <-> feature
// 16 jt + 4 q + r), which IS the B-operand layout of the consumer (reservoir_impl.h), so the
// producer stores its registers as they are; one s_barrier per iteration.  The layer-by-layer
// form (sgp_reservoir_f32 once per layer) runs L serial chains of T steps and re-reads every
// layer's slot from HBM; here the chain is T + L - 1 iterations and layer inputs never leave the
// CU.  PV-US shape (N = 5016, T = 8868, R = 16 x 8 layers): one chain of 8875 instead of 8 x 8868.
//
// The recurrent MFMAs of a step are issued before the input MFMAs, so the LDS read of the
// producer's state lands under them.  Nothing in an iteration waits for HBM: layer 0 requests
// its input rows 7 steps ahead by LDS-DMA into a ring, results are stored at the top of the NEXT
// step, and the barrier is a bare `s_waitcnt lgkmcnt(0); s_barrier` (hipcc's __syncthreads would
// drain vmcnt too).
#include "reservoir_impl.h"
#include <stdlib.h>

namespace {
using namespace sgp_res;

constexpr int kMaxLayers = 16;
template <int I> struct IntK { static constexpr int value = I; };

struct StackArgs {
    const float* x; long long xrs, xss;
    const float* wp;                 // packed weights of all layers (global workspace)
    float* out; long long ors, oss;
    float* h_state;                  // [L, N, R] or null
    float* tsum;                     // [n_tiles, T, L*R] per-tile column sums of the states, or null
    float alpha[kMaxLayers], oma[kMaxLayers];
    int act, T, N, F, R, L, ntw, n_tiles, debug;
};

// per-layer block of the packed weights (floats): bias [JT*16] | W_in | W_hh [JT][JT][64][4];
// W_in of layer 0 is the Wx fragment order of reservoir_impl.h ([JT][NKX][64] or
// [JT][NKX/4][64][4]), W_in of deeper layers the W_hh order applied to W_ih ([R, R]).
__host__ __device__ constexpr int win_floats(int JT, int NKX) {
    return JT * NKX * 64 > JT * JT * 256 ? JT * NKX * 64 : JT * JT * 256;
}
__host__ __device__ constexpr int layer_floats(int JT, int NKX) {
    return JT * 16 + win_floats(JT, NKX) + JT * JT * 256;
}

struct StackPtrs { const float* w_ih[kMaxLayers]; const float* w_hh[kMaxLayers]; const float* b[kMaxLayers]; };

__global__ void pack_stack(StackPtrs ptr, float* __restrict__ out, int F, int R, int JT, int NKX, int L) {
    const float* const* w_ih = ptr.w_ih;
    const float* const* w_hh = ptr.w_hh;
    const float* const* b = ptr.b;
    const int LB = layer_floats(JT, NKX), WIN = win_floats(JT, NKX);
    const long long total = (long long)L * LB;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int l = (int)(i / LB);
        const int o0 = (int)(i % LB);
        float v = 0.f;
        if (o0 < JT * 16) {
            v = o0 < R ? b[l][o0] : 0.f;
        } else if (o0 < JT * 16 + WIN) {
            const int o = o0 - JT * 16;
            if (l == 0) {
                if (o < JT * NKX * 64) {
                    int ln, ks, jt;
                    if (NKX % 4 == 0) {              // [JT][NKX/4][64][4]
                        const int s = o & 3;
                        ln = (o >> 2) & 63;
                        const int k4 = (o >> 8) % (NKX / 4);
                        jt = (o >> 8) / (NKX / 4);
                        ks = 4 * k4 + s;
                    } else {                         // [JT][NKX][64]
                        ln = o & 63;
                        ks = (o >> 6) % NKX;
                        jt = (o >> 6) / NKX;
                    }
                    const int j = 16 * jt + (ln & 15);
                    const int k = (ln >> 4) * NKX + ks;
                    v = (j < R && k < F) ? w_ih[0][(long long)j * F + k] : 0.f;
                }
            } else if (o < JT * JT * 256) {
                const int s = o & 3, ln = (o >> 2) & 63;
                const int kb = (o >> 8) % JT, jt = (o >> 8) / JT;
                const int j = 16 * jt + (ln & 15), k = 16 * kb + 4 * (ln >> 4) + s;
                v = (j < R && k < R) ? w_ih[l][(long long)j * R + k] : 0.f;
            }
        } else {
            const int o = o0 - JT * 16 - WIN;
            const int s = o & 3, ln = (o >> 2) & 63;
            const int kb = (o >> 8) % JT, jt = (o >> 8) / JT;
            const int j = 16 * jt + (ln & 15), k = 16 * kb + 4 * (ln >> 4) + s;
            v = (j < R && k < R) ? w_hh[l][(long long)j * R + k] : 0.f;
        }
        out[i] = v;
    }
}

// OVEC: 16-byte state stores (strides / pointers checked on the host) -- a compile-time switch: with
// both store flavours in the loop body hipcc's s_waitcnt bookkeeping drains vmcnt(0) every step.
// sums over the 16 lanes of a DPP row (= the 16 nodes of a tile that share q) of four values at once:
// quad butterfly, then the two mirrors; every lane ends up with its row's totals.  v_add_f32_dpp from
// inline asm (hipcc emits v_mov_b32_dpp + v_add_f32: twice the VALU instructions, and VALU time is
// matrix-pipe time in this kernel); a DPP source written by the previous VALU instruction needs two
// wait states: the four independent chains are interleaved, three instructions lie in between.
__device__ __forceinline__ void row16_sum4(f32x4& v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

// CSUM: also write the column sums of every tile's states (global_attr block of
// lib/nn/encoders/sgp_spatial_encoder.py:32-34: the mean over nodes of the tensor the reservoir just
// produced) -- the rows are in registers here; a separate pass re-reads the whole block from HBM.
template <int JT, int NKX, bool OVEC, bool CSUM>
__global__ __launch_bounds__(1024) void reservoir_stack(StackArgs a) {
    constexpr int WIN = win_floats(JT, NKX);
    constexpr int LB = layer_floats(JT, NKX);
    constexpr int PFD = 8;                               // input-row ring of layer 0 (time steps)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int L = a.L;
    {
        const int total4 = L * LB / 4;
        for (int i = threadIdx.x; i < total4; i += blockDim.x)
            reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(a.wp)[i];
    }
    const int lane = threadIdx.x & 63;
    const int n_in = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = wave % L, ts = wave / L;
    const int tile = blockIdx.x * a.ntw + ts;
    const int node = tile * 16 + n_in;
    const bool ok = tile < a.n_tiles && node < a.N;
    // hand-off slabs: [tile slot][layer][2][JT][64] f32x4
    f32x4* ring = reinterpret_cast<f32x4*>(lds + L * LB) + (long long)(ts * L) * 2 * JT * 64;
    f32x4* ring_out = ring + (long long)l * 2 * JT * 64;
    const f32x4* ring_in = ring + (long long)(l > 0 ? l - 1 : 0) * 2 * JT * 64;
    // input rows of layer 0: [tile slot][PFD][NKX][64] floats, filled by LDS-DMA PFD - 1 steps ahead
    float* xring = lds + L * LB + (long long)a.ntw * L * 2 * JT * 64 * 4 + (long long)ts * PFD * NKX * 64;
    const unsigned xring_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) float*)xring);

    const float* bias = lds + l * LB;
    const float* wx = bias + JT * 16;
    const float* wh = wx + WIN;
    const float al = a.alpha[l], om = a.oma[l];

    f32x4 h[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        float hv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.h_state && ok) {
            const int j0 = 16 * jt + 4 * q;
            const float* hp = a.h_state + ((long long)l * a.N + node) * a.R + j0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (j0 + r < a.R) hv[r] = hp[r];
        }
        h[jt] = f32x4{hv[0], hv[1], hv[2], hv[3]};
    }
    float* const out_l = a.out + (long long)node * a.ors + (long long)l * a.R;
    const bool tail_tile = tile * 16 + 16 > a.N;         // wave-uniform
    bool st_ok[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) st_ok[jt] = ok && 16 * jt + 4 * q < a.R && !(a.debug & 1);
    auto store_h = [&](int t) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int j0 = 16 * jt + 4 * q;
            if (st_ok[jt]) {
                float* op = out_l + (long long)t * a.oss + j0;
                if constexpr (OVEC) {
                    *reinterpret_cast<f32x4*>(op) = h[jt];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (j0 + r < a.R) op[r] = h[jt][r];
                }
            }
            if constexpr (CSUM) {
                f32x4 v = h[jt];
                if (tail_tile && !ok) v = f32x4{0.f, 0.f, 0.f, 0.f};      // rows past N (last tile) do not count
                row16_sum4(v);
                if (n_in == 0 && tile < a.n_tiles && j0 < a.R) {
                    float* sp = a.tsum + ((long long)tile * a.T + t) * ((long long)a.L * a.R) + (long long)l * a.R + j0;
                    if constexpr (OVEC) {
                        *reinterpret_cast<f32x4*>(sp) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (j0 + r < a.R) sp[r] = v[r];
                    }
                }
            }
        }
    };
    // input rows are requested UNCONDITIONALLY from a clamped address (no branch around a load: a
    // conditional load makes the compiler wait for it at the join) and masked where they are used
    bool x_ok[NKX];
    long long x_off[NKX];
    {
        const int nodec = min(node, a.N - 1);
#pragma unroll
        for (int ks = 0; ks < NKX; ++ks) {
            x_ok[ks] = ok && q * NKX + ks < a.F;
            x_off[ks] = (long long)nodec * a.xrs + min(q * NKX + ks, a.F - 1);
        }
    }
    // row of step t -> ring slot t mod PFD.  One global_load_lds_dword per k-step: lane (n, q)
    // fetches x[node n][q NKX + ks] to slot base + 256 ks + 4 lane.  Issued from inline asm and
    // waited for with a hand-counted vmcnt: the compiler's own bookkeeping drains vmcnt(0) every
    // step once loads and (conditional) stores are both pending in the loop.
    auto dma_x = [&](int t) {
        const float* xp = a.x + (long long)min(t, a.T - 1) * a.xss;
        const unsigned base = xring_lds + (unsigned)((t % PFD) * NKX * 256);
#pragma unroll
        for (int ks = 0; ks < NKX; ++ks) {
            const float* src = xp + x_off[ks];
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                         :: "v"(src), "s"(base + (unsigned)ks * 256u) : "memory");
        }
    };
    __syncthreads();                                     // weights in LDS
    // retire the set-up loads with a wait the compiler can see (otherwise it carries "may still be
    // in flight" into the loop and guards every step with vmcnt(0)); layer 0's first rows are
    // re-requested after it
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
    if (l == 0) {
        for (int p = 0; p < PFD - 1; ++p) dma_x(p);
    }

    // narrow layers keep their weight fragments in registers for the whole sequence (the barrier's
    // memory clobber keeps the compiler from hoisting the LDS reads itself: three dependent
    // ds_read + wait rounds per step otherwise); wide ones (JT = 4: 128 VGPRs of weights) read
    // them from LDS every step
    constexpr bool WREG = JT <= 2;
    f32x4 rb[JT], rwh[JT][JT], rwx[JT][JT];
    float rw0[JT][NKX];
    if constexpr (WREG) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            rb[jt] = *reinterpret_cast<const f32x4*>(bias + jt * 16 + q * 4);
#pragma unroll
            for (int kb = 0; kb < JT; ++kb) {
                rwh[jt][kb] = *reinterpret_cast<const f32x4*>(wh + ((jt * JT + kb) * 64 + lane) * 4);
                rwx[jt][kb] = *reinterpret_cast<const f32x4*>(wx + ((jt * JT + kb) * 64 + lane) * 4);
            }
#pragma unroll
            for (int ks = 0; ks < NKX; ++ks) {
                if constexpr (NKX % 4 == 0)
                    rw0[jt][ks] = wx[((jt * (NKX / 4) + ks / 4) * 64 + lane) * 4 + (ks & 3)];
                else
                    rw0[jt][ks] = wx[(jt * NKX + ks) * 64 + lane];
            }
        }
    }

    const int n_iter = a.T + L - 1;
    for (int i = 0; i < n_iter; ++i) {
        const int t = i - l;
        if (t >= 0 && t < a.T) {                         // wave-uniform
            // wide layers: keep the fragments in LDS (the compiler would hoist 128 VGPRs of
            // weights out of the time loop); narrow ones are welcome to stay in registers
            int wo = 0;
            if constexpr (JT >= 4) asm volatile("" : "+v"(wo));
            const float* bias_t = bias + wo;
            const float* wx_t = wx + wo;
            const float* wh_t = wh + wo;
            if (t > 0) store_h(t - 1);                   // h still holds the state of step t-1
            f32x4 hin[JT];
            if (l > 0) {
                const f32x4* src = ring_in + ((i - 1) & 1) * JT * 64;
#pragma unroll
                for (int kb = 0; kb < JT; ++kb) hin[kb] = src[kb * 64 + lane];
            } else {
                // request row t + PFD - 1 (into the slot consumed one step ago), then wait until at
                // most the (PFD - 1) NKX youngest memory operations are outstanding: row t is older
                // than that many requests (stores in between only make the wait stricter)
                dma_x(t + PFD - 1);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PFD - 1) * NKX < 63 ? (PFD - 1) * NKX : 63) : "memory");
            }
            f32x4 acc[JT];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                if constexpr (WREG) acc[jt] = rb[jt];
                else acc[jt] = *reinterpret_cast<const f32x4*>(bias_t + jt * 16 + q * 4);
            }
            // recurrent part first: its operands are this wave's registers
#pragma unroll
            for (int kb = 0; kb < JT; ++kb) {
                f32x4 wf[JT];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    if constexpr (WREG) wf[jt] = rwh[jt][kb];
                    else wf[jt] = *reinterpret_cast<const f32x4*>(wh_t + ((jt * JT + kb) * 64 + lane) * 4);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt)
                        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[jt][s], h[kb][s], acc[jt], 0, 0, 0);
            }
            // input part
            const float* xrow = xring + (t % PFD) * NKX * 64;
            if (l > 0) {
#pragma unroll
                for (int kb = 0; kb < JT; ++kb) {
                    f32x4 wf[JT];
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        if constexpr (WREG) wf[jt] = rwx[jt][kb];
                        else wf[jt] = *reinterpret_cast<const f32x4*>(wx_t + ((jt * JT + kb) * 64 + lane) * 4);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt)
                            acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[jt][s], hin[kb][s], acc[jt], 0, 0, 0);
                }
            } else if constexpr (NKX % 4 == 0) {
#pragma unroll
                for (int k4 = 0; k4 < NKX / 4; ++k4)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        f32x4 wv;
                        if constexpr (WREG) wv = f32x4{rw0[jt][4 * k4], rw0[jt][4 * k4 + 1], rw0[jt][4 * k4 + 2], rw0[jt][4 * k4 + 3]};
                        else wv = *reinterpret_cast<const f32x4*>(wx_t + ((jt * (NKX / 4) + k4) * 64 + lane) * 4);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s], x_ok[4 * k4 + s] ? xrow[(4 * k4 + s) * 64 + lane] : 0.f, acc[jt], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int ks = 0; ks < NKX; ++ks)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt)
                        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(WREG ? rw0[jt][ks] : wx_t[(jt * NKX + ks) * 64 + lane],
                                                                       x_ok[ks] ? xrow[ks * 64 + lane] : 0.f, acc[jt], 0, 0, 0);
            }
            if (a.act == SGP_ACT_TANH) {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[jt][r] = tanh_r(acc[jt][r]);
            } else if (a.act == SGP_ACT_RELU) {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[jt][r] = fmaxf(acc[jt][r], 0.f);
            } else if (a.act == SGP_ACT_TANH_REL) {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[jt][r] = tanh_rel(acc[jt][r]);
            } else if (a.act == SGP_ACT_SELF_NORM) {
                float ss = 0.f;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ss = fmaf(acc[jt][r], acc[jt][r], ss);
                ss += __shfl_xor(ss, 16);
                ss += __shfl_xor(ss, 32);
                const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize(eps=1e-12)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[jt][r] *= inv;
            }
            // leak, publish for the next layer (stored to HBM at the top of the next step)
            f32x4* dst = ring_out + (i & 1) * JT * 64;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    h[jt][r] = a.act == SGP_ACT_TANH ? leak_tanh_r(h[jt][r], acc[jt][r], al, om) : leak(h[jt][r], acc[jt][r], al, om);
                if (l + 1 < L) dst[jt * 64 + lane] = h[jt];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (a.T > 0) store_h(a.T - 1);
    if (a.h_state) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int j0 = 16 * jt + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ok && j0 + r < a.R) a.h_state[((long long)l * a.N + node) * a.R + j0 + r] = h[jt][r];
        }
    }
}

template <int JT, int NKX>
long long stack_lds_bytes(int L, int ntw) {
    return ((long long)L * layer_floats(JT, NKX) + (long long)ntw * L * 2 * JT * 64 * 4 +
            (long long)ntw * 8 * NKX * 64) * 4;
}

template <int JT, int NKX>
int launch_stack(StackArgs a, hipStream_t s) {
    a.n_tiles = (a.N + 15) / 16;
    // node tiles per workgroup: as many as 16 waves hold once there are more tiles than the chip
    // has room for side by side; small graphs spread one tile per workgroup over the CUs
    int ntw = 16 / a.L;
    if (ntw < 1) ntw = 1;
    while (ntw > 1 && (a.n_tiles + ntw - 1) / ntw < 512) ntw >>= 1;
    while (ntw > 1 && stack_lds_bytes<JT, NKX>(a.L, ntw) > kLdsLimit) ntw >>= 1;
    if (stack_lds_bytes<JT, NKX>(a.L, ntw) > kLdsLimit)
        return sgp::fail(SGP_EUNSUP, "sgp_reservoir_fused_f32: %d layers of %d units exceed the LDS", a.L, a.R);
    a.ntw = ntw;
    const bool ov = (a.R % 4 == 0) && (a.ors % 4 == 0) && (a.oss % 4 == 0) && sgp::aligned16(a.out) &&
                    (!a.tsum || sgp::aligned16(a.tsum));
    auto kern = a.tsum ? (ov ? reservoir_stack<JT, NKX, true, true> : reservoir_stack<JT, NKX, false, true>)
                       : (ov ? reservoir_stack<JT, NKX, true, false> : reservoir_stack<JT, NKX, false, false>);
    const int bytes = (int)stack_lds_bytes<JT, NKX>(a.L, ntw);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return sgp::fail((int)e, "reservoir_stack: LDS opt-in: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3((a.n_tiles + ntw - 1) / ntw), dim3(64 * a.L * ntw), (size_t)bytes, s, a);
    return sgp::check_launch("reservoir_stack");
}

template <int JT>
int launch_stack_nkx(const StackArgs& a, int nkx, hipStream_t s) {
    switch (nkx) {
        case 1: return launch_stack<JT, 1>(a, s);
        case 2: return launch_stack<JT, 2>(a, s);
        case 4: return launch_stack<JT, 4>(a, s);
        case 8: return launch_stack<JT, 8>(a, s);
        case 16: return launch_stack<JT, 16>(a, s);
    }
    return sgp::fail(SGP_EUNSUP, "sgp_reservoir_fused_f32: input size not supported");
}

int stack_nkx(int F) {
    const int need = (F + 3) / 4;
    const int opts[] = {1, 2, 4, 8, 16};
    for (int o : opts) if (o >= need) return o;
    return 0;
}
int stack_jt(int R) { return R <= 16 ? 1 : (R <= 32 ? 2 : (R <= 64 ? 4 : 0)); }

}  // namespace

extern "C" {

int64_t sgp_reservoir_fused_workspace_bytes(int32_t F, int32_t R, int32_t L) {
    const int jt = stack_jt(R), nkx = stack_nkx(F);
    if (!jt || !nkx || L < 1 || L > kMaxLayers) return -1;
    return (int64_t)L * layer_floats(jt, nkx) * 4;
}

int32_t sgp_reservoir_fused_supported(int32_t F, int32_t R, int32_t L) {
    const int jt = stack_jt(R), nkx = stack_nkx(F);
    if (!jt || !nkx || L < 2 || L > kMaxLayers) return 0;
    const long long lds = ((long long)L * layer_floats(jt, nkx) + (long long)L * 2 * jt * 64 * 4 + 8ll * nkx * 64) * 4;
    return lds <= kLdsLimit ? 1 : 0;
}

int sgp_reservoir_fused_f32(const float* x, int64_t xrs, int64_t xss,
                            const float* const* w_ih, const float* const* w_hh, const float* const* b,
                            const double* alpha, int32_t act,
                            float* out, int64_t ors, int64_t oss,
                            float* h_state, void* workspace,
                            int32_t T, int32_t N, int32_t F, int32_t R, int32_t L,
                            sgp_stream_t stream) {
    return sgp_reservoir_fused_sums_f32(x, xrs, xss, w_ih, w_hh, b, alpha, act, out, ors, oss, h_state,
                                        workspace, nullptr, T, N, F, R, L, stream);
}

int sgp_reservoir_fused_sums_f32(const float* x, int64_t xrs, int64_t xss,
                                 const float* const* w_ih, const float* const* w_hh, const float* const* b,
                                 const double* alpha, int32_t act,
                                 float* out, int64_t ors, int64_t oss,
                                 float* h_state, void* workspace, float* tile_sums,
                                 int32_t T, int32_t N, int32_t F, int32_t R, int32_t L,
                                 sgp_stream_t stream) {
    SGP_REQUIRE(x && w_ih && w_hh && b && alpha && out && workspace, "sgp_reservoir_fused_f32: null pointer");
    SGP_REQUIRE(T >= 0 && N >= 0 && F > 0 && R > 0 && L >= 1, "sgp_reservoir_fused_f32: bad size");
    SGP_REQUIRE(act >= SGP_ACT_TANH && act <= SGP_ACT_TANH_REL, "sgp_reservoir_fused_f32: unknown activation %d", act);
    SGP_REQUIRE(sgp::aligned16(workspace), "sgp_reservoir_fused_f32: workspace must be 16-byte aligned");
    for (int l = 0; l < L && l < kMaxLayers; ++l)
        SGP_REQUIRE(w_ih[l] && w_hh[l] && b[l], "sgp_reservoir_fused_f32: null weight pointer (layer %d)", l);
    if (T == 0 || N == 0) return 0;
    const int jt = stack_jt(R), nkx = stack_nkx(F);
    if (!jt || !nkx || L > kMaxLayers)
        return sgp::fail(SGP_EUNSUP, "sgp_reservoir_fused_f32: F=%d R=%d L=%d outside the fused kernel "
                         "(F <= 64, R <= 64, L <= %d): run sgp_reservoir_f32 per layer", F, R, L, kMaxLayers);
    hipStream_t s = (hipStream_t)stream;
    const long long wfloats = (long long)L * layer_floats(jt, nkx);
    StackPtrs ptr = {};
    for (int l = 0; l < L; ++l) { ptr.w_ih[l] = w_ih[l]; ptr.w_hh[l] = w_hh[l]; ptr.b[l] = b[l]; }
    int pg = (int)((wfloats + 255) / 256);
    if (pg > 1024) pg = 1024;
    hipLaunchKernelGGL(pack_stack, dim3(pg), dim3(256), 0, s, ptr, (float*)workspace, F, R, jt, nkx, L);
    int rc = sgp::check_launch("pack_stack");
    if (rc) return rc;

    StackArgs a;
    a.x = x; a.xrs = xrs; a.xss = xss;
    a.wp = (const float*)workspace;
    a.out = out; a.ors = ors; a.oss = oss;
    a.h_state = h_state;
    a.tsum = tile_sums;
    for (int l = 0; l < kMaxLayers; ++l) {
        const double al = l < L ? alpha[l] : 0.0;
        a.alpha[l] = (float)al;                    // scalars rounded to fp32 like torch does for
        a.oma[l] = (float)(1.0 - al);              // `(1 - alpha) * h` (reservoir.py:80)
    }
    a.act = act; a.T = T; a.N = N; a.F = F; a.R = R; a.L = L; a.ntw = 1; a.n_tiles = 0;
    a.debug = (int)sgp::tune("stack_debug", 0);   // bit 0: skip the result stores (timing ablation)
    switch (jt) {
        case 1: return launch_stack_nkx<1>(a, nkx, s);
        case 2: return launch_stack_nkx<2>(a, nkx, s);
        case 4: return launch_stack_nkx<4>(a, nkx, s);
    }
    return sgp::fail(SGP_EUNSUP, "sgp_reservoir_fused_f32: unreachable");
}

}  // extern "C"
