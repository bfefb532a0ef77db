// Small-N leaky echo-state layer on the 16-bit matrix cores with fp32-grade products (included by reservoir_impl.h
// inside namespace sgp_res behind reservoir_bf3.h; reference: lib/nn/reservoir/reservoir.py:77-81 stepped by :170-183 at
// the traffic configs' shapes, config/traffic/sgp_la.yaml R = 64 / sgp_bay.yaml R = 128, N = 207 / 325, F = 3).
//
// With a few hundred nodes the layer is one serial chain of T steps on a dozen node tiles.  reservoir_layer_splitj cuts
// a step's output tiles over the 4 waves of a workgroup (one per SIMD) and is then bound by the fp32 matrix pipe inside
// every SIMD: R = 128 needs 66 v_mfma_f32_16x16x4_f32 of 32 cycles per wave and step (2112 of the step's ~3400
// cycles; BENCH r4: C2 spends 86 of 92 ms here, C1 21 of 22).  This kernel keeps that structure -- one node tile per
// workgroup, JT / 4 output tiles per wave, the new state exchanged through a double-buffered LDS slab, one barrier per
// step, input rows by LDS-DMA into a ring, stores at the top of the next step -- and forms the products as in
// reservoir_bf3.h: every fp32 operand as THREE bf16 pieces (24 bits, no scale, any activation), six piece products of
// order <= 2^-16 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: R = 128: 60 MFMAs of 16 cycles per wave and step.
//
// What travels through the slab are the PIECES: a wave splits the 4 new values a lane holds of each of its tiles
// (bf3_split2 twice) and writes them where the consumers' B operand of k-block p = tile / 2 expects them (8 bytes per
// lane and piece at [piece][p][lane][tile & 1]: one ds_read_b128 per k-block and piece on the reading side); the fp32
// state of a wave's OWN tiles stays in its registers for the leak.  The weight fragments are those of
// pack_weights_bf3 (k order of the recurrent blocks = the accumulator layout, reservoir_bf3.h); a wave keeps the fragments
// of its own output tiles in REGISTERS for the whole sequence (one wave per SIMD: 120 of 512 registers at R = 128).
// The accumulation of a tile is cut into independent chains (added at the end) with the chain index as the inner
// loop, so that no MFMA waits for the result of the one before it.
//
// Widths: R <= 16 JT, F <= 4 NKX with NKX <= 8 (one input k-block) or NKX = 16 (two), any N, T; padded units / features carry zero
// weights and zero operands, stores are masked like the fp32 kernel's.

//
// H16 (tanh, state within [-1, 1]): the RECURRENT products from TWO fp16 pieces per operand instead of three bf16 ones --
// three products (lo hi, hi lo, hi hi) on v_mfma_f32_16x16x32_f16 instead of six, two pieces to cut, publish and read
// instead of three (R = 128: 36 MFMAs per wave and step instead of 60).  The state is scaled by 2^14 (|h| <= 1: the
// leak is a convex combination of the old state and a tanh), row j of W_hh by the power of two 2^e_j that puts its
// largest entry at 2^13 .. 2^14 (pack_weights_sj16), the sum of a row is scaled back by 2^(-e_j - 14) -- all exact.
// Error of one operand (spmm_split.hip's model): relative 2^-23 down to 2^-16 of the bound, absolute 2^-38 of the bound
// below it (the low piece is an fp16 subnormal there); the dropped lo lo product is of order 2^-22.  The input block
// keeps its three bf16 pieces (x is not bounded).  A workgroup whose INITIAL state leaves [-1, 1] (a caller's own
// h_state) runs the three-piece loop instead: the choice is made per workgroup at the top of the kernel.

__host__ __device__ constexpr bool sjbf3_supported(int JT, int NKX) { return (JT == 4 || JT == 8) && (NKX <= 8 || NKX == 16); }
// fp16 fragments of the recurrent blocks: [JT x 16 row scales 2^(-e_j - 14)] [JT][KBH][2 pieces][64 lanes][16 B]
__host__ __device__ constexpr long long sj16_packed_bytes(int JT) { return JT * 64ll + (long long)JT * bf3_kbh(JT) * 2 * 1024; }
__host__ __device__ constexpr int sj16_frag_off(int KBH, int jt, int kb, int pc) { return ((jt * KBH + kb) * 2 + pc) * 1024; }
// LDS: piece slab [2][3][KBH][64][16 B] | self_norm partials [2][64] | input ring [PFD][NKX][64]
__host__ __device__ constexpr long long sjbf3_slab_bytes(int JT) { return 2ll * 3 * bf3_kbh(JT) * 1024; }
__host__ __device__ constexpr int sjbf3_ring(int JT, int NKX) { return 8; }
__host__ __device__ constexpr long long sjbf3_lds_bytes(int JT, int NKX) {
    return sjbf3_slab_bytes(JT) + 2 * 64 * 4 + (long long)sjbf3_ring(JT, NKX) * NKX * 256;
}

// experiment switches (build with -DSGP_SJ_ABL=bits): 1 no result stores, 2 wave 0's row wait counts its stores too,
// 4 no row wait at all (wrong results), 8 no activation, 16 no piece exchange (wrong results), 32 no MFMAs,
// 128 no row requests inside the time loop (wrong results); -DSGP_SJ_NCHN=n: accumulation chains per tile
#ifndef SGP_SJ_ABL
#define SGP_SJ_ABL 0
#endif
#ifndef SGP_SJ_TILEWISE
#define SGP_SJ_TILEWISE 1
#endif
constexpr bool sj_abl(int bit) { return (SGP_SJ_ABL & bit) != 0; }

// ACT >= 0: the activation is known at compile time (the tanh instance carries no activation dispatch in its time loop:
// the run-time form spent ~40 scalar branches per step on it), -1: read from the arguments
template <int JT, int NKX, bool OVEC, int ACT, bool H16>
__device__ __forceinline__ void splitj_bf3_body(const ResArgs& a) {
    static_assert(sjbf3_supported(JT, NKX), "R = 64 / 128 (padded), one or two input k-blocks");
    const int act = ACT >= 0 ? ACT : a.act;
    float alpha_v = a.alpha;              // a VGPR copy for the leak: hipcc 7.2 emitted v_fma_f32 with BOTH scalars (alpha, 1 - alpha)
    asm("" : "+v"(alpha_v));              // as operands in this kernel ("violates constant bus restriction")
    constexpr int JW = JT / 4;                           // output tiles per wave
    constexpr int KBH = bf3_kbh(JT), KBX = bf3_kbx(NKX), KB = KBH + KBX;   // recurrent k-blocks (two state tiles each) + the input block(s)
    constexpr int PFD = sjbf3_ring(JT, NKX);
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    char* slab = lds_raw;                                                  // [2][3][KBH][64][16]
    float* red_base = reinterpret_cast<float*>(slab + sjbf3_slab_bytes(JT));
    float* xring = red_base + 2 * 64;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_in = lane & 15, q = lane >> 4;
    const int node = blockIdx.x * 16 + n_in;
    const bool ok = node < a.N;

    // fp32 state of this wave's own tiles (leak), zero in padded units
    f32x4 hown[JW];
#pragma unroll
    for (int w = 0; w < JW; ++w) {
        const int j0 = 16 * (wave * JW + w) + 4 * q;
        float hv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.h_state && ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (j0 + r < a.R) hv[r] = a.h_state[(long long)node * a.R + j0 + r];
        }
        hown[w] = f32x4{hv[0], hv[1], hv[2], hv[3]};
    }
    // pieces of a tile's 4 values -> the slab of parity `par`
    float hscale = kSj16StateScale;       // (a register: v_fma_mix takes no literal)
    asm("" : "+v"(hscale));
    auto publish = [&](int par, int w, const f32x4 hv) {
        const int jt = wave * JW + w;
        char* base = slab + (size_t)par * (3 * KBH * 1024) + (size_t)(jt >> 1) * 1024 + lane * 16 + (jt & 1) * 8;
        if constexpr (H16) {
            unsigned a1, a2, b1, b2;
            sj16_split2(hv[0], hv[1], hscale, a1, a2);
            sj16_split2(hv[2], hv[3], hscale, b1, b2);
            *reinterpret_cast<uint2*>(base) = uint2{a1, b1};
            *reinterpret_cast<uint2*>(base + KBH * 1024) = uint2{a2, b2};
        } else {
            unsigned a1, a2, a3, b1, b2, b3;
            bf3_split2(hv[0], hv[1], a1, a2, a3);
            bf3_split2(hv[2], hv[3], b1, b2, b3);
            *reinterpret_cast<uint2*>(base) = uint2{a1, b1};
            *reinterpret_cast<uint2*>(base + KBH * 1024) = uint2{a2, b2};
            *reinterpret_cast<uint2*>(base + 2 * KBH * 1024) = uint2{a3, b3};
        }
    };

    // input rows: lane (n, q) register ks <-> feature bf3_feature(NKX, q, ks), as pack_weights_bf3 orders the input block
    bool x_ok[NKX];
    long long x_off[NKX];
    {
        const int nodec = min(node, a.N - 1);
#pragma unroll
        for (int ks = 0; ks < NKX; ++ks) {
            const int f = bf3_feature(NKX, q, ks);
            x_ok[ks] = ok && f < a.F;
            x_off[ks] = (long long)nodec * a.xrs + min(f, a.F - 1);
        }
    }
    const unsigned xring_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) float*)xring);
    // running request state of wave 0: the rows of steps 0, 1, 2, .. in turn (the last step's row again beyond it); one
    // wave per SIMD pays ~5 cycles for every instruction, so the 64-bit products t x stride are replaced by additions
    const float* xsrc[NKX];
#pragma unroll
    for (int ks = 0; ks < NKX; ++ks) xsrc[ks] = a.x + x_off[ks];
    int x_next = 0;                                       // step whose row is requested next
    unsigned x_slot = 0;                                  // its ring slot (x_next % PFD)
    auto dma_next = [&]() {
        const unsigned base = xring_lds + x_slot * (unsigned)(NKX * 256);
        const long long inc = x_next < a.T - 1 ? a.xss : 0;
#pragma unroll
        for (int ks = 0; ks < NKX; ++ks) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                         :: "v"(xsrc[ks]), "s"(base + (unsigned)ks * 256u) : "memory");
            xsrc[ks] += inc;
        }
        ++x_next;
        x_slot = x_slot + 1 == (unsigned)PFD ? 0u : x_slot + 1;
    };
    bool st_ok[JW];
#pragma unroll
    for (int w = 0; w < JW; ++w) st_ok[w] = !a.no_store && ok && 16 * (wave * JW + w) + 4 * q < a.R;
    float* orow = a.out + (long long)node * a.ors + 16 * (wave * JW) + 4 * q;       // this lane's piece of step 0's row
    // OVEC: every lane stores 16 bytes per tile and step, unconditionally -- lanes without a row (or a padded unit) into a
    // dump area behind the packed weights -- through a running pointer: no exec-masked branch, no 64-bit product per step
    float* optr[JW];
    long long oinc[JW];
#pragma unroll
    for (int w = 0; w < JW; ++w) {
        optr[w] = st_ok[w] ? orow + 16 * w : a.dump + lane * 4;
        oinc[w] = st_ok[w] ? a.oss : 0;
    }
    auto store_h = [&](int t, const f32x4 (&hv)[JW]) {
        if constexpr (OVEC) {
#pragma unroll
            for (int w = 0; w < JW; ++w) {
                *reinterpret_cast<f32x4*>(optr[w]) = hv[w];
                optr[w] += oinc[w];
            }
        } else {
            float* ot = orow + (long long)t * a.oss;
#pragma unroll
            for (int w = 0; w < JW; ++w) {
                const int j0 = 16 * (wave * JW + w) + 4 * q;
                if (st_ok[w]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (j0 + r < a.R) ot[16 * w + r] = hv[w][r];
                }
            }
        }
    };
    // this wave's weight fragments and bias: ONE wave per SIMD owns the whole register file, so the 3 KB x KB x 3 pieces
    // of its JW output tiles stay resident for all T steps (the fp32 form re-reads them from LDS every step)
    constexpr int NP = H16 ? 2 : 3;                      // pieces per recurrent operand
    u32x4 W[JW][KB][3];                                  // (H16: recurrent blocks [..][kb < KBH][0 .. 1] are fp16 fragments)
    f32x4 bias[JW], rsc[JW];                             // rsc (H16): 2^(-e_j - 14) of this lane's 4 rows per tile
    {
        const char* wp = reinterpret_cast<const char*>(a.wp_bf3);
        const char* wh = reinterpret_cast<const char*>(a.wp_h16);
#pragma unroll
        for (int w = 0; w < JW; ++w) {
            bias[w] = *reinterpret_cast<const f32x4*>(wp + ((wave * JW + w) * 16 + q * 4) * 4);
            if constexpr (H16) rsc[w] = *reinterpret_cast<const f32x4*>(wh + ((wave * JW + w) * 16 + q * 4) * 4);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int pc = 0; pc < (kb < KBH ? NP : 3); ++pc)
                    W[w][kb][pc] = H16 && kb < KBH
                        ? *reinterpret_cast<const u32x4*>(wh + JT * 64 + sj16_frag_off(KBH, wave * JW + w, kb, pc) + lane * 16)
                        : *reinterpret_cast<const u32x4*>(wp + JT * 64 + bf3_frag_off(KB, wave * JW + w, kb, pc) + lane * 16);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0), visible to the compiler: state and weights have landed
#pragma unroll
    for (int w = 0; w < JW; ++w) publish(1, w, hown[w]);  // step 0 reads parity 1
    __syncthreads();                                     // initial pieces in LDS
    if (wave == 0) {
        for (int p = 0; p < PFD - 1; ++p) dma_next();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // rows 0 .. PFD-2 published

    // pieces of the input row of step 0 (rows 0 .. PFD - 2 are published)
    u32x4 X[KBX][3];
    auto input_pieces = [&](int t) {
        const float* xrow = xring + (t % PFD) * NKX * 64;
#pragma unroll
        for (int kx = 0; kx < KBX; ++kx) {
            float xv[8];
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const int ks = 8 * kx + s8;
                xv[s8] = ks < NKX && x_ok[ks < NKX ? ks : 0] ? xrow[(ks < NKX ? ks : 0) * 64 + lane] : 0.f;
            }
            bf3_split8(xv, X[kx][0], X[kx][1], X[kx][2]);
        }
    };
    input_pieces(0);
    // The input block does not depend on the state: its six products of step t + 1 are issued at the END of step t, behind
    // the publication of the new state and in front of the barrier, where the wave would otherwise wait for its LDS writes
    // and the other waves (two chains of three per tile: the second half's start does not wait for the first's result).
    constexpr int PW[6] = {2, 1, 0, 1, 0, 0}, PV[6] = {0, 1, 2, 0, 1, 0};         // W3 V1, W2 V2, W1 V3, W2 V1, W1 V2, W1 V1
    constexpr int PW16[3] = {1, 0, 0}, PV16[3] = {0, 1, 0};                        // Wlo Vhi, Whi Vlo, Whi Vhi
    f32x4 accx[JW][2];
    auto input_block = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kx = 0; kx < KBX; ++kx)
#pragma unroll
                    for (int w = 0; w < JW; ++w) {
                        const f32x4 c0 = h == 0 ? bias[w] : f32x4{0.f, 0.f, 0.f, 0.f};
                        const bool first = i == 0 && kx == 0;
                        if (!sj_abl(32)) accx[w][h] = bf3_mfma(W[w][KBH + kx][PW[3 * h + i]], X[kx][PV[3 * h + i]], first ? c0 : accx[w][h]);
                        else if (first) accx[w][h] = c0;
                    }
    };
    input_block();

    for (int t = 0; t < a.T; ++t) {
        if (t > 0 && !sj_abl(1)) store_h(t - 1, hown);
        if (wave == 0 && !sj_abl(128)) dma_next();       // the row of step t + PFD - 1, into the slot consumed one step ago
        // B operands: the state pieces of step t - 1 (all tiles), leading pieces of every k-block first (the first round
        // of products needs exactly those)
        u32x4 V[KBH][3];
        {
            const char* sp = slab + (size_t)((t & 1) ^ 1) * (3 * KBH * 1024) + lane * 16;
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
#pragma unroll
                for (int p = 0; p < KBH; ++p)
                    V[p][pc] = *reinterpret_cast<const u32x4*>(sp + (pc * KBH + p) * 1024);
        }
        // One accumulation chain per (output tile, recurrent k-block), added at the end; the six piece products (smallest
        // first, reservoir_bf3.h) are the OUTER loop, so that consecutive MFMAs never wait for each other's result.
        // Measured (N = 325, R = 128 / N = 207, R = 64, us per step, with the input block still inside this loop): this
        // order 1.12 / 0.55; a k-block's six products back to back on one accumulator 1.41 / 0.66 (a dependent MFMA issues
        // ~29 cycles after the one it waits for, an independent one after 17); tile by tile (the tail of tile 0 under the
        // MFMAs of tile 1) 1.27 / 0.60; the six products of a k-block as two chains of three 1.23 / 0.65.
        // NCHN chains per tile, the k-blocks dealt to them in turn: a chain's next MFMA is NCHN x JW issues (>= 34 cycles) behind
        // the one it waits for (29), and the end needs NCHN - 1 adds per tile instead of one per k-block -- with one chain
        // per k-block the accumulators' zeroing, read-out (v_accvgpr) and adds were 100 of the step's 508 instructions.
#ifdef SGP_SJ_NCHN
        constexpr int NCHN = SGP_SJ_NCHN < KBH ? SGP_SJ_NCHN : KBH;
#else
        constexpr int NCHN = JW == 1 ? (KBH < 3 ? KBH : 3) : 2;
#endif
        f32x4 pre[JW];
        auto finish = [&](int w) {                       // leak, publish the pieces of tile w
            f32x4 hn;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                hn[r] = act == SGP_ACT_TANH ? leak_tanh_r(hown[w][r], pre[w][r], alpha_v, a.one_minus_alpha)
                                            : leak(hown[w][r], pre[w][r], alpha_v, a.one_minus_alpha);
            hown[w] = hn;
            if (!sj_abl(16)) publish(t & 1, w, hn);
        };
        // Tile by tile (SGP_SJ_TILEWISE, JW = 2): the sum, activation, leak and piece cut of tile 0 sit in the program
        // BEHIND the first products of tile 1, so that they issue in the gaps of tile 1's MFMAs instead of behind them.
        constexpr bool TILEWISE = SGP_SJ_TILEWISE && JW == 2;
        constexpr bool XEARLY = JW == 2;                  // (one tile per wave, R = 64: measured 0.51 against 0.48 us per step -- behind the publication there)
        f32x4 xsum[JW];                                   // this step's input products (issued during the last step)
#pragma unroll
        for (int w = 0; w < JW; ++w) xsum[w] = accx[w][0] + accx[w][1];
#pragma unroll
        for (int w0 = 0; w0 < (TILEWISE ? JW : 1); ++w0) {
            f32x4 acc[JW][NCHN];
#pragma unroll
            for (int i = 0; i < (H16 ? 3 : 6); ++i)
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb)
#pragma unroll
                    for (int w = TILEWISE ? w0 : 0; w < (TILEWISE ? w0 + 1 : JW); ++w) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        const bool first = i == 0 && kb < NCHN;        // (compile time: the loops are unrolled)
                        const f32x4 c = first ? zero : acc[w][kb % NCHN];
                        if (sj_abl(32)) { if (first) acc[w][kb % NCHN] = zero; }
                        else if constexpr (H16) acc[w][kb % NCHN] = sj16_mfma(W[w][kb][PW16[i]], V[kb][PV16[i]], c);
                        else acc[w][kb % NCHN] = bf3_mfma(W[w][kb][PW[i]], V[kb][PV[i]], c);
                    }
            // the input row of step t + 1 is visible since the last barrier (wave 0 retires rows two steps ahead): its
            // pieces are cut in the shadow of the MFMAs
            if (w0 == 0 && t + 1 < a.T) input_pieces(t + 1);
            // the input products of step t + 1 go out right behind the last recurrent products of this step: the tail of
            // the last tile (sum, activation, leak, piece cut) then issues in THEIR gaps instead of in front of them
            if (XEARLY && w0 == (TILEWISE ? JW : 1) - 1 && t + 1 < a.T) input_block();
#pragma unroll
            for (int w = TILEWISE ? w0 : 0; w < (TILEWISE ? w0 + 1 : JW); ++w) {
                if constexpr (H16) {
                    f32x4 rec = acc[w][0];
#pragma unroll
                    for (int c = 1; c < NCHN; ++c) rec += acc[w][c];
                    pre[w] = __builtin_elementwise_fma(rec, rsc[w], xsum[w]);
                } else {
                    pre[w] = xsum[w];
#pragma unroll
                    for (int c = 0; c < NCHN; ++c) pre[w] += acc[w][c];
                }
            }
            if (act == SGP_ACT_TANH && !sj_abl(8)) {
#pragma unroll
                for (int w = TILEWISE ? w0 : 0; w < (TILEWISE ? w0 + 1 : JW); ++w) pre[w] = tanh_r4(pre[w]);
            } else if (act == SGP_ACT_RELU) {
#pragma unroll
                for (int w = TILEWISE ? w0 : 0; w < (TILEWISE ? w0 + 1 : JW); ++w)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre[w][r] = fmaxf(pre[w][r], 0.f);
            } else if (act == SGP_ACT_TANH_REL) {
#pragma unroll
                for (int w = TILEWISE ? w0 : 0; w < (TILEWISE ? w0 + 1 : JW); ++w)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre[w][r] = tanh_rel(pre[w][r]);
            }
            if (act != SGP_ACT_SELF_NORM) {
#pragma unroll
                for (int w = TILEWISE ? w0 : 0; w < (TILEWISE ? w0 + 1 : JW); ++w) finish(w);
            }
        }
        if (act == SGP_ACT_SELF_NORM) {
            // norm over all R features: partial sums of the 4 waves meet in LDS
            float ss = 0.f;
#pragma unroll
            for (int w = 0; w < JW; ++w)
#pragma unroll
                for (int r = 0; r < 4; ++r) ss = fmaf(pre[w][r], pre[w][r], ss);
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            float* red = red_base + (t & 1) * 64;
            if (q == 0) red[wave * 16 + n_in] = ss;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const float tot = red[n_in] + red[16 + n_in] + red[32 + n_in] + red[48 + n_in];
            const float inv = 1.f / fmaxf(sqrtf(tot), 1e-12f);
#pragma unroll
            for (int w = 0; w < JW; ++w) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pre[w][r] *= inv;
                finish(w);
            }
        }
        if (!XEARLY && t + 1 < a.T) input_block();        // step t + 1's input products, in the shadow of the exchange
        // wave 0: the row of step t + 2 has landed once at most the (PFD - 3) NKX younger requests
        // (+ this step's stores, which only make the wait stricter) are outstanding
        if (wave == 0 && !sj_abl(4)) {
            constexpr int kOps = sj_abl(2) ? (PFD - 3) * (NKX + JW) : (PFD - 3) * NKX;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kOps < 63 ? kOps : 63) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (a.T > 0) store_h(a.T - 1, hown);
    if (a.h_state) {
#pragma unroll
        for (int w = 0; w < JW; ++w) {
            const int j0 = 16 * (wave * JW + w) + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ok && j0 + r < a.R) a.h_state[(long long)node * a.R + j0 + r] = hown[w][r];
        }
    }
}

template <int JT, int NKX, bool OVEC, int ACT>
__global__ __launch_bounds__(256) void reservoir_layer_splitj_bf3(ResArgs a) {
    if (a.pred != nullptr && a.pred[0] != a.pred_want) return;
    if (a.n_pieces > 1) {
        // time piece blockIdx.y of this node tile (sgp_reservoir_pieces_f32): its own rows of x / out, its own state
        const int p = blockIdx.y;
        a.x += p * a.px;
        a.out += p * a.po;
        if (a.h_state) a.h_state += p * a.ps;
        if (p == a.n_pieces - 1) a.T = a.t_last;
        if (a.T <= 0) return;
    }
    if constexpr (ACT == SGP_ACT_TANH) {
        if (a.wp_h16) {
            // the two-piece fp16 loop needs the state inside [-1, 1] at every step: true from step 1 on, checked here for
            // the caller's initial state of this workgroup's 16 nodes (NaN fails the test too)
            int out_of_range = 0;
            if (a.h_state) {
                const int node = blockIdx.x * 16 + (threadIdx.x & 15);
                if (node < a.N)
                    for (int j = threadIdx.x >> 4; j < a.R; j += 16) out_of_range |= !(fabsf(a.h_state[(long long)node * a.R + j]) <= 1.f);
            }
            if (!__syncthreads_or(out_of_range)) { splitj_bf3_body<JT, NKX, OVEC, ACT, true>(a); return; }
        }
    }
    splitj_bf3_body<JT, NKX, OVEC, ACT, false>(a);
}
