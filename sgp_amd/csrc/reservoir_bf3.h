// Leaky echo-state layer on the 16-bit matrix cores with fp32-grade products (included by reservoir_impl.h
// inside namespace sgp_res; reference: lib/nn/reservoir/reservoir.py:77-81 stepped by :170-183).
//
// The exact-fp32 kernel (reservoir_layer) is bound by the matrix pipe: v_mfma_f32_16x16x4_f32 delivers 1/16 of the
// 16-bit rate, and at R = F = 64 its 128 MFMAs per tile and step take 4096 cycles of a SIMD -- 10.7 ms per 1024 steps
// of the target line against 8.7 ms for the layer's HBM traffic at the achievable rate.  Here every fp32 operand is cut
// into THREE bf16 pieces, v = b1 + b2 + b3 with b1 = rne(v), b2 = rne(v - b1), b3 = rne(v - b1 - b2): 24 significant
// bits, no scale and no bound on |v| needed (bf16 has the exponent range of fp32), and the product is summed from the
// six piece products of order <= 2^-16,
//     w v = w1 v1 + w1 v2 + w2 v1 + w1 v3 + w2 v2 + w3 v1        (+ O(2^-25) |w v| left out),
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6 x 1/8 of the MFMA instructions, each half as long -- 3/8 of
// the matrix time -- and an error per product below that of ONE fp32 rounding (measured against fp64 in
// tests/test_gpu_reservoir_bf3.py).  Any activation, any input: the kernel needs no bound on the state.
//
// Layout: as in reservoir_layer the contraction is computed transposed, D[j, n] += W[j, k] hT[k, n], so that the
// accumulator (lane = node n + 16 q, register r <-> unit 16 jt + 4 q + r) feeds the next step's B operand without
// leaving its lane: k-block p of the recurrent part takes the lane's 8 values of state tiles 2p and 2p + 1 as its
// 8 k-slots (slot i <-> unit 16 (2p + i / 4) + 4 q + i % 4), and the same permutation is baked into the packed
// W_hh fragments.  Of an input row a lane holds NKX features (register ks <-> feature bf3_feature(NKX, q, ks): 16-byte
// pieces that make 64 consecutive bytes per node and load instruction), k-block p of the input part takes
// ks = 8p .. 8p + 7.
//
// Packed weights (device workspace, copied to LDS by every workgroup):
//   bias  [JT][16] fp32
//   frag  [JT][KBH + KBX][3 pieces][64 lanes][8 bf16]     (16 bytes per lane: one ds_read_b128 per MFMA operand)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int bf3_kbh(int JT) { return (JT + 1) / 2; }
__host__ __device__ constexpr int bf3_kbx(int NKX) { return (NKX + 7) / 8; }
__host__ __device__ constexpr long long bf3_packed_bytes(int JT, int NKX) {
    return (long long)JT * 64 + (long long)JT * (bf3_kbh(JT) + bf3_kbx(NKX)) * 3 * 1024;
}

// narrow reservoirs and inputs: the state (16 JT), the input rows (4 NKX values per node) and the pieces of one
// k-block fit the 128 registers of four waves per SIMD
__host__ __device__ constexpr bool bf3_supported(int JT, int NKX) {
    return (JT == 2 || JT == 4) && (NKX == 4 || NKX == 8 || NKX == 16) && bf3_packed_bytes(JT, NKX) <= 64 * 1024;
}


// wide reservoirs (reservoir_layer_stream_bf3, reservoir_impl.h): bias [256] fp32, then per (k-block, half of the
// output tiles) 8 tiles x 3 pieces x 64 lanes x 16 bytes -- the order the kernel streams them through the LDS
__host__ __device__ constexpr bool sbf3_supported(int JT, int NKX) { return JT == 16 && NKX % 8 == 0 && NKX <= 32; }
__host__ __device__ constexpr long long sbf3_packed_bytes(int JT, int NKX) {
    return 1024 + 2ll * (JT / 2 + NKX / 8) * 8 * 3 * 1024;
}

// input feature held by lane group q in its register ks
__host__ __device__ constexpr int bf3_feature(int NKX, int q, int ks) {
    return NKX % 4 == 0 ? 16 * (ks >> 2) + 4 * q + (ks & 3) : q * NKX + ks;
}

__device__ __forceinline__ unsigned bf3_pk(float a, float b) {        // v_cvt_pk_bf16_f32 (round to nearest even)
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// two values -> one dword of each piece (low half = a, high half = b)
__device__ __forceinline__ void bf3_split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = bf3_pk(a, b);
    const float a1 = a - __builtin_bit_cast(float, p1 << 16), b1 = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = bf3_pk(a1, b1);
    const float a2 = a1 - __builtin_bit_cast(float, p2 << 16), b2 = b1 - __builtin_bit_cast(float, p2 & 0xffff0000u);
    p3 = bf3_pk(a2, b2);
}
__device__ __forceinline__ void bf3_split8(const float (&v)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned a, b, c;
        bf3_split2(v[2 * d], v[2 * d + 1], a, b, c);
        p1[d] = a; p2[d] = b; p3[d] = c;
    }
}
// acc += (w1 + w2 + w3) (v1 + v2 + v3), the six products of order <= 2^-16, smallest first
__device__ __forceinline__ f32x4 bf3_mac(const u32x4& w1, const u32x4& w2, const u32x4& w3,
                                         const u32x4& v1, const u32x4& v2, const u32x4& v3, f32x4 acc) {
    const bf16x8 W1 = __builtin_bit_cast(bf16x8, w1), W2 = __builtin_bit_cast(bf16x8, w2), W3 = __builtin_bit_cast(bf16x8, w3);
    const bf16x8 V1 = __builtin_bit_cast(bf16x8, v1), V2 = __builtin_bit_cast(bf16x8, v2), V3 = __builtin_bit_cast(bf16x8, v3);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W3, V1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W2, V2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W1, V3, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W2, V1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W1, V2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W1, V1, acc, 0, 0, 0);
    return acc;
}

// byte offset of the fragment of output tile jt, k-block kb, piece pc (0 = leading piece)
__host__ __device__ constexpr int bf3_frag_off(int KB, int jt, int kb, int pc) { return ((jt * KB + kb) * 3 + pc) * 1024; }
// units of the MFMA phase: (k-block, pair of output tiles); the input blocks come first
__host__ __device__ constexpr int bf3_unit_kb(int JT, int NKX, int u) {
    const int b = u / (JT / 2);
    return b < bf3_kbx(NKX) ? bf3_kbh(JT) + b : b - bf3_kbx(NKX);
}

// experiment switches (build with -DSGP_BF3_ABL=bits): 1 no MFMAs, 2 no piece conversion, 4 no activation,
// 8 no stores, 16 input rows loaded once, 32 no weight-fragment reads
#ifndef SGP_BF3_ABL
#define SGP_BF3_ABL 0
#endif
constexpr bool bf3_abl(int bit) { return (SGP_BF3_ABL & bit) != 0; }

template <int I> struct Bf3C { static constexpr int value = I; };
template <int B, int E, class F>
__device__ __forceinline__ void bf3_for(F&& f) {
    if constexpr (B < E) { f(Bf3C<B>{}); bf3_for<B + 1, E>(f); }
}
// weight fragments are read by hand (the compiler would put every ds_read in front of an s_waitcnt lgkmcnt(0))
template <int OFF> __device__ __forceinline__ void bf3_rd(u32x4& d, unsigned addr) {
    if constexpr (bf3_abl(32)) { asm volatile("" : "=v"(d) : "v"(addr)); return; }
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void bf3_wait(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
__device__ __forceinline__ f32x4 bf3_mfma(const u32x4& w, const u32x4& v, f32x4 acc) {
    if constexpr (bf3_abl(1)) { asm volatile("" : "+v"(acc) : "v"(w), "v"(v)); return acc; }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, v), acc, 0, 0, 0);
}

// fp16 two-piece products for a bounded state (arithmetic and error model: reservoir_splitj_bf3.h)
constexpr float kSj16StateScale = 16384.f;
// two scaled fp16 pieces of a pair of values (v_fma_mixlo / mixhi_f16: fp32 fma rounded once to fp16 into one half of
// the destination; the remainder of an 11-bit rounding of a 24-bit value is exact in the fma)
// (the low-half instruction leaves the other half of its destination alone and the high-half one then writes it: the
// destination needs no initial value -- "=v", not a zeroed "+v")
__device__ __forceinline__ void sj16_split2(float v0, float v1, float s, unsigned& hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(s), "v"(hi));
}
// the same with the (wave-uniform) scale in a scalar register
__device__ __forceinline__ void sj16_split2s(float v0, float v1, float s, unsigned& hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "s"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(v0), "s"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "s"(s), "v"(hi));
}
__device__ __forceinline__ f32x4 sj16_mfma(const u32x4& w, const u32x4& v, f32x4 acc) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, v), acc, 0, 0, 0);
}

// Exact widths only (R = 16 JT, F = 4 NKX, 16-byte aligned rows): every load and store of the time loop is
// unconditional and the loop body is straight-line code, so that the compiler counts its memory operations exactly
// (s_waitcnt vmcnt(4) for the input rows with the four stores behind them still in flight).  With a branch around any
// of them -- an exec-masked store, a wave-uniform `tile exists` test -- it falls back to vmcnt(0) at the top of every
// tile and step, and the step waits for its own stores to reach memory (measured: 5360 cycles per tile and step).
// N is a multiple of 16 here: the nodes of a ragged last tile go to the exact-fp32 kernel (launch_layer).
// PAIR (NT = 2, at most three waves per SIMD: 170 registers, 162-168 used): a wave multiplies its two tiles TOGETHER --
// every weight fragment read from the LDS serves both (24 instead of 48 KB of LDS reads per tile and step), four
// accumulators take turns behind every pair of fragments.  Measured N = 100 000, 256 steps: 3.46 -> 3.36 ms.  (The
// `-DSGP_BF3_ABL=32` build without any fragment read runs at 2.75 ms, but on undefined operands: NaNs through the matrix
// pipe draw less power and the clock rises -- the LDS is 28 % busy here, not the limit.)
// H16 (tanh): the recurrent products from two fp16 pieces of the bounded state (reservoir_splitj_bf3.h has the arithmetic and
// its error model; fragments and row scales of pack_weights_bf3h).  The launcher starts this instance under the launch
// predicate "no initial state outside [-1, 1]" and the three-piece instance under the opposite one (ResArgs::pred).
template <int JT, int NKX, int NT, bool PAIR = false, bool H16 = false>
__global__ __launch_bounds__(PAIR ? 768 : 1024, PAIR ? 3 : 4) void reservoir_layer_bf3(ResArgs a) {
    static_assert(!PAIR || NT == 2, "the pair loop is written for two tiles per wave");
    if (a.pred != nullptr && a.pred[0] != a.pred_want) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int KBH = bf3_kbh(JT), KBX = bf3_kbx(NKX), KB = KBH + KBX, NP = JT / 2, NU = KB * NP;
    const int lane = threadIdx.x & 63;
    const int n_in = lane & 15, q = lane >> 4;
    int tile0, tile1;                                        // the deal of reservoir_layer
    if (a.tiles_per_wave > 0) {
        const int per = a.tiles_per_wave;
        // waves c, c + 4, c + 8, .. of the workgroup share SIMD c and its `per` consecutive tiles; the workgroup has just as
        // many waves per SIMD as deal them evenly (per = 6: three waves of two tiles, not 2 + 2 + 1 + 1 -- a wave with
        // one tile runs ahead, finishes at half time and leaves the SIMD to two waves)
        const int wl = threadIdx.x >> 6, c = wl & 3, k = wl >> 2, wps = blockDim.x >> 8;
        const int base = per / wps, extra = per % wps;
        tile0 = (blockIdx.x * 4 + c) * per + k * base + min(k, extra);
        tile1 = min(tile0 + base + (k < extra ? 1 : 0), a.n_tiles);
    } else {
        const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        const int n_waves = gridDim.x * (blockDim.x >> 6);
        tile0 = (int)((long long)wave * a.n_tiles / n_waves);
        tile1 = (int)((long long)(wave + 1) * a.n_tiles / n_waves);
    }
    const int my_nt = max(tile1 - tile0, 0);

    // byte offsets of this lane's 16-byte piece of its node's input row / state row (32 bits: checked on the host)
    unsigned xo[NT], oo[NT];
    f32x4 h[NT][JT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int node = (tile0 + (i < my_nt ? i : 0)) * 16 + n_in;
        xo[i] = (unsigned)((node * a.xrs + 4 * q) * 4);
        oo[i] = (unsigned)((node * a.ors + 4 * q) * 4);
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            h[i][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.h_state && i < my_nt) h[i][jt] = *reinterpret_cast<const f32x4*>(a.h_state + (long long)node * a.R + 16 * jt + 4 * q);
        }
    }
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(H16 ? a.wp_h16l : a.wp_bf3);
        const int total4 = (int)((bf3_packed_bytes(JT, NKX) + (H16 ? JT * 64 : 0)) / 16);
        for (int i = threadIdx.x; i < total4; i += blockDim.x) reinterpret_cast<f32x4*>(lds)[i] = src[i];
        __syncthreads();
    }
    if (my_nt == 0) return;
    // the input rows of ONE tile and step (register 4 k4 + s <-> feature 16 k4 + 4 q + s: the four lanes of a node read
    // 64 consecutive bytes per instruction): requested right after the rows before them were consumed (the input
    // part runs first), so they are in flight under the recurrent part, the activation and the stores of that tile
    f32x4 xr[PAIR ? 2 : 1][NKX / 4];
    auto load_x = [&](int t, int i) {
        // the step's base is opaque to the optimiser and stays a scalar: otherwise it folds xo[i] into a 64-bit pointer per
        // lane and tile, which did not fit (one scratch reload and a vmcnt(0) per step in front of these loads)
        unsigned long long xb = (unsigned long long)(a.x + (long long)t * a.xss);
        asm volatile("" : "+s"(xb));
        typedef const __attribute__((address_space(1))) char* gptr;
        const gptr xp = (gptr)xb;
#pragma unroll
        for (int k4 = 0; k4 < NKX / 4; ++k4)
            xr[PAIR ? i : 0][k4] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(xp + xo[i] + 64 * k4);
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the initial state has landed
    load_x(0, 0);
    if constexpr (PAIR) load_x(0, my_nt > 1 ? 1 : 0);
    const unsigned fa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds + JT * 16) + lane * 16;
    float hscale = kSj16StateScale;
    if constexpr (H16) asm("" : "+s"(hscale));            // (a register: v_fma_mix takes no literal)

    // The time loop for G tiles multiplied together (1, or 2 = PAIR), MT tiles per wave, H16 = fp16 pieces for the
    // recurrent k-blocks.  A UNIT is (k-block, pair of output tiles); the input blocks come first.  Ring of six fragment
    // registers: [0,1] = third pieces of the unit's two output tiles, [2,3] = second (H16 recurrent: lo), [4,5] = leading
    // (hi).  An input unit multiplies 2 x (1 + 2 + 3) products per tile, a recurrent H16 unit 2 x (1 + 2), the output
    // tiles' (and the G tiles') chains alternating (no MFMA waits for the one before it); the next unit's pieces are
    // requested as soon as their registers are free.  LDS reads return in order: every wait counts the reads issued BEHIND
    // the pair it needs.
    auto run = [&](auto gc, auto mtc) {
        constexpr int G = decltype(gc)::value, MT = decltype(mtc)::value;
        auto rec = [](int u) constexpr { return H16 && bf3_unit_kb(JT, NKX, u) < KBH; };     // a two-piece fp16 unit
        for (int t = 0; t < a.T; ++t) {
            int wo = 0;                                      // keeps the bias reads inside the loop (see reservoir_layer)
            asm volatile("" : "+s"(wo));
            const float* bias_t = lds + wo;
#pragma unroll
            for (int i0 = 0; i0 < MT; i0 += G) {
                f32x4 acc[G][JT];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    acc[0][jt] = *reinterpret_cast<const f32x4*>(bias_t + jt * 16 + q * 4);
                    if constexpr (G == 2) acc[1][jt] = acc[0][jt];
                }
                u32x4 ring[6];
                bf3_rd<bf3_frag_off(KB, 0, bf3_unit_kb(JT, NKX, 0), 2)>(ring[0], fa); bf3_rd<bf3_frag_off(KB, 1, bf3_unit_kb(JT, NKX, 0), 2)>(ring[1], fa);
                bf3_rd<bf3_frag_off(KB, 0, bf3_unit_kb(JT, NKX, 0), 1)>(ring[2], fa); bf3_rd<bf3_frag_off(KB, 1, bf3_unit_kb(JT, NKX, 0), 1)>(ring[3], fa);
                u32x4 v1[G], v2[G], v3[G];
                bf3_for<0, NU>([&](auto uc) {
                    constexpr int u = decltype(uc)::value, kb = bf3_unit_kb(JT, NKX, u), j0 = 2 * (u % NP), j1 = j0 + 1;
                    constexpr bool last = u + 1 == NU, R = rec(u);
                    constexpr int un = last ? 0 : u + 1, kbn = bf3_unit_kb(JT, NKX, un), n0 = 2 * (un % NP), n1 = n0 + 1;
                    constexpr bool Rn = !last && rec(un);        // the next unit is a two-piece one
                    static_assert(u > 0 || !R, "the first unit is an input unit (its three pieces are requested ahead)");
                    if constexpr (u % NP == 0) {             // the pieces of a new k-block
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            float v[8];
                            if constexpr (kb >= KBH) {
                                constexpr int p = kb - KBH;
#pragma unroll
                                for (int s = 0; s < 8; ++s) v[s] = 8 * p + s < NKX ? xr[PAIR ? g : 0][(8 * p + s < NKX ? 8 * p + s : 0) >> 2][s & 3] : 0.f;
                            } else {
#pragma unroll
                                for (int s = 0; s < 8; ++s) v[s] = h[i0 + g][2 * kb + (s >> 2)][s & 3];
                            }
                            if constexpr (R) {
#pragma unroll
                                for (int d = 0; d < 4; ++d) {
                                    unsigned hi, lo;
                                    sj16_split2s(v[2 * d], v[2 * d + 1], hscale, hi, lo);
                                    v1[g][d] = hi; v2[g][d] = lo;
                                }
                            } else if constexpr (bf3_abl(2)) {
#pragma unroll
                                for (int d = 0; d < 4; ++d) { v1[g][d] = __builtin_bit_cast(unsigned, v[2 * d]); v2[g][d] = __builtin_bit_cast(unsigned, v[2 * d + 1]); v3[g][d] = v1[g][d] ^ v2[g][d]; }
                            } else {
                                bf3_split8(v, v1[g], v2[g], v3[g]);
                            }
                        }
                        if constexpr (kb == KBH + KBX - 1 && !bf3_abl(16)) {
                            // the input rows are consumed: request those of this wave's next tile and step (the last
                            // step re-reads its own rows instead of branching)
                            if constexpr (G == 2) {
                                const int tn = t + 1 < a.T ? t + 1 : t;
                                load_x(tn, 0); load_x(tn, 1);
                            } else {
                                if (i0 + 1 < MT) load_x(t, i0 + 1 < MT ? i0 + 1 : 0);
                                else load_x(t + 1 < a.T ? t + 1 : t, 0);
                            }
                        }
                    }
                    auto mm = [&](const u32x4& f0, const u32x4& f1, const u32x4 (&v)[G]) {
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            if constexpr (R) { acc[g][j0] = sj16_mfma(f0, v[g], acc[g][j0]); acc[g][j1] = sj16_mfma(f1, v[g], acc[g][j1]); }
                            else { acc[g][j0] = bf3_mfma(f0, v[g], acc[g][j0]); acc[g][j1] = bf3_mfma(f1, v[g], acc[g][j1]); }
                        }
                    };
                    bf3_rd<bf3_frag_off(KB, j0, kb, 0)>(ring[4], fa); bf3_rd<bf3_frag_off(KB, j1, kb, 0)>(ring[5], fa);
                    if constexpr (!R) {
                        bf3_wait<4>(ring[0], ring[1]);
                        mm(ring[0], ring[1], v1);
                        if constexpr (!last && !Rn) { bf3_rd<bf3_frag_off(KB, n0, kbn, 2)>(ring[0], fa); bf3_rd<bf3_frag_off(KB, n1, kbn, 2)>(ring[1], fa); }
                    }
                    bf3_wait<2 + (!R && !last && !Rn ? 2 : 0)>(ring[2], ring[3]);
                    if constexpr (R) {
                        mm(ring[2], ring[3], v1);                              // lo hi
                    } else {
                        mm(ring[2], ring[3], v2);
                        mm(ring[2], ring[3], v1);
                    }
                    if constexpr (!last) { bf3_rd<bf3_frag_off(KB, n0, kbn, 1)>(ring[2], fa); bf3_rd<bf3_frag_off(KB, n1, kbn, 1)>(ring[3], fa); }
                    bf3_wait<(!last ? 2 : 0) + (!R && !last && !Rn ? 2 : 0)>(ring[4], ring[5]);
                    if constexpr (R) {
                        mm(ring[4], ring[5], v2);                              // hi lo
                        mm(ring[4], ring[5], v1);                              // hi hi
                    } else {
                        mm(ring[4], ring[5], v3);
                        mm(ring[4], ring[5], v2);
                        mm(ring[4], ring[5], v1);
                    }
                });
                if constexpr (H16) {
                    // the rows' sums carry 2^(e_j + 14) (bias and input fragments were scaled to match): back, exactly
                    const float* rsc_t = bias_t + (int)(bf3_packed_bytes(JT, NKX) / 4);
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        const f32x4 rs = *reinterpret_cast<const f32x4*>(rsc_t + jt * 16 + q * 4);
#pragma unroll
                        for (int g = 0; g < G; ++g) acc[g][jt] *= rs;
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int i = i0 + g;
                    // activation
                    if (bf3_abl(4)) {
                    } else if (H16 || a.act == SGP_ACT_TANH) {
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt) acc[g][jt] = tanh_r4(acc[g][jt]);
                    } else if (a.act == SGP_ACT_RELU) {
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[g][jt][r] = fmaxf(acc[g][jt][r], 0.f);
                    } else if (a.act == SGP_ACT_TANH_REL) {
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[g][jt][r] = tanh_rel(acc[g][jt][r]);
                    } else if (a.act == SGP_ACT_SELF_NORM) {
                        float ss = 0.f;
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) ss = fmaf(acc[g][jt][r], acc[g][jt][r], ss);
                        ss += __shfl_xor(ss, 16);
                        ss += __shfl_xor(ss, 32);
                        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize(eps=1e-12)
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[g][jt][r] *= inv;
                    }
                    // leak + store (64 bytes of each of 16 rows per instruction.  Swapping pieces between lanes n and n + 8 so
                    // that an instruction writes 8 whole 128-byte lines was measured: no gain, 3.60 vs 3.45-3.6 ms)
                    // (scalar base + 32-bit lane offset, like the input rows: a 64-bit pointer per lane and tile did not fit)
                    unsigned long long ob = (unsigned long long)(a.out + (long long)t * a.oss);
                    asm volatile("" : "+s"(ob));
                    typedef __attribute__((address_space(1))) char* gptr;
                    const gptr op = (gptr)ob;
                    // (vector copies made HERE: as loop invariants the pairs {alpha, alpha} .. of the packed fmas were spilled and
                    // reloaded every step behind an s_waitcnt vmcnt(0))
                    float al = a.alpha, om = a.one_minus_alpha;
                    asm volatile("" : "+v"(al), "+v"(om));
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            h[i][jt][r] = (H16 || a.act == SGP_ACT_TANH) ? leak_tanh_r(h[i][jt][r], acc[g][jt][r], al, om)
                                                                        : leak(h[i][jt][r], acc[g][jt][r], al, om);
                        if (!bf3_abl(8) || t == 0) *reinterpret_cast<__attribute__((address_space(1))) f32x4*>(op + oo[i] + 64 * jt) = h[i][jt];
                    }
                }
            }
        }
    };
    if constexpr (PAIR) {
        if (my_nt > 1) run(Bf3C<2>{}, Bf3C<2>{}); else run(Bf3C<1>{}, Bf3C<1>{});
    } else {
        if (NT > 1 && my_nt > 1) run(Bf3C<1>{}, Bf3C<NT>{}); else run(Bf3C<1>{}, Bf3C<1>{});
    }
    if (a.h_state) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
                if (i < my_nt) *reinterpret_cast<f32x4*>(a.h_state + (long long)((tile0 + i) * 16 + n_in) * a.R + 16 * jt + 4 * q) = h[i][jt];
    }
}
