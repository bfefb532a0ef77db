// Steady-state rate of the CURRENT spmm_pipe inner loop (one float + one int per lane and quad,
// cbsz:2 / abid weights, DPP quad_perm address add, prologue in steady-state order) and what its
// pieces cost.  VARIANT bit0: X addresses loop-invariant (no VALU add, offsets still loaded)
//                       bit1: no X reads    bit2: no W/I reads    bit3: no MFMAs
// Build: hipcc --offload-arch=gfx950 -O3 -o quad_loop2 quad_loop2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(1024) void quad_loop(float* sink, unsigned long long* cyc, int n_quads, int rounds) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, li = tid & 15, q = lane >> 4;
    const int wave = tid >> 6;
    for (int i = tid; i < 448 * 64; i += blockDim.x) ((float*)lds)[i] = (i % 977) * 1e-3f;
    char* wl = lds + 448 * 256;
    char* il = wl + 128 * 256;
    for (int i = tid; i < 128 * 64; i += blockDim.x) ((float*)wl)[i] = 1e-3f;
    for (int i = tid; i < 128 * 16 + 64; i += blockDim.x) ((int*)il)[i] = ((i * 37 + 11) % 448) * 256;
    __syncthreads();
    typedef const __attribute__((address_space(3))) f32x4* lds_f4_t;
    typedef const __attribute__((address_space(3))) float* lds_f1_t;
    typedef const __attribute__((address_space(3))) unsigned* lds_u1_t;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned WP = lds0 + 448 * 256 + ((wave * 8) % 100) * 256 + lane * 4;
    unsigned IP = lds0 + 448 * 256 + 128 * 256 + ((wave * 8) % 100) * 64 + q * 16 + (lane & 3) * 4;
    asm volatile("" : "+v"(WP), "+v"(IP));
    const unsigned xmine = lds0 + li * 16;
    unsigned xfix[4] = {xmine + 256u * (wave + 1), xmine + 256u * (wave + 17), xmine + 256u * (wave + 33), xmine + 256u * (wave + 49)};
    asm volatile("" : "+v"(xfix[0]), "+v"(xfix[1]), "+v"(xfix[2]), "+v"(xfix[3]));
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    float Wa = 1e-3f, Wb = 1e-3f;
    f32x4 Xa[4], Xb[4];
    unsigned Ia = 256, Ib = 512;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) u32x4* lds_u4_t;
    u32x4 Ia4 = {256, 512, 768, 1024}, Ib4 = Ia4;
    const unsigned IP4 = lds0 + 448 * 256 + 128 * 256 + ((wave * 8) % 100) * 64 + q * 16;
    Xa[0] = Xa[1] = Xa[2] = Xa[3] = Xb[0] = Xb[1] = Xb[2] = Xb[3] = f32x4{1, 1, 1, 1};
#define LDW(DST, C) if (!(VARIANT & 4)) DST = *(lds_f1_t)(WP + ((C) & 15) * 256)
#define LDI(DST, C) if (VARIANT & 32) DST##4 = *(lds_u4_t)(IP4 + ((C) & 15) * 64); else if (!(VARIANT & 4)) DST = *(lds_u1_t)(IP + ((C) & 15) * 64)
#define QP(S) ((S) | ((S) << 2) | ((S) << 4) | ((S) << 6))
#define LD1(DST, I, S) if (VARIANT & 32) { DST = *(lds_f4_t)(xmine + I##4[S]); } else if (!(VARIANT & 2)) { if (VARIANT & 1) { asm volatile("" :: "v"(I)); DST = *(lds_f4_t)(xfix[S]); } \
    else DST = *(lds_f4_t)(xmine + (unsigned)__builtin_amdgcn_update_dpp(0, (int)(I), QP(S), 0xf, 0xf, true)); }
#define LDX(X, I) LD1(X[0], I, 0) LD1(X[1], I, 1) LD1(X[2], I, 2) LD1(X[3], I, 3)
#define SUPER(W, XV, S) if (!(VARIANT & 8)) {                                    \
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.x, acc0, 2, S, 0);           \
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.y, acc1, 2, S, 0);           \
    acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.z, acc2, 2, S, 0);           \
    acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.w, acc3, 2, S, 0); }         \
    else { asm volatile("" :: "v"(XV.x), "v"(XV.y), "v"(XV.z), "v"(XV.w), "v"(W)); }
#define SG(M, N) __builtin_amdgcn_sched_group_barrier(M, N, 0);
#define LDXA(X, A) X[0] = *(lds_f4_t)(A[0]); X[1] = *(lds_f4_t)(A[1]); X[2] = *(lds_f4_t)(A[2]); X[3] = *(lds_f4_t)(A[3]);
#define BODY_T(W, X) SUPER(W, X[0], 0) SUPER(W, X[1], 1) SUPER(W, X[2], 2) SUPER(W, X[3], 3)
#define BODY_L(W, X, I, C)                                                       \
    SUPER(W, X[0], 0) LD1(X[0], I, 0) SUPER(W, X[1], 1) LD1(X[1], I, 1)          \
    SUPER(W, X[2], 2) LD1(X[2], I, 2) SUPER(W, X[3], 3) LD1(X[3], I, 3)          \
    LDW(W, (C) + 2); LDI(I, (C) + 4);                                            \
    SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 3)
    unsigned long long t0 = __builtin_readcyclecounter();
    if (VARIANT & 64) {
        // single operand buffer, addresses in registers (no VALU, no offset reads): super-step s of
        // the next quad is requested right after the MFMAs of super-step s have issued
        LDXA(Xa, xfix) LDW(Wa, 0); LDW(Wb, 1);
#define QUAD1(W, C)                                                               \
                SUPER(W, Xa[0], 0) Xa[0] = *(lds_f4_t)(xfix[0]);                  \
                SUPER(W, Xa[1], 1) Xa[1] = *(lds_f4_t)(xfix[1]);                  \
                SUPER(W, Xa[2], 2) Xa[2] = *(lds_f4_t)(xfix[2]);                  \
                SUPER(W, Xa[3], 3) Xa[3] = *(lds_f4_t)(xfix[3]);                  \
                LDW(W, (C) + 2);                                                  \
                SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 2)
        for (int r = 0; r < rounds; ++r) {
            for (int c = 0; c < n_quads; c += 2) {
                QUAD1(Wa, c)
                QUAD1(Wb, c + 1)
            }
        }
    } else if (VARIANT & 16) {
        // address adds hoisted out of the MFMA stream: all offsets of a 4-quad phase are read and
        // added up front, the quads read their operands through ready-made address registers
        unsigned I0, I1, I2, I3, A0[4], A1[4], A2[4], A3[4];
        u32x4 I04, I14, I24, I34; (void)I04; (void)I14; (void)I24; (void)I34;
#define ADDR(A, I) A[0] = xmine + (unsigned)__builtin_amdgcn_update_dpp(0, (int)(I), QP(0), 0xf, 0xf, true); \
                   A[1] = xmine + (unsigned)__builtin_amdgcn_update_dpp(0, (int)(I), QP(1), 0xf, 0xf, true); \
                   A[2] = xmine + (unsigned)__builtin_amdgcn_update_dpp(0, (int)(I), QP(2), 0xf, 0xf, true); \
                   A[3] = xmine + (unsigned)__builtin_amdgcn_update_dpp(0, (int)(I), QP(3), 0xf, 0xf, true);
#define BODY_LA(W, X, A, C)                                                      \
    SUPER(W, X[0], 0) X[0] = *(lds_f4_t)(A[0]); SUPER(W, X[1], 1) X[1] = *(lds_f4_t)(A[1]); \
    SUPER(W, X[2], 2) X[2] = *(lds_f4_t)(A[2]); SUPER(W, X[3], 3) X[3] = *(lds_f4_t)(A[3]); \
    LDW(W, (C) + 2);                                                             \
    SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 2)
        for (int r = 0; r < rounds * (n_quads / 4); ++r) {
            LDI(I0, r); LDI(I1, r + 1); LDI(I2, r + 2); LDI(I3, r + 3);
            ADDR(A0, I0) ADDR(A1, I1) ADDR(A2, I2) ADDR(A3, I3)
            LDXA(Xa, A0) LDW(Wa, 0); LDXA(Xb, A1) LDW(Wb, 1);
            __builtin_amdgcn_s_setprio(2);
            BODY_LA(Wa, Xa, A2, 0)
            __builtin_amdgcn_s_setprio(0);
            BODY_LA(Wb, Xb, A3, 1)
            BODY_T(Wa, Xa)
            BODY_T(Wb, Xb)
        }
    } else
    for (int r = 0; r < rounds; ++r) {
        LDI(Ia, 0); LDI(Ib, 1);
        LDX(Xa, Ia) LDW(Wa, 0); LDI(Ia, 2);
        LDX(Xb, Ib) LDW(Wb, 1); LDI(Ib, 3);
        int c = 0;
        for (; c + 3 < n_quads; c += 2) {
            __builtin_amdgcn_s_setprio(2);
            BODY_L(Wa, Xa, Ia, c)
            __builtin_amdgcn_s_setprio(0);
            BODY_L(Wb, Xb, Ib, c + 1)
        }
        BODY_T(Wa, Xa)
        BODY_T(Wb, Xb)
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + tid] = Ia4.x + Ib4.y + acc0.x + acc1.y + acc2.z + acc3.w + Xa[0].x + Xb[1].y + Wa + Wb + Ia + Ib;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VARIANT>
void run(const char* name, int threads, int n_quads) {
    const int blocks = 256, rounds = 12800 / n_quads;
    float* sink; unsigned long long* cyc;
    (void)hipMalloc(&sink, (size_t)blocks * threads * 4); (void)hipMalloc(&cyc, blocks * 8);
    auto k = quad_loop<VARIANT>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 160 * 1024, 0, sink, cyc, n_quads, rounds);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 160 * 1024, 0, sink, cyc, n_quads, rounds);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double quads_per_simd = (double)n_quads * rounds * (threads / 256.0);
    printf("%-40s waves/SIMD=%d quads/phase=%3d  ns/quad/SIMD=%6.1f  (%5.1f%% of the matrix pipe at 2.4 GHz)\n", name, threads / 256, n_quads,
           ms * 1e6 / quads_per_simd, 100.0 * 53.3 / (ms * 1e6 / quads_per_simd));
    (void)hipFree(sink); (void)hipFree(cyc);
}

int main() {
    for (int nq : {64, 4}) {
        run<0>("full (current kernel form)", 1024, nq);
        run<1>("no address add (offsets still loaded)", 1024, nq);
        run<4>("no W/I reads", 1024, nq);
        run<5>("no W/I reads, no add", 1024, nq);
        run<2>("no X reads", 1024, nq);
        run<6>("MFMA only", 1024, nq);
        run<8>("no MFMA", 1024, nq);
        run<64>("single buffer, addresses in registers", 1024, nq);
        run<64>("single buffer, addr regs, 2 waves/SIMD", 512, nq);
        run<64>("single buffer, addr regs, 1 wave/SIMD", 256, nq);
        run<32>("plain v_add, offsets 4 ints per lane", 1024, nq);
        run<16>("adds hoisted (4-quad phases)", 1024, nq);
        run<16>("adds hoisted, 2 waves/SIMD", 512, nq);
        run<0>("full, 2 waves/SIMD", 512, nq);
        run<0>("full, 1 wave/SIMD", 256, nq);
    }
    return 0;
}
