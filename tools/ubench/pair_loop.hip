// Steady-state rate of a row-group-PAIR inner loop on gfx950: a wave owns two 4-row groups and walks
// the union of their columns; per union quad 4 operand reads + 1 offset read + 2 weight reads
// feed 32 v_mfma_f32_4x4x1_16b_f32 (two accumulator sets).  Compare with quad_loop.hip (one group
// per wave: 6 reads per 16 MFMAs).  Build: hipcc --offload-arch=gfx950 -O3 -o pair_loop pair_loop.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH3>
__global__ __launch_bounds__(1024) void pair_loop(float* sink, unsigned long long* cyc, int n_quads, int rounds, int phase, int compute_waves) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, li = tid & 15, q = lane >> 4;
    const int wave = tid >> 6;
    for (int i = tid; i < 448 * 64; i += blockDim.x) ((float*)lds)[i] = (i % 977) * 1e-3f;
    char* wl = lds + 448 * 256;
    char* il = wl + 128 * 256;
    for (int i = tid; i < 128 * 64; i += blockDim.x) ((float*)wl)[i] = 1e-3f;
    for (int i = tid; i < 128 * 16 + 64; i += blockDim.x) ((int*)il)[i] = ((i * 37 + 11) % 448) * 256;
    __syncthreads();
    const char* WP = wl + ((wave * 8) % 90) * 256 + (q * 4 + (lane & 3)) * 16;
    const char* IP = il + ((wave * 8) % 90) * 64 + q * 16;
    const char* xmine = lds + li * 16;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, b0 = a0, b1 = a0, b2 = a0, b3 = a0;
    f32x4 W1a, W2a, W1b, W2b, Xa[4], Xb[4];
    int4 Ia, Ib;
#define LDW(DST, C) DST = *reinterpret_cast<const f32x4*>(WP + ((C) & 15) * 256)
#define LDI(DST, C) DST = *reinterpret_cast<const int4*>(IP + ((C) & 15) * 64)
#define LD1(DST, OFF) DST = *reinterpret_cast<const f32x4*>(xmine + (OFF))
#define LDX(X, I) LD1(X[0], (I).x); LD1(X[1], (I).y); LD1(X[2], (I).z); LD1(X[3], (I).w);
#define SUPER2(WA, WB, XV)                                                       \
    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(WA, XV.x, a0, 0, 0, 0);              \
    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(WA, XV.y, a1, 0, 0, 0);              \
    a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(WA, XV.z, a2, 0, 0, 0);              \
    a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(WA, XV.w, a3, 0, 0, 0);              \
    b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(WB, XV.x, b0, 0, 0, 0);              \
    b1 = __builtin_amdgcn_mfma_f32_4x4x1f32(WB, XV.y, b1, 0, 0, 0);              \
    b2 = __builtin_amdgcn_mfma_f32_4x4x1f32(WB, XV.z, b2, 0, 0, 0);              \
    b3 = __builtin_amdgcn_mfma_f32_4x4x1f32(WB, XV.w, b3, 0, 0, 0);
#define SG(M, N) __builtin_amdgcn_sched_group_barrier(M, N, 0);
#define BODY_T(W1, W2, X) SUPER2(W1.x, W2.x, X[0]) SUPER2(W1.y, W2.y, X[1]) SUPER2(W1.z, W2.z, X[2]) SUPER2(W1.w, W2.w, X[3])
#define BODY_L(W1, W2, X, I, C)                                                  \
    SUPER2(W1.x, W2.x, X[0]) LD1(X[0], (I).x); SUPER2(W1.y, W2.y, X[1]) LD1(X[1], (I).y); \
    SUPER2(W1.z, W2.z, X[2]) LD1(X[2], (I).z); SUPER2(W1.w, W2.w, X[3]) LD1(X[3], (I).w); \
    LDW(W1, (C) + 2); LDW(W2, (C) + 3); LDI(I, (C) + 4);                         \
    SG(0x008, 8) SG(0x100, 1) SG(0x008, 8) SG(0x100, 1) SG(0x008, 8) SG(0x100, 1) SG(0x008, 8) SG(0x100, 4)
    W1a = W2a = W1b = W2b = f32x4{1e-3f, 1e-3f, 1e-3f, 1e-3f};
    Ia = Ib = int4{256, 512, 768, 1024};
    Xa[0] = Xa[1] = Xa[2] = Xa[3] = Xb[0] = Xb[1] = Xb[2] = Xb[3] = f32x4{1, 1, 1, 1};
    unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < compute_waves) {
        for (int r = 0; r < rounds; ++r) {
            const int phases = n_quads / phase;
            for (int ph = 0; ph < phases; ++ph) {
                LDI(Ia, 0); LDI(Ib, 1); LDW(W1a, 0); LDW(W2a, 1); LDW(W1b, 2); LDW(W2b, 3);
                asm volatile("s_barrier" ::: "memory");
                LDX(Xa, Ia) LDX(Xb, Ib)
                LDI(Ia, 2); LDI(Ib, 3);
                int c = 0;
                for (; c + 3 < phase; c += 2) {
                    BODY_L(W1a, W2a, Xa, Ia, c)
                    BODY_L(W1b, W2b, Xb, Ib, c + 1)
                }
                BODY_T(W1a, W2a, Xa)
                BODY_T(W1b, W2b, Xb)
            }
        }
    } else {
        for (int r = 0; r < rounds; ++r)
            for (int ph = 0; ph < n_quads / phase; ++ph) asm volatile("s_barrier" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + tid] = a0.x + a1.y + a2.z + a3.w + b0.x + b1.y + b2.z + b3.w + Xa[0].x + Xb[1].y + W1a.x + W2b.y + Ia.x + Ib.y;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

void run(const char* name, int threads, int compute_waves, int phase) {
    const int blocks = 256, n_quads = 64, rounds = 200;
    float* sink; unsigned long long* cyc;
    (void)hipMalloc(&sink, (size_t)blocks * threads * 4); (void)hipMalloc(&cyc, blocks * 8);
    auto k = pair_loop<0>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 160 * 1024, 0, sink, cyc, n_quads, rounds, phase, compute_waves);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 160 * 1024, 0, sink, cyc, n_quads, rounds, phase, compute_waves);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // MFMA-equivalent quads (16 MFMAs) per SIMD
    const double q16 = 2.0 * n_quads * rounds * (compute_waves / 4.0);
    printf("%-40s threads=%4d compute waves=%2d phase=%2d  %.3f ms  -> %.1f ns per 16 MFMAs per SIMD\n", name, threads, compute_waves, phase, ms, ms * 1e6 / q16);
    (void)hipFree(sink); (void)hipFree(cyc);
}

int main() {
    run("pair loop", 512, 8, 4);
    run("pair loop", 512, 8, 8);
    run("pair loop + 4 idle waves", 768, 8, 4);
    run("pair loop", 1024, 16, 4);
    run("pair loop", 1024, 16, 2);
    run("pair loop", 768, 12, 4);
    return 0;
}
