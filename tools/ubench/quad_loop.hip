// Steady-state rate of the spmm_pipe inner loop on gfx950: per quad 4 operand reads (ds_read_b128 at
// offsets that come from LDS), 1 weight read, 1 offset read, 16 v_mfma_f32_4x4x1_16b_f32, software
// pipelined two quads deep exactly as in sgp_amd/csrc/spmm_pipe.hip.  Variants isolate the pipes.
// Build: hipcc --offload-arch=gfx950 -O3 -o quad_loop quad_loop.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// VARIANT bit0: no MFMA   bit1: no X reads   bit2: no W/I reads   bit3: barrier every PHASE quads
template <int VARIANT, int PHASE>
__global__ __launch_bounds__(1024) void quad_loop(float* sink, unsigned long long* cyc, int n_quads, int rounds) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, li = tid & 15, q = lane >> 4;
    const int wave = tid >> 6;
    // stage: 448 rows x 256 B of data; stream: per wave n_quads x (256 B weights + 64 B offsets)
    for (int i = tid; i < 448 * 64; i += blockDim.x) ((float*)lds)[i] = (i % 977) * 1e-3f;
    char* wl = lds + 448 * 256;
    char* il = wl + 128 * 256;
    for (int i = tid; i < 128 * 64; i += blockDim.x) ((float*)wl)[i] = 1e-3f;
    for (int i = tid; i < 128 * 16 + 64; i += blockDim.x) ((int*)il)[i] = ((i * 37 + 11) % 448) * 256;
    __syncthreads();
    const char* WP = wl + ((wave * 8) % 100) * 256 + (q * 4 + (lane & 3)) * 16;
    const char* IP = il + ((wave * 8) % 100) * 64 + q * 16;
    const char* xmine = lds + li * 16;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    f32x4 Wa, Wb, Xa[4], Xb[4];
    int4 Ia, Ib;
#define LDW(DST, C) if (!(VARIANT & 4)) DST = *reinterpret_cast<const f32x4*>(WP + ((C) & 15) * 256)
#define LDI(DST, C) if (!(VARIANT & 4)) DST = *reinterpret_cast<const int4*>(IP + ((C) & 15) * 64)
#define LD1(DST, OFF) if (!(VARIANT & 2)) DST = *reinterpret_cast<const f32x4*>(xmine + (OFF))
#define LDX(X, I) LD1(X[0], (I).x); LD1(X[1], (I).y); LD1(X[2], (I).z); LD1(X[3], (I).w);
#define MFA(ACC, W, XS) asm("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(W), "v"(XS));
#define SUPER(W, XV) if (VARIANT & 48) { MFA(acc0, W, XV.x) MFA(acc1, W, XV.y) MFA(acc2, W, XV.z) MFA(acc3, W, XV.w) } else if (!(VARIANT & 1)) { \
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.x, acc0, 0, 0, 0);           \
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.y, acc1, 0, 0, 0);           \
    acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.z, acc2, 0, 0, 0);           \
    acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(W, XV.w, acc3, 0, 0, 0); }         \
    else { asm volatile("" :: "v"(XV.x), "v"(XV.y), "v"(XV.z), "v"(XV.w), "v"(W)); }
#define SG(M, N) __builtin_amdgcn_sched_group_barrier(M, N, 0);
#define BODY_T(W, X) SUPER(W.x, X[0]) SUPER(W.y, X[1]) SUPER(W.z, X[2]) SUPER(W.w, X[3])
#define FENCE __builtin_amdgcn_sched_barrier(0);
#define BODY_L(W, X, I, C)                                                       \
    if (VARIANT & 32) {                                                          \
    SUPER(W.x, X[0]) FENCE LD1(X[0], (I).x); FENCE SUPER(W.y, X[1]) FENCE LD1(X[1], (I).y); FENCE \
    SUPER(W.z, X[2]) FENCE LD1(X[2], (I).z); FENCE SUPER(W.w, X[3]) FENCE LD1(X[3], (I).w);       \
    LDI(I, (C) + 4); LDW(W, (C) + 2); FENCE                                      \
    } else {                                                                     \
    SUPER(W.x, X[0]) LD1(X[0], (I).x); SUPER(W.y, X[1]) LD1(X[1], (I).y);        \
    SUPER(W.z, X[2]) LD1(X[2], (I).z); SUPER(W.w, X[3]) LD1(X[3], (I).w);        \
    LDW(W, (C) + 2); LDI(I, (C) + 4);                                            \
    SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 1) SG(0x008, 4) SG(0x100, 3) }
    Wa = Wb = f32x4{1e-3f, 1e-3f, 1e-3f, 1e-3f};
    Ia = Ib = int4{256, 512, 768, 1024};
    Xa[0] = Xa[1] = Xa[2] = Xa[3] = Xb[0] = Xb[1] = Xb[2] = Xb[3] = f32x4{1, 1, 1, 1};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        const int NQ = (VARIANT & 8) ? PHASE : n_quads;
        const int phases = (VARIANT & 8) ? n_quads / PHASE : 1;
        for (int ph = 0; ph < phases; ++ph) {
            LDI(Ia, 0); LDI(Ib, 1); LDW(Wa, 0); LDW(Wb, 1);
            if (VARIANT & 8) asm volatile("s_barrier" ::: "memory");
            LDX(Xa, Ia) LDX(Xb, Ib)
            LDI(Ia, 2); LDI(Ib, 3);
            int c = 0;
            for (; c + 3 < NQ; c += 2) {
                BODY_L(Wa, Xa, Ia, c)
                BODY_L(Wb, Xb, Ib, c + 1)
            }
            BODY_T(Wa, Xa)
            BODY_T(Wb, Xb)
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + tid] = acc0.x + acc1.y + acc2.z + acc3.w + Xa[0].x + Xb[1].y + Wa.x + Wb.y + Ia.x + Ib.y;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VARIANT, int PHASE>
void run(const char* name, int threads) {
    const int blocks = 256, n_quads = 64, rounds = 200;
    float* sink; unsigned long long* cyc;
    hipMalloc(&sink, (size_t)blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    auto k = quad_loop<VARIANT, PHASE>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 160 * 1024, 0, sink, cyc, n_quads, rounds);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 160 * 1024, 0, sink, cyc, n_quads, rounds);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    const double quads_per_simd = (double)n_quads * rounds * (threads / 256.0);
    printf("%-34s waves/SIMD=%d  cycles/quad/SIMD=%7.1f  (ideal 128 MFMA, %5.1f%% of MFMA peak)  %.3f ms\n", name, threads / 256,
           (double)h[0] / quads_per_simd, 12800.0 / ((double)h[0] / quads_per_simd), ms);
    hipFree(sink); hipFree(cyc);
}

int main() {
    for (int thr : {256, 512, 1024}) {
        if (thr == 256) { run<0, 4>("full", 256); run<1, 4>("no MFMA", 256); run<2, 4>("no X reads", 256); run<4, 4>("no W/I reads", 256); run<6, 4>("MFMA only", 256); run<8, 4>("full + barrier/4 quads", 256); run<8, 8>("full + barrier/8 quads", 256); }
        if (thr == 512) { run<0, 4>("full", 512); run<1, 4>("no MFMA", 512); run<2, 4>("no X reads", 512); run<4, 4>("no W/I reads", 512); run<6, 4>("MFMA only", 512); run<8, 4>("full + barrier/4 quads", 512); run<8, 8>("full + barrier/8 quads", 512); }
        if (thr == 1024) { run<32, 4>("full, AGPR + fenced order", 1024); run<40, 4>("full+barrier/4, AGPR + fenced", 1024); run<16, 4>("full, acc in AGPRs", 1024); run<24, 4>("full + barrier/4, acc in AGPRs", 1024); run<22, 4>("MFMA only, AGPR", 1024); run<0, 4>("full", 1024); run<1, 4>("no MFMA", 1024); run<2, 4>("no X reads", 1024); run<4, 4>("no W/I reads", 1024); run<6, 4>("MFMA only", 1024); run<8, 4>("full + barrier/4 quads", 1024); run<8, 8>("full + barrier/8 quads", 1024); }
    }
    return 0;
}
