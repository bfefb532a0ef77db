// Instruction-cost microbenchmark for gfx950: cycles per wave-instruction for the VALU / DPP /
// LDS / MFMA forms the SpMM kernels choose between.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void bench(unsigned long long* out, float* sink, int iters, float sval) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = i * 0.001f;
    __syncthreads();
    float a0 = lane, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    float x = lane * 0.5f, w = 0.25f;
    int ad = lane * 4, ad16 = lane * 16, ad8 = lane * 8;
    f32x4 m = {0, 0, 0, 0}, m2 = {0, 0, 0, 0};
    f32x4 r0, r1, r2, r3;
    double d0 = lane, d1 = 1, d2 = 2, d3 = 3, dm = 0.5, dn = 0.25;
    float q0, q1, q2, q3;
    unsigned long long w0 = wall_clock64();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {   // v_fmac vgpr
            REP16(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                               "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(x));)
        } else if constexpr (KIND == 1) {   // v_fmac with sgpr
            REP16(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                               "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sval), "v"(x));)
        } else if constexpr (KIND == 2) {   // fmac dpp row_ror
            REP16(asm volatile("v_fmac_f32_dpp %0, %8, %9 row_ror:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_ror:2 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %2, %8, %9 row_ror:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %4, %8, %9 row_ror:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_ror:6 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %6, %8, %9 row_ror:7 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(x));)
        } else if constexpr (KIND == 3) {   // fmac dpp row_newbcast
            REP16(asm volatile("v_fmac_f32_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(x));)
        } else if constexpr (KIND == 4) {   // pk_fma (8 instr = 16 fma lanes-pairs)
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                               "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm), "v"(dn));)
        } else if constexpr (KIND == 5) {   // v_add_u32_dpp + v_mov_dpp
            REP16(asm volatile("v_add_u32_dpp %0, %8, %9 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_ror:2 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32_dpp %2, %8, %9 row_ror:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32_dpp %4, %8, %9 row_ror:5 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_ror:6 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32_dpp %6, %8, %9 row_ror:7 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_ror:8 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(x));)
        } else if constexpr (KIND == 6) {   // ds_read_b32 x8
            REP16(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                               "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(ad));)
        } else if constexpr (KIND == 7) {   // ds_read_b64 x4
            REP16(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n"
                               "ds_read_b64 %0, %4 offset:2048\n ds_read_b64 %1, %4 offset:2560\n ds_read_b64 %2, %4 offset:3072\n ds_read_b64 %3, %4 offset:3584\n s_waitcnt lgkmcnt(0)"
                               : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(ad8));)
        } else if constexpr (KIND == 8) {   // ds_read_b128 x8
            REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                               "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n s_waitcnt lgkmcnt(0)"
                               : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(ad16));)
        } else if constexpr (KIND == 9) {   // mfma 4x4x1 16b f32 x8 (2 accumulators)
            REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %2, %3, %1\n v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %2, %3, %1\n"
                               "v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %2, %3, %1\n v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %2, %3, %1"
                               : "+v"(r0), "+v"(r1) : "v"(w), "v"(x));)
        } else if constexpr (KIND == 10) {  // v_readlane
            REP16(asm volatile("v_readlane_b32 s20, %0, 1\n v_readlane_b32 s21, %0, 2\n v_readlane_b32 s22, %0, 3\n v_readlane_b32 s23, %0, 4\n"
                               "v_readlane_b32 s24, %0, 5\n v_readlane_b32 s25, %0, 6\n v_readlane_b32 s26, %0, 7\n v_readlane_b32 s27, %0, 8"
                               :: "v"(a0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if constexpr (KIND == 11) {  // overlap: 2x ds_read_b128 + 8 fmac on previous data
            REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n"
                               "v_fmac_f32 %5, %13, %14\n v_fmac_f32 %6, %13, %14\n v_fmac_f32 %7, %13, %14\n v_fmac_f32 %8, %13, %14\n"
                               "v_fmac_f32 %9, %13, %14\n v_fmac_f32 %10, %13, %14\n v_fmac_f32 %11, %13, %14\n v_fmac_f32 %12, %13, %14\n s_waitcnt lgkmcnt(0)"
                               : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(ad16),
                                 "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(w), "v"(x));)
        } else if constexpr (KIND == 12) {  // overlap: 2x ds_read_b128 + 8 fmac_dpp
            REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n"
                               "v_fmac_f32_dpp %5, %13, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %6, %13, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %7, %13, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %8, %13, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %9, %13, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %10, %13, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %11, %13, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %12, %13, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n s_waitcnt lgkmcnt(0)"
                               : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(ad16),
                                 "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(w), "v"(x));)
        } else if constexpr (KIND == 15 || KIND == 16) {  // 4 b128 -> 16 mfma on the loaded data
            // KIND 15: read, wait, compute.  KIND 16: the next iteration's reads are issued first.
            for (int u = 0; u < 16; ++u) {
                int o = (it * 16 + u) & 7;
                asm volatile("" : "+v"(o));
                const char* base = lds + ad16 + o * 4096;
                f32x4 n0 = *(const f32x4*)(base), n1 = *(const f32x4*)(base + 1024);
                f32x4 n2 = *(const f32x4*)(base + 2048), n3 = *(const f32x4*)(base + 3072);
                f32x4 c0 = n0, c1 = n1, c2 = n2, c3 = n3;
                if constexpr (KIND == 16) { c0 = r0; c1 = r1; c2 = r2; c3 = r3; }
#define M4(XV) m = __builtin_amdgcn_mfma_f32_4x4x1f32(w, XV.x, m, 0, 0, 0); m2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, XV.y, m2, 0, 0, 0); \
               m = __builtin_amdgcn_mfma_f32_4x4x1f32(w, XV.z, m, 0, 0, 0); m2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, XV.w, m2, 0, 0, 0);
                M4(c0) M4(c1) M4(c2) M4(c3)
#undef M4
                if constexpr (KIND == 16) { r0 = n0; r1 = n1; r2 = n2; r3 = n3; }
            }
        } else if constexpr (KIND == 13) {  // ds_read_b32 broadcast (all lanes same address) x8
            REP16(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                               "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(0));)
        } else if constexpr (KIND == 14) {  // ds_read_b128 broadcast x4 rows distinct per 16 lanes
            REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                               "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n s_waitcnt lgkmcnt(0)"
                               : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"((lane & 15) * 16));)
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r0.x + r1.y + r2.z + r3.w + m.x + m2.y + (float)(d0 + d1 + d2 + d3);
    unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = w1 - w0; }
    (void)q0; (void)q1; (void)q2; (void)q3;
}

template <int KIND>
void run(const char* name, int per_iter, int threads, int blocks) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 2 * blocks * sizeof(unsigned long long));
    hipMalloc(&sink, (size_t)blocks * threads * sizeof(float));
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 65536, 0, d, sink, iters, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 65536, 0, d, sink, iters, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), d, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * per_iter;         // wave-instructions per wave
    const double waves_per_simd = threads / 256.0;
    printf("%-28s thr=%4d  memtime/instr(wave)=%7.2f  wallclk(100MHz)->%.2f ns/instr/wave  event: %.3f ms -> %.2f ns/instr/SIMD\n", name, threads,
           (double)h[0] / n, (double)h[1] * 10.0 / n, ms, ms * 1e6 / (n * waves_per_simd));
    hipFree(d); hipFree(sink);
}

int main() {
    for (int thr : {256, 1024}) {
        run<0>("v_fmac vgpr", 8, thr, 256);
        run<1>("v_fmac sgpr", 8, thr, 256);
        run<2>("v_fmac_dpp row_ror", 8, thr, 256);
        run<3>("v_fmac_dpp row_newbcast", 8, thr, 256);
        run<4>("v_pk_fma_f32", 8, thr, 256);
        run<5>("add_dpp/mov_dpp", 8, thr, 256);
        run<6>("ds_read_b32", 8, thr, 256);
        run<7>("ds_read_b64", 8, thr, 256);
        run<8>("ds_read_b128", 8, thr, 256);
        run<9>("mfma_4x4x1_16b", 8, thr, 256);
        run<10>("v_readlane", 8, thr, 256);
        run<11>("2 b128 + 8 fmac", 10, thr, 256);
        run<12>("2 b128 + 8 fmac_dpp", 10, thr, 256);
        run<15>("4 b128 -> 16 mfma (dep)", 20, thr, 256);
        run<16>("4 b128 -> 16 mfma (pipelined)", 20, thr, 256);
        run<13>("ds_read_b32 bcast", 8, thr, 256);
        run<14>("ds_read_b128 4-row bcast", 8, thr, 256);
    }
    return 0;
}
