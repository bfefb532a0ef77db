// Feasibility probe for the mixed dense / sparse hop kernel: the DENSE part alone.
// A wave owns 16 rows x 16 features; a step = one v_mfma_f32_16x16x4_f32 with A = the weights of
// 16 rows x 4 columns (one resident VGPR) and B = 4 staged rows x 16 features read by ONE
// ds_read_b32 per lane (lane (k, n) reads float perm(n) of staged row k: rows are 256 B apart, so the
// four rows of an instruction collide on the same banks).  16 waves per workgroup, one per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o dense_part dense_part.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// VAR bit0: B operands from LDS (else constants)   bit1: staged rows padded to 272 B (no bank conflict)
template <int STEPS, int VAR>
__global__ __launch_bounds__(1024) void k(float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 448 * 68; i += blockDim.x) lds[i] = 1e-3f * (i % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63, kq = lane >> 4, n = lane & 15, wave = threadIdx.x >> 6;
    const int fq = wave & 3;
    const int perm = 4 * (n & 3) + (n >> 2);
    const int stride = (VAR & 2) ? 68 : 64;              // floats per staged row
    unsigned addr[STEPS];
    float w[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const unsigned row = (unsigned)((s * 29 + kq * 7 + wave * 13) % 448);
        addr[s] = (row * stride + 16 * fq + perm) * 4;
        w[s] = 1e-3f * (s + lane);
        asm volatile("" : "+v"(addr[s]), "+v"(w[s]));
    }
    f32x4 acc = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            float b;
            if (VAR & 1) b = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + addr[s]);
            else b = w[(s + 1) % STEPS];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], b, acc, 0, 0, 0);
        }
        asm volatile("" : "+v"(acc));
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int STEPS, int VAR>
void run(const char* name) {
    const int blocks = 256, iters = 4000;
    float* sink; (void)hipMalloc(&sink, (size_t)blocks * 1024 * 4);
    (void)hipFuncSetAttribute((const void*)k<STEPS, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 448 * 68 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<STEPS, VAR>), dim3(blocks), dim3(1024), 448 * 68 * 4, 0, sink, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<STEPS, VAR>), dim3(blocks), dim3(1024), 448 * 68 * 4, 0, sink, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 16 * iters * STEPS;
    printf("%-52s %.3f ms  %.1f TF/s  (%.1f ns per MFMA per SIMD)\n", name, ms, mfma * 2048 / (ms * 1e-3) / 1e12,
           ms * 1e6 / (mfma / 1024));
    (void)hipFree(sink);
}

int main() {
    run<17, 0>("17 dense steps, operands in registers");
    run<17, 1>("17 dense steps, ds_read_b32 per MFMA (256-B rows)");
    run<17, 3>("17 dense steps, ds_read_b32 per MFMA (272-B rows)");
    run<34, 1>("34 dense steps, ds_read_b32 per MFMA (256-B rows)");
    return 0;
}
