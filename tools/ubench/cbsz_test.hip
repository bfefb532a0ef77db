// Semantics check: v_mfma_f32_4x4x1_16b_f32 with cbsz = 2 / abid = s broadcasts the A values of block
// (4 * (b / 4) + s) to the 4 blocks of its set; v_add_u32_dpp quad_perm:[s,s,s,s] broadcasts a lane of a quad.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int S> __device__ f32x4 mf(float a, float b) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 2, S, 0);
}
template <int S> __device__ unsigned addq(unsigned i, unsigned base) {
    unsigned r;
    asm volatile("v_add_u32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(i), "v"(base), "n"(S));
    return r;
}
__global__ void k(const float* a, const float* b, float* d, unsigned* u) {
    const int l = threadIdx.x;
    f32x4 r0 = mf<0>(a[l], b[l]), r1 = mf<1>(a[l], b[l]), r2 = mf<2>(a[l], b[l]), r3 = mf<3>(a[l], b[l]);
    for (int i = 0; i < 4; ++i) { d[(0 * 4 + i) * 64 + l] = r0[i]; d[(1 * 4 + i) * 64 + l] = r1[i]; d[(2 * 4 + i) * 64 + l] = r2[i]; d[(3 * 4 + i) * 64 + l] = r3[i]; }
    u[0 * 64 + l] = addq<0>(l * 100, 7); u[1 * 64 + l] = addq<1>(l * 100, 7); u[2 * 64 + l] = addq<2>(l * 100, 7); u[3 * 64 + l] = addq<3>(l * 100, 7);
}
int main() {
    float ha[64], hb[64], hd[16 * 64]; unsigned hu[4 * 64];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 1000.f + 3 * l; }
    float *a, *b, *d; unsigned* u;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, sizeof(hd)); hipMalloc(&u, sizeof(hu));
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, b, d, u);
    hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost); hipMemcpy(hu, u, sizeof(hu), hipMemcpyDeviceToHost);
    int bad = 0, badu = 0;
    for (int s = 0; s < 4; ++s) for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) {
        const int src_block = (l >> 4) * 4 + s;
        const float want = ha[src_block * 4 + i] * hb[l];
        if (hd[(s * 4 + i) * 64 + l] != want) { if (bad < 8) printf("mfma s=%d i=%d l=%d got %g want %g\n", s, i, l, hd[(s * 4 + i) * 64 + l], want); ++bad; }
    }
    for (int s = 0; s < 4; ++s) for (int l = 0; l < 64; ++l) {
        const unsigned want = ((l & ~3) + s) * 100 + 7;
        if (hu[s * 64 + l] != want) { if (badu < 8) printf("dpp s=%d l=%d got %u want %u\n", s, l, hu[s * 64 + l], want); ++badu; }
    }
    printf("cbsz/abid mismatches: %d   dpp mismatches: %d\n", bad, badu);
    return bad || badu;
}
