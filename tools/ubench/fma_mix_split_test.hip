#include <hip/hip_runtime.h>
__global__ void k(const float* x, unsigned* out, float s) {
    float x0 = x[threadIdx.x * 2], x1 = x[threadIdx.x * 2 + 1];
    unsigned hi = 0, lo = 0;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x0), "s"(s));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "s"(s));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x0), "s"(s), "v"(hi));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "s"(s), "v"(hi));
    out[threadIdx.x * 2] = hi; out[threadIdx.x * 2 + 1] = lo;
}
int main() {
    float hx[128]; unsigned ho[128];
    for (int i = 0; i < 128; ++i) hx[i] = (i % 2 ? -1.f : 1.f) * (0.001f + 0.0137f * i) * (i % 7 == 0 ? 1e-3f : 1.f);
    float* dx; unsigned* dout; hipMalloc(&dx, 512); hipMalloc(&dout, 512);
    hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dx, dout, 4096.f);
    hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost);
    int bad = 0; double worst = 0;
    for (int t = 0; t < 64; ++t) for (int j = 0; j < 2; ++j) {
        const float v = hx[t * 2 + j] * 4096.f;
        const unsigned short hb = (ho[t * 2] >> (16 * j)) & 0xffff, lb = (ho[t * 2 + 1] >> (16 * j)) & 0xffff;
        _Float16 h, l; __builtin_memcpy(&h, &hb, 2); __builtin_memcpy(&l, &lb, 2);
        const double err = std::abs((double)(float)h + (double)(float)l - (double)v) / std::abs((double)v);
        worst = err > worst ? err : worst;
        if (err > 1.0 / (1 << 21)) { if (bad < 5) printf("t=%d j=%d v=%g h=%g l=%g\n", t, j, v, (float)h, (float)l); ++bad; }
    }
    printf("fma_mix split: worst relative error %.3e (2^-22 = %.3e), bad %d\n", worst, 1.0 / (1 << 22), bad);
    return bad;
}
