// Does v_pk_fma_f32 keep denormal inputs?  fma(f * 2^-149, C, D) for f = 0..2047, scalar against packed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v2f __attribute__((ext_vector_type(2)));
#define C 0x1.921p55f
__global__ void probe(uint32_t *out)
{
    const uint32_t f = threadIdx.x + blockIdx.x * blockDim.x;
    const float d = __builtin_bit_cast(float, 0x1FFF0000u);
    const float x = __builtin_bit_cast(float, f);
    const float s = __builtin_fmaf(x, C, d);
    v2f r;
    const v2f xx = { x, __builtin_bit_cast(float, f ^ 1u) };
    const v2f cc = { C, C }, dd = { d, d };
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(xx), "v"(cc), "v"(dd));
    // the compiler's own form: splat constants (one scalar register, op_sel_hi 0)
    const v2f q = __builtin_elementwise_fma(xx, cc, dd);
    const float s1 = __builtin_fmaf(xx.y, C, d);
    const float rx = r.x, ry = r.y, qx = q.x, qy = q.y; // bit_cast of a vector element reads element 0 whichever is named
    out[8 * f] = __builtin_bit_cast(uint32_t, s);
    out[8 * f + 1] = __builtin_bit_cast(uint32_t, rx);
    out[8 * f + 2] = __builtin_bit_cast(uint32_t, s1);
    out[8 * f + 3] = __builtin_bit_cast(uint32_t, ry);
    out[8 * f + 4] = __builtin_bit_cast(uint32_t, s);
    out[8 * f + 5] = __builtin_bit_cast(uint32_t, qx);
    out[8 * f + 6] = __builtin_bit_cast(uint32_t, s1);
    out[8 * f + 7] = __builtin_bit_cast(uint32_t, qy);
}
int main()
{
    static uint32_t h[8 * 2048]; uint32_t *d;
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 2;
    probe<<<8, 256>>>(d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    int bad = 0;
    for (int f = 0; f < 2048; f++)
        for (int k = 0; k < 4; k++)
            if (h[8 * f + 2 * k] != h[8 * f + 2 * k + 1]) {
                if (bad < 8) printf("f=%d form %d scalar %08x packed %08x\n", f, k, h[8 * f + 2 * k], h[8 * f + 2 * k + 1]);
                bad++;
            }
    printf("mismatches: %d of 2048\n", bad);
    return 0;
}
