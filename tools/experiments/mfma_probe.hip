// Probe of v_mfma_i32_32x32x32_i8 operand layouts on gfx950 (hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_probe.hip): checks
//   C[i][j] = sum_{h<2} sum_{t<16} A(lane i + 32 h, byte t) * B(lane j + 32 h, byte t)
// with C element (i, j) in lane j + 32 * ((i / 4) % 2), register 4 * (i / 8) + i % 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const v4i *a, const v4i *b, v16i *c)
{
    v16i acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    c[threadIdx.x] = acc;
}
int main()
{
    int8_t ha[64][16], hb[64][16];
    int hc[64][16];
    srand(1);
    for (int l = 0; l < 64; l++)
        for (int t = 0; t < 16; t++) { ha[l][t] = (int8_t)(rand() % 256 - 128); hb[l][t] = (int8_t)(rand() % 256 - 128); }
    void *da, *db, *dc;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, sizeof hc);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k<<<1, 64>>>((const v4i *)da, (const v4i *)db, (v16i *)dc);
    if (hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
    int bad = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            int want = 0;
            for (int h = 0; h < 2; h++)
                for (int t = 0; t < 16; t++) want += (int)ha[i + 32 * h][t] * (int)hb[j + 32 * h][t];
            const int lane = j + 32 * ((i / 4) % 2), reg = 4 * (i / 8) + i % 4;
            if (hc[lane][reg] != want) bad++;
        }
    printf("mismatches: %d of 1024\n", bad);
    return bad ? 1 : 0;
}
