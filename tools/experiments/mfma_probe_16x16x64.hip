// Probe of v_mfma_i32_16x16x64_i8 operand layouts on gfx950
// (hipcc --offload-arch=gfx950 -O2 -o mfma_probe_16x16x64 mfma_probe_16x16x64.hip): checks
//   C[i][j] = sum_{g<4} sum_{t<16} A(lane i + 16 g, byte t) * B(lane j + 16 g, byte t)
// with C element (i, j) in lane j + 16 * (i / 4), register i % 4: the first operand carries the
// rows, the second the columns, and both read their 64 K slots as (lane / 16, byte).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const v4i *a, const v4i *b, v4i *c)
{
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    c[threadIdx.x] = acc;
}
int main()
{
    int8_t ha[64][16], hb[64][16];
    int hc[64][4];
    srand(1);
    for (int l = 0; l < 64; l++)
        for (int t = 0; t < 16; t++) { ha[l][t] = (int8_t)(rand() % 256 - 128); hb[l][t] = (int8_t)(rand() % 256 - 128); }
    void *da, *db, *dc;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, sizeof hc);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k<<<1, 64>>>((const v4i *)da, (const v4i *)db, (v4i *)dc);
    if (hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
    int bad = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            int want = 0;
            for (int g = 0; g < 4; g++)
                for (int t = 0; t < 16; t++) want += (int)ha[i + 16 * g][t] * (int)hb[j + 16 * g][t];
            const int lane = j + 16 * (i / 4), reg = i % 4;
            if (hc[lane][reg] != 2 * want) bad++;
        }
    printf("mismatches: %d of 256\n", bad);
    return bad ? 1 : 0;
}
