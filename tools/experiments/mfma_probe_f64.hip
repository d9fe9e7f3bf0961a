// Probe of v_mfma_f64_16x16x4_f64 on gfx950 (hipcc --offload-arch=gfx950 -O2 -o mfma_probe_f64 mfma_probe_f64.hip):
//   D[i][j] = sum_{k<4} A(lane i + 16 k) * B(lane j + 16 k) + C, D element (i, j) in lane j + 16 * (i / 4),
//   register pair i % 4; exact for integer operands whose partial sums stay below 2^53; and what one
//   instruction costs per SIMD with 1..4 waves issuing them back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double *a, const double *b, v4d *c)
{
    v4d acc = {1.0, 1.0, 1.0, 1.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    c[threadIdx.x] = acc;
}
__global__ void rate(const double *a, const double *b, v4d *c, int n)
{
    const double x = a[threadIdx.x & 63], y = b[threadIdx.x & 63];
    v4d acc0 = {x, 0, 0, 0}, acc1 = {0, x, 0, 0}, acc2 = {0, 0, x, 0}, acc3 = {0, 0, 0, x};
    for (int i = 0; i < n; i++) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc3, 0, 0, 0);
    }
    c[threadIdx.x & 63] = acc0 + acc1 + acc2 + acc3;
}
int main()
{
    double ha[64], hb[64];
    double hc[64][4];
    srand(1);
    for (int l = 0; l < 64; l++) { ha[l] = (double)((long long)rand() << 5); hb[l] = (double)(rand() % 65536 - 32768); }
    void *da, *db, *dc;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, sizeof hc);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k<<<1, 64>>>((const double *)da, (const double *)db, (v4d *)dc);
    if (hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
    int bad = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            long long want = 1;
            for (int kk = 0; kk < 4; kk++) want += (long long)ha[i + 16 * kk] * (long long)hb[j + 16 * kk];
            // where is it?
            int found = -1;
            for (int l = 0; l < 64 && found < 0; l++)
                for (int r = 0; r < 4; r++)
                    if (hc[l][r] == (double)want) { found = l * 4 + r; break; }
            const int lane = j + 16 * (i / 4), reg = i % 4;
            if (found != lane * 4 + reg) {
                if (bad < 12) printf("D[%d][%d] found at lane %d reg %d (assumed lane %d reg %d)\n", i, j, found / 4, found % 4, lane, reg);
                bad++;
            }
        }
    printf("layout / exactness mismatches against the assumed layout: %d of 256\n", bad);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 4; waves++) {
        const int n = 20000;
        rate<<<256, 256 * waves>>>((const double *)da, (const double *)db, (v4d *)dc, 100); // warm
        hipEventRecord(e0);
        rate<<<256, 256 * waves>>>((const double *)da, (const double *)db, (v4d *)dc, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // 256 workgroups on 256 CUs (one each if the dispatcher spreads them), waves per SIMD = waves
        printf("%d wave(s) per SIMD: %.3f ms for %d instructions per wave = %.1f ns per instruction per SIMD\n", waves, ms,
               4 * n, ms * 1e6 / (4.0 * n * waves));
    }
    return bad ? 1 : 0;
}
