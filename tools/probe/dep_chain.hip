// Latency probe: one wave per workgroup walks a first-order recursion y = d + A * y (separate
// multiply and add, as the DC blocker of the panadapter prefilter has it), with 0, 2 or 3 independent
// instructions per step beside it.  Prints ns per step; build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)

template <int EXTRA>
__global__ __launch_bounds__(64) void chain(const int *__restrict__ in, float *__restrict__ out, int n, float a)
{
    float y = (float)threadIdx.x, acc = 0.0f;
    int prev = in[threadIdx.x];
    for (int i = 0; i < n; i += 16) {
        int v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = prev + k * (int)threadIdx.x + i; // stands for the unpacked sample
#pragma unroll
        for (int k = 0; k < 16; k++) {
            float d;
            if (EXTRA == 0) d = 1.0f;
            else if (EXTRA == 2) d = (float)(v[k] - prev);          // sub + cvt
            else d = (float)((v[k] >> 3) - prev);                    // shift + sub + cvt
            prev = v[k];
            const float m = a * y;
            y = d + m;
            acc += (EXTRA == 0 && k == 15) ? y : 0.0f;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = y + acc + (float)prev;
}

template <int EXTRA>
static void run(const char *name, int grid)
{
    int *in; float *out;
    hipMalloc(&in, 64 * sizeof(int)); hipMemset(in, 0, 64 * sizeof(int));
    hipMalloc(&out, grid * 64 * sizeof(float));
    const int n = 1 << 18;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain<EXTRA>, dim3(grid), dim3(64), 0, 0, in, out, n, 0.99951171875f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-28s grid %4d: %.2f ns per step\n", name, grid, best * 1e6f / n);
    hipFree(in); hipFree(out);
}

int main()
{
    for (int grid : { 1, 8, 256, 4096 }) {
        run<0>("mul+add chain only", grid);
        run<2>("chain + sub + cvt", grid);
        run<3>("chain + shift + sub + cvt", grid);
    }
    return 0;
}
