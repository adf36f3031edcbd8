// Row-gather microbenchmark: what does the vector memory path charge for — bytes, lines, or lanes?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_width.hip -o tools/micro/gather_width && ./tools/micro/gather_width
// Every wavefront instruction gathers independent random rows of a table:
//   A  128-B rows, 8 lanes x 16 B per row   (fp32 rows of the D = 32 kernels: 8 rows per instruction)
//   B   64-B rows, 8 lanes x  8 B per row   (bf16 rows as the round-1 kernels read them: 8 rows per instruction)
//   C   64-B rows, 4 lanes x 16 B per row   (bf16 rows, 16 rows per instruction)
//   D  128-B rows read as 4 lanes x 32 B (two 16-B loads per lane): 16 rows per pair of instructions
// for a table that fits one XCD's L2 (2 MB), the 8 L2s together (24 MB) and only the Infinity Cache (192 MB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int ROWB, int LANES, int VEC>      // bytes per row, lanes per row, bytes per lane per load
__global__ __launch_bounds__(256) void gather(const unsigned char *__restrict__ table, unsigned rows, int iters,
                                              float *__restrict__ sink)
{
    constexpr int LPR = ROWB / (LANES * VEC); // loads per lane per row
    constexpr int U = 16 / LPR;               // rows in flight per lane group: 16 loads per lane, like the forward kernel
    const int lane = threadIdx.x & 63, grp = lane / LANES, j = lane % LANES;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned s = (wave * 64 + grp) * 2654435761u + 12345u;       // one random stream per lane group (no index loads)
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { s = s * 1664525u + 1013904223u; r[u] = (s >> 8) & (rows - 1); }      // rows is a power of two
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < LPR; ++k) {
                const unsigned char *p = table + (long long)r[u] * ROWB + (k * LANES + j) * VEC;
                if (VEC == 16) { const float4 v = *reinterpret_cast<const float4 *>(p); acc += (v.x + v.y) + (v.z + v.w); }
                else { const float2 v = *reinterpret_cast<const float2 *>(p); acc += v.x + v.y; }
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int ROWB, int LANES, int VEC>
static void run(const char *name, size_t table_bytes, int nblocks, int iters)
{
    size_t rows = 1; while (rows * 2 * ROWB <= table_bytes) rows *= 2;
    table_bytes = rows * ROWB;
    constexpr int RPI = 64 / LANES, LPR = ROWB / (LANES * VEC), U = 16 / LPR;
    unsigned char *t; float *sink;
    CK(hipMalloc(&t, table_bytes)); CK(hipMemset(t, 1, table_bytes));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((gather<ROWB, LANES, VEC>), dim3(nblocks), dim3(256), 0, 0, t, (unsigned)rows, iters, sink);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((gather<ROWB, LANES, VEC>), dim3(nblocks), dim3(256), 0, 0, t, (unsigned)rows, iters, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double nrows = (double)nblocks * 4 * iters * U * RPI;
    printf("  %-44s table %6.1f MB: %8.1f us  %7.2f G rows/s  %6.2f TB/s of row bytes\n", name, table_bytes / 1e6, ms * 1e3,
           nrows / ms / 1e6, nrows * ROWB / ms / 1e9);
    CK(hipFree(t)); CK(hipFree(sink));
}

int main()
{
    const int nblocks = 4096, iters = 16;
    for (size_t mb : {2, 24, 192}) {
        const size_t b = mb << 20;
        run<128, 8, 16>("A 128-B rows, 8 lanes x 16 B", b, nblocks, iters);
        run<64, 8, 8>("B  64-B rows, 8 lanes x 8 B", b, nblocks, iters);
        run<64, 4, 16>("C  64-B rows, 4 lanes x 16 B", b, nblocks, iters);
        run<128, 4, 16>("D 128-B rows, 4 lanes x 2 x 16 B", b, nblocks, iters);
    }
    return 0;
}
