// Calibration of rocprofv3 FETCH_SIZE on gfx950 for THIS project's access shape (128-B rows read
// as 8 lanes x 16 B), as MI355X_MICROARCH.md asks before trusting an absolute byte count.
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// Kernels (known bytes in the name's comment, printed by main):
//   stream16      1 GiB read once, 16 B/lane fully coalesced
//   rows_once     1 GiB of 128-B rows, each read once in random order (8 lanes per row)
//   rows_table    a 6.27 MB table (the cfg-2 value tensor's size); 307200 random row reads (39.3 MB
//                 of requests, the forward's tap count), i.e. served by L2 / Infinity Cache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ void stream16(const float4 *p, size_t n, float *sink)
{
    float4 a = make_float4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 12345.f) *sink = a.x;
}
__global__ void rows_once(const float *p, const int *rowidx, size_t nrows, float *sink)
{
    float4 a = make_float4(0, 0, 0, 0);
    const int j = threadIdx.x & 7;
    for (size_t r = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 3; r < nrows; r += ((size_t)gridDim.x * blockDim.x) >> 3) {
        const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)rowidx[r] * 32 + j * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 12345.f) *sink = a.x;
}
__global__ void rows_table(const float *p, const int *rowidx, size_t nreads, float *sink)
{
    float4 a = make_float4(0, 0, 0, 0);
    const int j = threadIdx.x & 7;
    for (size_t r = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 3; r < nreads; r += ((size_t)gridDim.x * blockDim.x) >> 3) {
        const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)rowidx[r] * 32 + j * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 12345.f) *sink = a.x;
}

int main()
{
    const size_t big = 1ull << 30, nrows = big / 128;
    float *buf, *sink; int *idx;
    hipMalloc(&buf, big); hipMemset(buf, 0, big); hipMalloc(&sink, 4);
    std::vector<int> perm(nrows); std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
    hipMalloc(&idx, nrows * 4); hipMemcpy(idx, perm.data(), nrows * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const float4 *>(buf), big / 16, sink);
        hipLaunchKernelGGL(rows_once, dim3(2048), dim3(256), 0, 0, buf, idx, nrows, sink);
    }
    const size_t trows = 48960, treads = 307200;          // 2*3060*8 rows of 128 B = 6.27 MB
    std::vector<int> tr(treads); for (auto &x : tr) x = (int)(rng() % trows);
    int *tidx; hipMalloc(&tidx, treads * 4); hipMemcpy(tidx, tr.data(), treads * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 20; ++rep)
        hipLaunchKernelGGL(rows_table, dim3(600), dim3(256), 0, 0, buf, tidx, treads, sink);
    hipDeviceSynchronize();
    printf("stream16: %zu bytes per launch (+0 index)\nrows_once: %zu bytes of rows + %zu bytes of index per launch\n"
           "rows_table: %zu bytes requested from a %zu-byte table + %zu bytes of index per launch\n",
           big, big, nrows * 4, treads * 128, trows * 128, treads * 4);
    return 0;
}
