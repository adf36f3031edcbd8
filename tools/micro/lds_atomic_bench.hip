// Microbenchmark: LDS float-atomic throughput on gfx950 for the access shapes the grad_value
// tile kernel produces.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_bench.hip -o lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ITERS = 256;

// mode 0: ds_add_f32, mode 1: ds_add_u32, mode 2: non-atomic read+add+write (racy, timing only)
template <int MODE>
__global__ __launch_bounds__(512) void k(const int *__restrict__ idx, float *out, int nrows, int rs)
{
    extern __shared__ float tile[];
    for (int i = threadIdx.x; i < nrows * rs; i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    const int *my = idx + (blockIdx.x * blockDim.x + threadIdx.x) * ITERS;
    for (int it = 0; it < ITERS; it += 4) {
        int a0 = my[it], a1 = my[it + 1], a2 = my[it + 2], a3 = my[it + 3];
        if (MODE == 0) {
            atomicAdd(&tile[a0], 1.0f); atomicAdd(&tile[a1], 1.0f); atomicAdd(&tile[a2], 1.0f); atomicAdd(&tile[a3], 1.0f);
        } else if (MODE == 1) {
            atomicAdd(reinterpret_cast<unsigned *>(&tile[a0]), 1u); atomicAdd(reinterpret_cast<unsigned *>(&tile[a1]), 1u);
            atomicAdd(reinterpret_cast<unsigned *>(&tile[a2]), 1u); atomicAdd(reinterpret_cast<unsigned *>(&tile[a3]), 1u);
        } else {
            tile[a0] += 1.0f; tile[a1] += 1.0f; tile[a2] += 1.0f; tile[a3] += 1.0f;
        }
    }
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < nrows * rs; i += blockDim.x) s += tile[i];
    if (s == -1.f) out[0] = s;
}

int main()
{
    const int blocks = 512, threads = 512;
    const size_t n = (size_t)blocks * threads * ITERS;
    std::vector<int> h(n);
    int *d; float *o;
    hipMalloc(&d, n * sizeof(int)); hipMalloc(&o, 4);
    struct Pat { const char *name; int nrows, rs, lpp; bool random; };
    // lpp lanes x 4 floats per point; row = random pixel of nrows; address = row*rs + j*4 + c
    Pat pats[] = {{"contiguous (lane i -> i)", 64, 32, 0, false},
                  {"LPP=8 rs=33 rows=576", 576, 33, 8, true}, {"LPP=8 rs=33 rows=36", 36, 33, 8, true},
                  {"LPP=2 rs=9 rows=576", 576, 9, 2, true},  {"LPP=1 rs=5 rows=144", 144, 5, 1, true},
                  {"LPP=1 rs=5 rows=36", 36, 5, 1, true},    {"LPP=1 rs=5 rows=2000", 2000, 5, 1, true},
                  {"all lanes same address", 1, 1, -1, false}};
    for (auto &p : pats) {
        srand(1);
        for (int b = 0; b < blocks; ++b)
            for (int t = 0; t < threads; ++t)
                for (int it = 0; it < ITERS; ++it) {
                    int a;
                    const int lane = t & 63;
                    if (p.lpp == 0) a = lane + 64 * ((it / 4) % (p.nrows * p.rs / 64));
                    else if (p.lpp < 0) a = 0;
                    else {
                        // points are per (wave, it/4, lane/lpp): deterministic hash -> row
                        unsigned key = (unsigned)(((b * 8 + (t >> 6)) * ITERS + it / 4) * 64 + lane / p.lpp);
                        key = key * 2654435761u; key ^= key >> 15; key *= 2246822519u; key ^= key >> 13;
                        a = (int)(key % p.nrows) * p.rs + (lane % p.lpp) * 4 + (it & 3);
                    }
                    h[((size_t)b * threads + t) * ITERS + it] = a;
                }
        hipMemcpy(d, h.data(), n * sizeof(int), hipMemcpyHostToDevice);
        const size_t lds = (size_t)p.nrows * p.rs * 4 + 64;
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), lds, 0, d, o, p.nrows, p.rs);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), lds, 0, d, o, p.nrows, p.rs);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), lds, 0, d, o, p.nrows, p.rs);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // wave-instructions per CU: blocks*8 waves*ITERS / 256 CUs
            const double winstr_per_cu = (double)blocks * 8 * ITERS / 256.0;
            printf("%-28s mode %d: %8.1f us  -> %6.1f ns per wave-instr per CU (~%.0f cyc @2.4GHz)\n", p.name, mode,
                   ms * 1e3, ms * 1e6 / winstr_per_cu, ms * 1e6 / winstr_per_cu * 2.4);
        }
    }
    return 0;
}
