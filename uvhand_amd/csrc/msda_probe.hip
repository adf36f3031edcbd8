// msda_probe_row_gather (include/msda.h): the measurement behind bench.py's second ceiling.  The sampling kernels gather
// 128-byte rows (one (pixel, head) slice of value / grad_out, 8 lanes x 16 B) at addresses known only at run time; on
// cache-resident data what bounds them is not HBM bytes but the rate at which a CU's vector memory path serves independent
// ROW REQUESTS (profiles/r02_notes.md section 2).  This kernel issues exactly that access pattern — every 8-lane group gathers
// independent pseudo-random rows of a table, sixteen loads in flight per lane like the forward kernel — and nothing else, so
// rows / time on a table of the workload's size is the ceiling a gather kernel could reach on this box.  No reference
// counterpart (the reference publishes no performance model); not used by any product path.
#include "msda_launch.h"

namespace msda {

template <int ROWB>                             // bytes per row: 128 (fp32 rows: 8 lanes x 16 B) or 64 (bf16 rows: 8 lanes x 8 B)
__global__ __launch_bounds__(256) void probe_row_gather_kernel(const unsigned char *__restrict__ table, unsigned row_mask, int iters,
                                                               float *__restrict__ sink)
{
    constexpr int U = 16;                                            // rows in flight per lane group
    const int lane = threadIdx.x & 63, grp = lane >> 3, j = lane & 7;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned s = (wave * 8 + grp) * 2654435761u + 12345u;            // one random stream per lane group: no index loads
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { s = s * 1664525u + 1013904223u; r[u] = (s >> 8) & row_mask; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned char *p = table + (size_t)r[u] * ROWB + j * (ROWB / 8);
            if (ROWB == 128) { const float4 v = *reinterpret_cast<const float4 *>(p); acc += (v.x + v.y) + (v.z + v.w); }
            else { const float2 v = *reinterpret_cast<const float2 *>(p); acc += v.x + v.y; }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;                            // keeps the loads alive; never true for a zero / finite table
}

}  // namespace msda

extern "C" int msda_probe_row_gather(const void *table, unsigned long long table_bytes, int row_bytes, int blocks, int iters,
                                     float *sink, unsigned long long *rows_gathered, msda_stream_t stream)
{
    if (!table || !sink || blocks <= 0 || iters <= 0 || (row_bytes != 128 && row_bytes != 64) || table_bytes < (unsigned long long)row_bytes)
        return msda::set_error(MSDA_ERR_ARGUMENT, "msda_probe_row_gather: need a table of at least one row, row_bytes 128 or 64, blocks, iters > 0");
    unsigned long long rows = 1;
    while (rows * 2 * row_bytes <= table_bytes && rows < (1ull << 31)) rows *= 2;          // a power of two: the mask below
    const unsigned mask = (unsigned)(rows - 1);
    if (row_bytes == 128)
        hipLaunchKernelGGL((msda::probe_row_gather_kernel<128>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const unsigned char *>(table), mask, iters, sink);
    else
        hipLaunchKernelGGL((msda::probe_row_gather_kernel<64>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const unsigned char *>(table), mask, iters, sink);
    if (rows_gathered) *rows_gathered = (unsigned long long)blocks * 4 * 8 * 16 * (unsigned long long)iters;
    return msda::check_launch("msda_probe_row_gather");
}
