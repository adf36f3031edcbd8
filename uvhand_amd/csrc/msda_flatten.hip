// Transformer input assembly: the flatten block of the reference's DeformableTransformer.forward
// (UVHand models/arctic_transformer.py:162-173) as ONE kernel per direction.
//
// Per level l the reference builds  src_l.flatten(2).transpose(1, 2)  ([N,C,H,W] -> [N,HW,C], a strided VIEW) and
// pos_l^T + level_embed[l], then torch.cat's the levels along the pixel axis: the cat is the kernel that actually moves
// the data, reading every input with a stride of H*W floats between consecutive channels (uncoalesced) — twice, for src
// and for pos + level embedding, plus the add.  Here:
//
//   forward   for every (level, batch element, 64-pixel x 64-channel tile): read the tile along the pixel axis (coalesced
//             256-B runs of the NCHW input), transpose it through LDS (64 x 65 floats, conflict-free), write it along the
//             channel axis as float4 (coalesced runs of the [N, S, C] output at row level_start + pixel); the positional
//             tile gets level_embed[l][c] added on the way out.  Both tensors in the same launch.
//   backward  the inverse copy: [N, S, C] gradient rows back into per-level [N, C, H, W] gradients (same tiles, roles of
//             the two axes swapped).  The level-embedding gradient is a column sum the caller takes over the level's rows.
//
// HBM bound: every byte read once and written once (algorithmic bytes 16 * N * S * C for the two tensors).
#include "msda_common.h"
#include "msda_launch.h"

namespace msda {

constexpr int kFlTile = 64;
constexpr int kFlBlock = 256;

__global__ __launch_bounds__(kFlBlock) void flatten_levels_kernel(const FlattenPlan plan, int N, int C, int S,
                                                                  const float *__restrict__ level_embed,
                                                                  float *__restrict__ src_out, float *__restrict__ pos_out,
                                                                  int unflatten, float *__restrict__ embed_partial)
{
    __shared__ float tile[kFlTile][kFlTile + 1];
    // block -> (level, batch element, pixel tile, channel tile); the level from the prefix table (uniform)
    int l = 0;
    while (l + 1 < plan.L && (int)blockIdx.x >= plan.first_block[l + 1]) ++l;
    const int HW = plan.hw[l], lstart = plan.start[l];
    const int tiles_c = (C + kFlTile - 1) / kFlTile, tiles_p = (HW + kFlTile - 1) / kFlTile;
    int rest = (int)blockIdx.x - plan.first_block[l];
    const int tc = rest % tiles_c; rest /= tiles_c;
    const int tp = rest % tiles_p;
    const int n = rest / tiles_p;
    const int p0 = tp * kFlTile, c0 = tc * kFlTile;
    const int tid = threadIdx.x;
    for (int which = 0; which < 2; ++which) {
        // NCHW-side tensor of this level and the flattened one
        float *nchw = which == 0 ? plan.src[l] : plan.pos[l];
        float *flat = which == 0 ? src_out : pos_out;
        if (nchw == nullptr || flat == nullptr) continue;
        if (which == 1) __syncthreads();                               // the tile buffer is reused
        const bool vec = (HW & 3) == 0;                                // rows of the NCHW tensor are 16-byte aligned
        if (!unflatten) {
            // read along pixels (coalesced runs of the NCHW input) into the LDS tile [channel][pixel]
            if (vec) {
                const int x4 = (tid & 15) * 4, r = tid >> 4;             // thread: channels c0 + r + 16k, pixels p0 + x4 .. +3
                float4 v[kFlTile / 16];
#pragma unroll
                for (int k = 0; k < kFlTile / 16; ++k) {
                    const int c = c0 + r + 16 * k, p = p0 + x4;
                    v[k] = (c < C && p < HW) ? *reinterpret_cast<const float4 *>(nchw + ((long long)n * C + c) * HW + p)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < kFlTile / 16; ++k) {
                    float *t = &tile[r + 16 * k][x4];
                    t[0] = v[k].x; t[1] = v[k].y; t[2] = v[k].z; t[3] = v[k].w;
                }
            } else {
                const int x = tid & 63, r = tid >> 6;
                float v[kFlTile / 4];
#pragma unroll
                for (int k = 0; k < kFlTile / 4; ++k) {
                    const int c = c0 + r + 4 * k, p = p0 + x;
                    v[k] = (c < C && p < HW) ? nchw[((long long)n * C + c) * HW + p] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < kFlTile / 4; ++k) tile[r + 4 * k][x] = v[k];
            }
            __syncthreads();
            // write along channels: thread (row, q) takes pixel p0 + row + 16k, channels c0 + 4q .. 4q+3
            const int q = tid & 15, row = tid >> 4;
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            if (which == 1 && level_embed != nullptr && c0 + 4 * q < C)
                e = *reinterpret_cast<const float4 *>(level_embed + (long long)l * C + c0 + 4 * q);
#pragma unroll
            for (int k = 0; k < kFlTile / 16; ++k) {
                const int p = p0 + row + 16 * k, c = c0 + 4 * q;
                if (p < HW && c < C) {                                  // C is a multiple of 4 (host check)
                    const float4 v = make_float4(tile[4 * q][row + 16 * k] + e.x, tile[4 * q + 1][row + 16 * k] + e.y,
                                                 tile[4 * q + 2][row + 16 * k] + e.z, tile[4 * q + 3][row + 16 * k] + e.w);
                    *reinterpret_cast<float4 *>(flat + ((long long)n * S + lstart + p) * C + c) = v;
                }
            }
        } else {
            // inverse: read rows of the flattened gradient along channels, write the NCHW gradient along pixels
            const int q = tid & 15, row = tid >> 4;
            float4 v[kFlTile / 16];
#pragma unroll
            for (int k = 0; k < kFlTile / 16; ++k) {
                const int p = p0 + row + 16 * k, c = c0 + 4 * q;
                v[k] = (p < HW && c < C) ? *reinterpret_cast<const float4 *>(flat + ((long long)n * S + lstart + p) * C + c)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < kFlTile / 16; ++k) {
                tile[4 * q][row + 16 * k] = v[k].x; tile[4 * q + 1][row + 16 * k] = v[k].y;
                tile[4 * q + 2][row + 16 * k] = v[k].z; tile[4 * q + 3][row + 16 * k] = v[k].w;
            }
            __syncthreads();
            if (which == 1 && embed_partial != nullptr && tid < kFlTile) {
                // level-embedding gradient: this tile's column sums over its 64 pixels (rows past HW hold zeros), one
                // partial per (block, channel); level_embed_reduce_kernel adds them in block order
                float t = 0.f;
#pragma unroll 8
                for (int x = 0; x < kFlTile; ++x) t += tile[tid][x];
                embed_partial[(long long)blockIdx.x * kFlTile + tid] = t;
            }
            if (vec) {
                const int x4 = (tid & 15) * 4, r = tid >> 4;
#pragma unroll
                for (int k = 0; k < kFlTile / 16; ++k) {
                    const int c = c0 + r + 16 * k, p = p0 + x4;
                    const float *t = &tile[r + 16 * k][x4];
                    if (c < C && p < HW)
                        *reinterpret_cast<float4 *>(nchw + ((long long)n * C + c) * HW + p) = make_float4(t[0], t[1], t[2], t[3]);
                }
            } else {
                const int x = tid & 63, r = tid >> 6;
#pragma unroll
                for (int k = 0; k < kFlTile / 4; ++k) {
                    const int c = c0 + r + 4 * k, p = p0 + x;
                    if (c < C && p < HW) nchw[((long long)n * C + c) * HW + p] = tile[r + 4 * k][x];
                }
            }
        }
    }
}

// grad_level_embed[l][c] = sum over the level's (batch element, pixel tile) blocks of their partial column sums, 4 chains
// per column combined in a fixed order.  grid = (channel tiles, L).
__global__ __launch_bounds__(256) void level_embed_reduce_kernel(const FlattenPlan plan, int N, int C, const float *__restrict__ partial,
                                                                float *__restrict__ grad_embed)
{
    __shared__ float red[4][kFlTile];
    const int l = (int)blockIdx.y, tc = (int)blockIdx.x, j = (int)threadIdx.x & 63, part = (int)threadIdx.x >> 6;
    const int tiles_c = (C + kFlTile - 1) / kFlTile, tiles_p = (plan.hw[l] + kFlTile - 1) / kFlTile;
    const int nblk = N * tiles_p;
    float t = 0.f;
    for (int i = part; i < nblk; i += 4)
        t += partial[((long long)plan.first_block[l] + (long long)i * tiles_c + tc) * kFlTile + j];
    red[part][j] = t;
    __syncthreads();
    if (part == 0 && tc * kFlTile + j < C)
        grad_embed[(long long)l * C + tc * kFlTile + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
}

static long long flatten_blocks(FlattenPlan &plan, int N, int C)
{
    const int tiles_c = (C + kFlTile - 1) / kFlTile;
    long long blocks = 0;
    for (int l = 0; l < plan.L; ++l) {
        plan.first_block[l] = (int)blocks;
        blocks += (long long)N * ((plan.hw[l] + kFlTile - 1) / kFlTile) * tiles_c;
        if (blocks > 0x7fffffffLL) return -1;
    }
    return blocks;
}

size_t unflatten_workspace_bytes(const FlattenPlan &plan_in, int N, int C)
{
    FlattenPlan plan = plan_in;
    const long long blocks = flatten_blocks(plan, N, C);
    return blocks <= 0 ? 0 : (size_t)blocks * kFlTile * sizeof(float);
}

int launch_flatten_levels(const FlattenPlan &plan_in, int N, int C, int S, const float *level_embed, float *src_flat,
                          float *pos_flat, bool unflatten, hipStream_t stream, float *grad_level_embed, float *workspace)
{
    FlattenPlan plan = plan_in;
    const long long blocks = flatten_blocks(plan, N, C);
    if (blocks < 0) return set_error(MSDA_ERR_ARGUMENT, "msda flatten: too many tiles");
    if (blocks == 0) return MSDA_OK;
    const bool want_embed = unflatten && grad_level_embed != nullptr && pos_flat != nullptr;
    if (want_embed && workspace == nullptr) return set_error(MSDA_ERR_ARGUMENT, "msda unflatten: workspace required for grad_level_embed");
    hipLaunchKernelGGL(flatten_levels_kernel, dim3((unsigned)blocks), dim3(kFlBlock), 0, stream, plan, N, C, S, level_embed, src_flat,
                       pos_flat, unflatten ? 1 : 0, want_embed ? workspace : nullptr);
    if (int rc = check_launch(unflatten ? "msda unflatten levels" : "msda flatten levels")) return rc;
    if (want_embed) {
        hipLaunchKernelGGL(level_embed_reduce_kernel, dim3((C + kFlTile - 1) / kFlTile, plan.L), dim3(256), 0, stream, plan, N, C,
                           workspace, grad_level_embed);
        return check_launch("msda level-embedding gradient");
    }
    return MSDA_OK;
}

}  // namespace msda
