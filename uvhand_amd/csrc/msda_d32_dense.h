// D = 32 backward, role B for the COARSE levels of the pyramid: grad_value of a level of at most 64 pixels as a dense
// product on the matrix cores.  Device code only; included by msda_d32.hip after msda_d32_value.h.
//
// Why: every level receives the same number of sampling points, so a 4x4 or 7x7 level collects as many taps as the 28x28
// one — on sixteen or forty-nine rows.  The sort + gather bodies spend on such a level what they spend on a fine one (scan,
// list, histogram, scatter, then rows of hundreds or thousands of records walked 8 at a time: ~30 us per workgroup at
// cfg-4 encoder, and half of role B's workgroups belong to the two coarse levels), all of it vector instructions, which is
// what these kernels are bound by (profiles/r04_notes.md §1).  But for so few rows the tap weights ARE a dense matrix:
//
//     grad_value[pixel r of the level, channel c] = sum over queries q of  Wt[r][q] * grad_out[q][c]
//     Wt[r][q] = sum of (bilinear x attention) weights of the taps of query q's P points that land on pixel r
//
// (a 4x4 level: every query has 4P taps on 16 pixels).  So a workgroup
//   1. builds Wt for a chunk of QC queries in LDS — one thread per query: the query's P points of this level are contiguous in
//      sampling_loc / attn_weight, and the thread is the only one that touches its column (plain read-add-write, no atomics,
//      no sort);
//   2. multiplies: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: the same arithmetic as an fmaf chain), A = 16 pixels x 4
//      queries from LDS, B = 4 queries x 16 channels straight from grad_out (one dword per lane, 64-byte row segments), the
//      k-steps of a chunk dealt to the 8 wavefronts;
//   3. after the last chunk adds the wavefronts' partial tiles in wavefront order through LDS and stores every row of the
//      level once.
// The matrix pipe is idle in every other part of these kernels, so the multiply costs the vector-bound workgroups next to it
// nothing.  The summation order is fixed (queries in order inside a wavefront, wavefronts in order): the result is the same on
// every run, so the deterministic flag takes this path too.
// A workgroup takes the level 32 pixels at a time (two 16-pixel tiles x two 16-channel tiles = four accumulators): one of the
// W_l workgroups a launch deals to the level, or two for 33..64 pixels; the others exit at once.  Each reads all the level's
// points (Lq*P*12 bytes) and builds its own 32 rows of Wt.
// Replaces, for such levels, the value half of ms_deform_im2col_cuda.cuh:301-403 / :537-641 (atomicAdd into grad_value).
#pragma once

namespace msda {

#ifndef MSDA_DENSE_MAX_W
#define MSDA_DENSE_MAX_W 2          // ranges per level up to which a coarse level goes dense (A/B: 8; dense_level below)
#endif
constexpr int kDenseMaxRows = 64;                  // pixels of a level served this way (four 16-row tiles)
constexpr int kDenseChunk = 512;                   // queries per chunk (QC): one per thread
constexpr int kDenseStride = kDenseChunk + 2;      // Wt row stride in words: the MFMA operand reads hit 32 different banks
constexpr int kDenseK = kDenseChunk / 4 / kSWaves; // k-steps (4 queries) of a chunk per wavefront
constexpr int kDensePassRows = 32;                 // pixels per pass: two 16-row tiles x two 16-channel tiles = 4 accumulators
static_assert(kDenseChunk == kSBlock, "one query of the chunk per thread");

// LDS the dense body needs (the reduction of step 3 reuses the same bytes: 8 wavefronts x 1 KB per output tile = 32 KB)
__host__ __device__ inline int dense_lds_bytes() { return (kDensePassRows + 1) * kDenseStride * 4; }     // (+ the spare row)
// Does level (H, Wd) of a call take the dense path in a workgroup with `lds_bytes` of LDS?  (uniform; the same answer in every
// workgroup of the launch)
__device__ __forceinline__ bool dense_level(long long H, long long Wd, long long start, int S, int Wl, int lds_bytes)
{
    // Wl <= 2: the level's work sits in one or two workgroups either way, and this body has a third of their instructions
    // (cfg-4 encoder backward 212 -> 196 us).  With more ranges per level (cfg-2 encoder: 6) one dense workgroup would take
    // over what six short ones share, and become the launch's longest: 57.4 -> 59.2 us.
    return Wl <= MSDA_DENSE_MAX_W && H * Wd <= (Wl >= 2 ? kDenseMaxRows : kDensePassRows) && H * Wd > 0 && level_fits(H, Wd, start, S) &&
           lds_bytes >= dense_lds_bytes();
}

template <typename VT> struct GoElem;              // one channel of a grad_out row through the pair's descriptor
template <> struct GoElem<float> {
    static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t rs, unsigned off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0)); }
};
template <> struct GoElem<bf16_t> {
    static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t rs, unsigned off) { return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0) << 16); }
};

using f32x4_t = __attribute__((ext_vector_type(4))) float;

// Loads the compiler should issue as global_load, not flat_load: inside a non-inlined function a pointer parameter has lost
// its address space.
using f32x2_t = __attribute__((ext_vector_type(2))) float;
#define MSDA_GLOBAL_PTR(T, p) reinterpret_cast<const __attribute__((address_space(1))) T *>(reinterpret_cast<unsigned long long>(p))
__device__ __forceinline__ float4 ld_global4(const float *p) { const f32x4_t v = *MSDA_GLOBAL_PTR(f32x4_t, p); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float2 ld_global2(const float *p) { const f32x2_t v = *MSDA_GLOBAL_PTR(f32x2_t, p); return make_float2(v[0], v[1]); }
__device__ __forceinline__ float ld_global1(const float *p) { return *MSDA_GLOBAL_PTR(float, p); }

// One pass: pixels [r0, r0 + 16 nmt) of the level (nmt = 1 or 2 tiles of 16), all 32 channels.
// A real function (its own register allocation: inlined into kernels whose sort bodies sit at the 128-register cap the
// compiler spilled this body's loop); its prologue saves the callee-saved registers once per workgroup.
template <typename VT, typename GT>
__device__ __attribute__((noinline)) void bwd_value_dense_pass(
    const VT *__restrict__ grad_out, const float *__restrict__ loc, const float *__restrict__ attn, int H, int Wd, int lstart,
    int S, int M, int L, int Lq, int P, GT *__restrict__ grad_value, int r0, int nmt, int l, int pr, unsigned char *smem)
{
    constexpr int QC = kDenseChunk, WS = kDenseStride, KW = kDenseK, KH = KW / 2;
    // (the arguments of a real function arrive in vector registers: say that they are uniform, or every test on them
    // becomes an execution-mask round trip)
    H = __builtin_amdgcn_readfirstlane(H); Wd = __builtin_amdgcn_readfirstlane(Wd); lstart = __builtin_amdgcn_readfirstlane(lstart);
    S = __builtin_amdgcn_readfirstlane(S); M = __builtin_amdgcn_readfirstlane(M); L = __builtin_amdgcn_readfirstlane(L);
    Lq = __builtin_amdgcn_readfirstlane(Lq); P = __builtin_amdgcn_readfirstlane(P); r0 = __builtin_amdgcn_readfirstlane(r0);
    nmt = __builtin_amdgcn_readfirstlane(nmt); l = __builtin_amdgcn_readfirstlane(l); pr = __builtin_amdgcn_readfirstlane(pr);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = H * Wd, rows = nmt * 16;
    const int b = pr / M, m = pr - b * M;
    float *Wt = reinterpret_cast<float *>(smem);
    const long long item_base = (long long)b * Lq * M + m;
    const GoBuf<VT> go = pair_rows<VT>(grad_out, item_base, Lq, M, 0);
    const int MLP = M * L * P;
    const unsigned pt0 = (unsigned)((item_base * L + l) * P);                        // point (b, q = 0, m, l, p = 0)
    const int ar = lane & 15, ak = lane >> 4;
    const unsigned boff = (unsigned)(ar * sizeof(VT));                               // this lane's channel inside a 16-channel tile

    f32x4_t acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // A chunk is three short phases between barriers, and on a busy chip a global load takes longer than any of them: no
    // load sits in front of its phase.  Half of the chunk's grad_out operands (independent of Wt) are requested before Wt is
    // built, the other half once the build's registers are free; the next chunk's sampling points before this chunk's products.
    // (First version, loads where they are used: 6.8 us per chunk of a 4x4 level, four exposed round trips in the product
    // loop alone.)
    float4 pxa = make_float4(0.f, 0.f, 0.f, 0.f), pxb = pxa, pa4 = pxa;              // P = 4: the thread's query of the chunk
    if (P == 4) {
        const unsigned pi = pt0 + (unsigned)(min(tid, Lq - 1) * MLP);
        pxa = ld_global4(loc + 2 * pi); pxb = ld_global4(loc + 2 * pi + 4);
        pa4 = ld_global4(attn + pi);
    }
    for (int c0 = 0; c0 < Lq; c0 += QC) {
        __syncthreads();                                     // the previous chunk's products have read Wt
        // grad_out operands of this wavefront's k-steps ks = wave, wave + 8, ...: lane -> query c0 + 4 ks + (lane >> 4), channel
        // lane & 15 of the tile; a query past Lq reads zeros (the descriptor ends with the pair's last row)
        float bv[KW][2];
        auto request_b = [&](int i) {
            const unsigned qoff = (unsigned)__mul24(min(c0 + (wave + i * kSWaves) * 4 + ak, Lq), go.stride_b) + boff;
            bv[i][0] = GoElem<VT>::load(go.rs, qoff);
            bv[i][1] = GoElem<VT>::load(go.rs, qoff + (unsigned)(16 * sizeof(VT)));
        };
#pragma unroll
        for (int i = 0; i < KH; ++i) request_b(i);
        // ---- 1a. Wt = 0 ----
        {
            float4 *w4 = reinterpret_cast<float4 *>(Wt);
            const int n4 = (rows * WS) >> 2;                 // (rows a multiple of 16: whole float4s)
            for (int i = tid; i < n4; i += kSBlock) w4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if (c0 == 0 && r0 == 0) MSDA_STAMP(1);
        // ---- 1b. one thread per query of the chunk: its P points' taps into its own column ----
        if (c0 + tid < Lq) {
            // the valid taps of ONE point land on different pixels: their four read-add-writes go side by side (the compiler
            // must assume they alias and would chain sixteen LDS round trips per query).  Word index of pixel r in this
            // thread's column: (r - r0) * WS + tid (24-bit multiply).  A tap that is absent or outside the pass goes to the
            // thread's word of a spare row nobody reads: no branch, no execution mask (the first version's `&&` and guarded
            // stores compiled to five mask round trips per tap).
            auto add_point = [&](float x, float y, float a) {
                const PointGeom<float> g = point_geom<float>(x, y, H, Wd);
                const int pix = __mul24(g.h0, Wd) + g.w0 - r0;
                const float hh = 1.f - g.lh, hw = 1.f - g.lw;
                const float tw[4] = {hh * hw * a, hh * g.lw * a, g.lh * hw * a, g.lh * g.lw * a};
                const int px[4] = {pix, pix + 1, pix + Wd, pix + Wd + 1};
                const bool ok[4] = {g.ok00, g.ok01, g.ok10, g.ok11};
                int at[4]; float old[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool use = ok[k] & ((unsigned)px[k] < (unsigned)rows);
                    at[k] = __mul24(use ? px[k] : kDensePassRows, WS) + tid;
                    old[k] = Wt[at[k]];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) Wt[at[k]] = old[k] + tw[k];
            };
            if (P == 4) {                                    // (uniform) the usual case: 32 + 16 contiguous, aligned bytes, requested a chunk ahead
                add_point(pxa.x, pxa.y, pa4.x); add_point(pxa.z, pxa.w, pa4.y); add_point(pxb.x, pxb.y, pa4.z); add_point(pxb.z, pxb.w, pa4.w);
            } else {
                const unsigned pi = pt0 + (unsigned)((c0 + tid) * MLP);
                for (int p = 0; p < P; ++p) { const float2 xy = ld_global2(loc + 2 * (pi + p)); add_point(xy.x, xy.y, ld_global1(attn + pi + p)); }
            }
        }
#pragma unroll
        for (int i = KH; i < KW; ++i) request_b(i);
        if (P == 4 && c0 + QC < Lq) {
            const unsigned pi = pt0 + (unsigned)(min(c0 + QC + tid, Lq - 1) * MLP);
            pxa = ld_global4(loc + 2 * pi); pxb = ld_global4(loc + 2 * pi + 4);
            pa4 = ld_global4(attn + pi);
        }
        __syncthreads();
        if (c0 == 0 && r0 == 0) MSDA_STAMP(2);
        // ---- 2. products.  A k-step past the chunk's queries multiplies zeros by zeros (Wt's columns there were never
        // built, grad_out's rows there are past the descriptor): no guard needed, only whole groups are skipped ----
        const int ksteps = (min(QC, Lq - c0) + 3) >> 2;
        constexpr int G = 4;                                 // k-steps whose Wt operands are read together
#pragma unroll
        for (int i0 = 0; i0 < KW; i0 += G) {
            if (wave + i0 * kSWaves >= ksteps) break;        // (scalar)
            float av[G][2];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float *wp = Wt + ar * WS + (wave + (i0 + g) * kSWaves) * 4 + ak;
                av[g][0] = wp[0];
                av[g][1] = nmt > 1 ? wp[16 * WS] : 0.f;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][0], bv[i0 + g][nt], acc[0][nt], 0, 0, 0);
                if (nmt > 1) {                               // (scalar)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][1], bv[i0 + g][nt], acc[1][nt], 0, 0, 0);
                }
            }
        }
        if (c0 == 0 && r0 == 0) MSDA_STAMP(3);
    }
    // ---- 3. the wavefronts' partial tiles, added in wavefront order; every row of the pass stored once ----
    __syncthreads();
    if (r0 == 0) MSDA_STAMP(4);
    float *red = Wt;                                         // [tile = 2 mt + nt][wavefront][16 rows][16 channels]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
        if (mt < nmt) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float *dst = red + (((mt * 2 + nt) * kSWaves + wave) << 8) + (lane >> 4) * 64 + (lane & 15);      // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
                dst[0] = acc[mt][nt][0]; dst[16] = acc[mt][nt][1]; dst[32] = acc[mt][nt][2]; dst[48] = acc[mt][nt][3];
            }
        }
    __syncthreads();
    if (wave < nmt * 2) {                                    // 64 threads per tile
        const int mt = wave >> 1, nt = wave & 1;
        const int row = lane >> 2, c4 = lane & 3;
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < kSWaves; ++w) add4(s4, *reinterpret_cast<const float4 *>(red + ((wave * kSWaves + w) << 8) + row * 16 + c4 * 4));
        const int r = r0 + mt * 16 + row;
        if (r < HW) Row<GT>::store(grad_value + ((long long)(b * S + lstart + r) * M + m) * kD + nt * 16 + c4 * 4, s4);
    }
}

// ti: this workgroup's index among the Wl workgroups the launch deals to (pair pr, level l).  A pass covers 32 pixels: a
// level of up to 32 pixels is one pass (workgroup 0; the others exit at once), a level of 33..64 is two, one each for
// workgroups 0 and 1 (dense_level() sends such a level of a one-workgroup-per-level launch to the sort + gather body: two
// passes in turn measured slower than that — cfg-4 decoder's 7x7 level, 21.8 against 19.0 us per workgroup).
template <typename VT, typename GT>
__device__ __forceinline__ void bwd_value_dense_body(
    const VT *__restrict__ grad_out, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L, int Lq, int P,
    GT *__restrict__ grad_value, int ti, int Wl, int l, int pr, unsigned char *smem)
{
    const int H = (int)shapes[2 * l], Wd = (int)shapes[2 * l + 1], lstart = (int)level_start[l];
    const int MT = (H * Wd + 15) >> 4;
    MSDA_STAMP(0);
    if (2 * ti < MT)
        bwd_value_dense_pass<VT, GT>(grad_out, loc, attn, H, Wd, lstart, S, M, L, Lq, P, grad_value, ti * kDensePassRows,
                                     min(2, MT - 2 * ti), l, pr, smem);
    MSDA_STAMP(5);
}

}  // namespace msda
