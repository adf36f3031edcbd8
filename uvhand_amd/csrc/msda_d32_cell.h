// Role B of the D = 32 backward, second generation: grad_value from POINT records sorted by bilinear cell.
// Included by msda_d32.hip (after its row / DPP helpers); replaces the reference's atomicAdd scatter
// (UVHand models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159, lines 125-152) and the first-generation
// per-tap sort of this repository.
//
// A sampling point lands in one bilinear CELL (its top-left tap (h0, w0), h0 in [-1, H-1], w0 in [-1, W-1])
// and feeds the four pixels at the cell's corners.  The per-tap formulation loads the point's grad_out row
// four times — once per corner, from four different destination rows — and sorts four records per point.
// Here a record is a POINT:
//
//   * a workgroup owns a TILE of TH x TW destination pixels of one (batch, head, level) (TH*TW <= 256, TW <= 16,
//     chosen on the device from the level's shape) and, for large problems, one query CHUNK of it;
//   * it scans its chunk's sampling points (loc / attn straight from HBM, one lane per point, prefetched one
//     round ahead), keeps those whose cell touches the tile, and appends {query, lh, lw, attn} to a pending list
//     in LDS; the list is sorted by cell with a STABLE counting sort (per-wavefront histograms: the order inside
//     a cell is wave-major, then the order in which the wavefront met the points — the same on every run);
//   * gather: 8 lanes x float4 per cell walk the cell's records — ONE coalesced 128-B read of grad_out per
//     point — and keep the four corner sums in registers; the sums are then added into the tile's fp32 image
//     in LDS with plain read-add-write.  Cells are processed in four COLOURS ((row parity, column parity));
//     two cells of one colour never share a corner, so no two lane groups touch the same pixel between two
//     barriers: no atomics of any kind on floating-point data, and a fixed summation order;
//   * the tile image accumulates over as many sort+gather batches as the chunk needs and is written once:
//     to grad_value, or — when the level is cut into C > 1 query chunks — to one of C partial slabs that
//     msda::slab_reduce_kernel adds up in chunk order (bitwise reproducible).
//
// Every pixel of every level is written exactly once (zeros included); nothing is zero-filled beforehand.
#pragma once

namespace msda {

#ifndef MSDA_CELL_THREADS
#define MSDA_CELL_THREADS 512
#endif
constexpr int kCBlock = MSDA_CELL_THREADS;   // threads per role-B workgroup (and per role-A workgroup of the fused launch)
constexpr int kCWaves = kCBlock / kWave;
#ifndef MSDA_CELL_TILE_ROWS
#define MSDA_CELL_TILE_ROWS (MSDA_CELL_THREADS / 2)
#endif
#ifndef MSDA_CELL_PEND
#define MSDA_CELL_PEND (2 * MSDA_CELL_THREADS)
#endif
#ifndef MSDA_CELL_MIN_WAVES
#define MSDA_CELL_MIN_WAVES ((128 * 8) / MSDA_CELL_THREADS >= 4 ? 4 : 2)
#endif
constexpr int kTileRows = MSDA_CELL_TILE_ROWS;   // destination pixels per tile: 128 B of fp32 image each
constexpr int kTileW = 16;                   // widest tile
// cells of a tile are numbered colour-major: 4 * ((TH/2)+1) * ((TW/2)+1) keys; the maximum over TH*TW <= kTileRows,
// TW <= 16 is at TW = 2 (130 for 256 rows, 66 for 128)
constexpr int kMaxKeys = kTileRows == 256 ? 576 : kTileRows == 128 ? 320 : 4 * (kTileRows / 2 + 2);
constexpr int kPendCap = MSDA_CELL_PEND;     // pending points per sort+gather batch
constexpr int kScanPPT = 1;                  // points per thread per scan round
constexpr int kSortedCap = kPendCap + 3 * kMaxKeys;             // sorted positions: every cell starts at a multiple of 4

// One record per kept sampling point, written once by the scan (one lane per point): the four corner weights
// (bilinear x attention) and the element offset of the query's grad_out row relative to the (batch, head) base.
struct alignas(16) WRec { float k00, k01, k10, k11; };

// Diagnostic build only (-DMSDA_STAMPS, tools/micro/kbench.cpp): thread 0 sums the time a workgroup spends in each
// phase of its items; [blocks][8] = {start, end, scan, sort, gather, flush, batches, kept points} in region 0.
#ifdef MSDA_STAMPS
#define CELL_T() __builtin_amdgcn_s_memrealtime()
#define CELL_ACC(var, t0) do { const unsigned long long t1_ = CELL_T(); (var) += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define CELL_T() 0ull
#define CELL_ACC(var, t0) do { } while (0)
#endif

// Host -> device plan of role B (all uniform).
struct CellPlan {
    int slots;                 // G: workgroup slots per (batch, head) pair; a slot loops over items slot, slot+G, ...
    int c_max;                 // most query chunks per (level, tile); 1 = no slabs
    int k_chunk;               // sampling points a (tile, chunk) should keep, for the choice of C
    int slab_rows;             // slab rows (of 32 floats) available per pair
    float *slabs;              // [pairs][slab_rows][32] or null
};

// LDS of a role-B workgroup: tile image | weights | row offsets | (cell, wave, rank) of each pending point | sorted
// position -> pending slot | per-(cell, wavefront) counters, two 16-bit counters per word | non-empty cells | scalars
constexpr size_t kCellLdsBytes = (size_t)kTileRows * kD * 4 + (size_t)kPendCap * (sizeof(WRec) + 4 + 4) +
                                 (size_t)kSortedCap * 2 + (size_t)kMaxKeys * (kCWaves / 2) * 4 + (size_t)kMaxKeys * 8 + 128;

// Tiling of one level, identical in every kernel that needs it (role B, slab reduce).
struct LevelTiles {
    int H, Wd, lstart, HW;
    int TH, TW, nH, nW, ntile;
    int C;                     // query chunks per tile
    int slab0;                 // first slab row of the level (C > 1 only)
};

__device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// Levels whose pixels do not lie inside [0, S) are skipped altogether (ntile = 0): nothing is ever read or
// written outside the tensors, whatever spatial_shapes / level_start_index say (include/msda.h).
__device__ __forceinline__ void level_tiles(const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
                                            int l, int S, int NP, int Lq, const CellPlan &pl, int &slab_cursor,
                                            LevelTiles &t)
{
    const long long H = shapes[2 * l], W = shapes[2 * l + 1], st = level_start[l];
    t.H = (int)H; t.Wd = (int)W; t.lstart = (int)st; t.HW = 0;
    t.TH = t.TW = 1; t.nH = t.nW = t.ntile = 0; t.C = 1; t.slab0 = 0;
    if (H <= 0 || W <= 0 || st < 0 || H * W > (long long)S || st + H * W > (long long)S) return;
    t.HW = (int)(H * W);
    t.nW = cdiv(t.Wd, kTileW);
    t.TW = cdiv(t.Wd, t.nW);
    const int thmax = max(1, kTileRows / t.TW);
    t.nH = cdiv(t.H, thmax);
    t.TH = cdiv(t.H, t.nH);
    t.ntile = t.nH * t.nW;
    if (pl.c_max > 1 && pl.slabs != nullptr) {
        int c = cdiv(NP, max(1, t.ntile * pl.k_chunk));
        c = max(1, min(min(c, pl.c_max), Lq));
        if (c > 1 && (long long)slab_cursor + (long long)c * t.HW <= (long long)pl.slab_rows) {
            t.C = c;
            t.slab0 = slab_cursor;
            slab_cursor += c * t.HW;
        }
    }
}

// ---- gather of one colour: compact entries [e0, e1) of `nz` ---------------------------------------------
// nz[e] = { first sorted position (a multiple of 4), count << 16 | tile-local cell row << 5 | cell column }.
// SLOTS lane groups (of one wavefront) share a cell: group s takes the records [s*per, (s+1)*per), per a multiple of 4,
// so every group reads its sorted indices as aligned quads.  Full quads run without any select; the last, partial quad
// of a group zeroes the weights of the missing records and points their row at the quad's first (valid) record.
__device__ __forceinline__ void fma_corners(float4 &a00, float4 &a01, float4 &a10, float4 &a11, const WRec &w, const float4 &g)
{
    fma4(a00, w.k00, g); fma4(a01, w.k01, g); fma4(a10, w.k10, g); fma4(a11, w.k11, g);
}

template <int SLOTS, typename VT>
__device__ __forceinline__ void gather_cells(const VT *__restrict__ go_base, const int2 *nz, int e0, int e1,
                                             const unsigned short *sidx, const WRec *wrec, const int *qrec, float *tile,
                                             int th, int tw)
{
    constexpr int CPW = 8 / SLOTS;                           // cells per wavefront trip
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = lane / (SLOTS * 8), slot = (lane >> 3) % SLOTS, j = lane & 7;
    for (int eb = e0 + wave * CPW; eb < e1; eb += kCWaves * CPW) {          // wave-uniform trip count
        const int e = eb + sub;
        const bool have = e < e1;
        const int2 ent = have ? nz[e] : make_int2(0, 0);
        const int cnt = have ? (int)((unsigned)ent.y >> 16) : 0;
        int lo = 0, hi = cnt;
        if (SLOTS > 1) {
            const int per = (((cnt + SLOTS - 1) / SLOTS) + 3) & ~3;
            lo = min(cnt, slot * per); hi = min(cnt, lo + per);
        }
        const unsigned short *ix = sidx + ent.x;
        float4 a00 = make_float4(0.f, 0.f, 0.f, 0.f), a01 = a00, a10 = a00, a11 = a00;
        int i = lo;
        for (; i + 4 <= hi; i += 4) {                                          // full quads: 4 row loads in flight
            const uint2 q4 = *reinterpret_cast<const uint2 *>(ix + i);
            const int i0 = q4.x & 0xffff, i1 = q4.x >> 16, i2 = q4.y & 0xffff, i3 = q4.y >> 16;
            const WRec w0 = wrec[i0], w1 = wrec[i1], w2 = wrec[i2], w3 = wrec[i3];
            const float4 g0 = Row<VT>::load(go_base + qrec[i0]), g1 = Row<VT>::load(go_base + qrec[i1]);
            const float4 g2 = Row<VT>::load(go_base + qrec[i2]), g3 = Row<VT>::load(go_base + qrec[i3]);
            fma_corners(a00, a01, a10, a11, w0, g0); fma_corners(a00, a01, a10, a11, w1, g1);
            fma_corners(a00, a01, a10, a11, w2, g2); fma_corners(a00, a01, a10, a11, w3, g3);
        }
        if (i < hi) {                                                          // last quad, 1..3 records
            const uint2 q4 = *reinterpret_cast<const uint2 *>(ix + i);
            const int i0 = q4.x & 0xffff;
            const int n = hi - i;
            const int i1 = n > 1 ? (int)(q4.x >> 16) : i0, i2 = n > 2 ? (int)(q4.y & 0xffff) : i0;
            WRec w0 = wrec[i0], w1 = wrec[i1], w2 = wrec[i2];
            if (n <= 1) w1.k00 = w1.k01 = w1.k10 = w1.k11 = 0.f;
            if (n <= 2) w2.k00 = w2.k01 = w2.k10 = w2.k11 = 0.f;
            const float4 g0 = Row<VT>::load(go_base + qrec[i0]), g1 = Row<VT>::load(go_base + qrec[i1]);
            const float4 g2 = Row<VT>::load(go_base + qrec[i2]);
            fma_corners(a00, a01, a10, a11, w0, g0); fma_corners(a00, a01, a10, a11, w1, g1);
            fma_corners(a00, a01, a10, a11, w2, g2);
        }
        if (SLOTS >= 2) { add4(a00, shfl_xor4(a00, 8)); add4(a01, shfl_xor4(a01, 8)); add4(a10, shfl_xor4(a10, 8)); add4(a11, shfl_xor4(a11, 8)); }
        if (SLOTS >= 4) { add4(a00, shfl_xor4(a00, 16)); add4(a01, shfl_xor4(a01, 16)); add4(a10, shfl_xor4(a10, 16)); add4(a11, shfl_xor4(a11, 16)); }
        if (SLOTS >= 8) { add4(a00, shfl_xor4(a00, 32)); add4(a01, shfl_xor4(a01, 32)); add4(a10, shfl_xor4(a10, 32)); add4(a11, shfl_xor4(a11, 32)); }
        if (have && slot == 0) {
            // corners of cell (tch, tcw): pixels (tch-1, tcw-1) (tch-1, tcw) (tch, tcw-1) (tch, tcw), tile-local
            const int tch = (ent.y >> 5) & 2047, tcw = ent.y & 31;
            float4 *t4 = reinterpret_cast<float4 *>(tile) + ((tch - 1) * tw + tcw - 1) * 8 + j;
            const bool r0 = tch >= 1, r1 = tch < th, c0 = tcw >= 1, c1 = tcw < tw;
            if (r0 && c0) { float4 o = t4[0]; add4(o, a00); t4[0] = o; }
            if (r0 && c1) { float4 o = t4[8]; add4(o, a01); t4[8] = o; }
            if (r1 && c0) { float4 o = t4[tw * 8]; add4(o, a10); t4[tw * 8] = o; }
            if (r1 && c1) { float4 o = t4[tw * 8 + 8]; add4(o, a11); t4[tw * 8 + 8] = o; }
        }
    }
}

// ---- one work item: (pair, level, tile, chunk) ---------------------------------------------------------------
// VT = storage of grad_out, GT = storage of grad_value.
template <typename VT, typename GT>
__device__ __forceinline__ void cell_item(
    const VT *__restrict__ grad_out, const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L,
    int Lq, int P, int p_shift, GT *__restrict__ grad_value, const LevelTiles &lt, int l, int tile_i, int chunk, int pr,
    float *slab_pair, unsigned char *smem)
{
    float *tile = reinterpret_cast<float *>(smem);                               // [kTileRows][32]
    WRec *wrec = reinterpret_cast<WRec *>(tile + kTileRows * kD);               // [kPendCap] corner weights
    int *qrec = reinterpret_cast<int *>(wrec + kPendCap);                       // [kPendCap] grad_out row offset (elements)
    unsigned *pk = reinterpret_cast<unsigned *>(qrec + kPendCap);               // [kPendCap] (key*8+wave) << 16 | rank
    unsigned *cnt = pk + kPendCap;                                              // [kMaxKeys][kCWaves/2]: two 16-bit counters per word
    int2 *nz = reinterpret_cast<int2 *>(cnt + kMaxKeys * (kCWaves / 2));        // [kMaxKeys] non-empty cells
    unsigned short *sidx = reinterpret_cast<unsigned short *>(nz + kMaxKeys);   // [kSortedCap] sorted position -> pending slot
    int *wcnt = reinterpret_cast<int *>(sidx + kSortedCap);                     // [2][8] kept points per wavefront
    int *cstart = wcnt + 16;                                                    // [5] first compact entry of each colour
    int *wsum = cstart + 8;                                                     // [8] prefix-scan partials
    constexpr int CW = kCWaves / 2;                                             // counter words per cell

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = pr / M, m = pr - b * M;
    const int ti_h = tile_i / lt.nW, ti_w = tile_i - ti_h * lt.nW;
    const int h_lo = ti_h * lt.TH, w_lo = ti_w * lt.TW;
    const int th = min(lt.TH, lt.H - h_lo), tw = min(lt.TW, lt.Wd - w_lo);      // >= 1 by construction
    const int npx = th * tw;
    const int kcw = (tw >> 1) + 1, kc = ((th >> 1) + 1) * kcw, nkeys = 4 * kc;    // key = colour*kc + (tch/2)*kcw + tcw/2
    const int q0 = (int)((long long)chunk * Lq / lt.C), q1 = (int)((long long)(chunk + 1) * Lq / lt.C);
    const int pt0 = q0 * P, pt1 = q1 * P;
    const long long item_base = (long long)b * Lq * M + m;                       // item(q) = item_base + q*M
    const int row_stride = M * kD;                                               // elements between consecutive queries' rows
    const VT *go_base = grad_out + item_base * kD + (lane & 7) * 4;

    for (int i = tid; i < npx * 8; i += kCBlock) reinterpret_cast<float4 *>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < nkeys * CW; i += kCBlock) cnt[i] = 0;
    // (the first round's barrier orders these stores before any use)

    unsigned long long st_t = CELL_T(), st_begin = st_t, st_scan = 0, st_sort = 0, st_gather = 0, st_flush = 0, st_batches = 0, st_kept = 0;
    (void)st_begin; (void)st_scan; (void)st_sort; (void)st_gather; (void)st_flush; (void)st_batches; (void)st_kept;
    int pcount = 0;                                                              // pending points (uniform)
    const int rounds = (pt1 - pt0 + kCBlock * kScanPPT - 1) / (kCBlock * kScanPPT);
    float2 xy_n = make_float2(-8.f, -8.f); float at_n = 0.f; int q_n = -1;
    {
        const int idx = pt0 + tid;
        if (idx < pt1) {
            const int q = fdiv(idx, P, p_shift), p = idx - q * P;
            const long long pi = ((item_base + (long long)q * M) * L + l) * P + p;
            xy_n = reinterpret_cast<const float2 *>(loc)[pi]; at_n = attn[pi]; q_n = q;
        }
    }
    for (int r = 0; r < rounds; ++r) {
        int key = -1; WRec rec; rec.k00 = rec.k01 = rec.k10 = rec.k11 = 0.f; int qoff = 0;
        {
            const float2 xy = xy_n; const float at = at_n; const int q = q_n;
            // prefetch the next round's point
            xy_n = make_float2(-8.f, -8.f); at_n = 0.f; q_n = -1;
            const int idx = pt0 + (r + 1) * kCBlock + tid;
            if (idx < pt1) {
                const int qn = fdiv(idx, P, p_shift), p = idx - qn * P;
                const long long pi = ((item_base + (long long)qn * M) * L + l) * P + p;
                xy_n = reinterpret_cast<const float2 *>(loc)[pi]; at_n = attn[pi]; q_n = qn;
            }
            const PointGeom<float> g = point_geom<float>(xy.x, xy.y, lt.H, lt.Wd);
            const int tch = g.h0 + 1 - h_lo, tcw = g.w0 + 1 - w_lo;              // cell, tile-local
            if (q >= 0 && g.inside && tch >= 0 && tch <= th && tcw >= 0 && tcw <= tw) {
                key = ((tch & 1) * 2 + (tcw & 1)) * kc + (tch >> 1) * kcw + (tcw >> 1);
                const float hh = 1.f - g.lh, hw = 1.f - g.lw;
                rec.k00 = hh * hw * at; rec.k01 = hh * g.lw * at; rec.k10 = g.lh * hw * at; rec.k11 = g.lh * g.lw * at;
                qoff = q * row_stride;                                          // < 2^31: d32_supported()
            }
        }
        const unsigned long long bal = __ballot(key >= 0);
        if (lane == 0) wcnt[(r & 1) * 8 + wave] = __popcll(bal);
        __syncthreads();
        {
            int base = pcount, total = 0;
#pragma unroll
            for (int w2 = 0; w2 < kCWaves; ++w2) { const int c = wcnt[(r & 1) * 8 + w2]; if (w2 < wave) base += c; total += c; }
            if (key >= 0) {
                const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
                // this wavefront's own 16-bit counter of the cell: the order inside a cell is the same on every run
                const unsigned old = atomicAdd(&cnt[key * CW + (wave >> 1)], 1u << ((wave & 1) * 16));
                const unsigned rank = (old >> ((wave & 1) * 16)) & 0xffffu;
                wrec[pos] = rec; qrec[pos] = qoff;
                pk[pos] = ((unsigned)(key * kCWaves + wave) << 16) | rank;
            }
            pcount += total;
        }
        // sort + gather when the next round might not fit, or at the end of the chunk
        if (!(pcount > kPendCap - kCBlock * kScanPPT || r + 1 >= rounds)) continue;
        __syncthreads();
        CELL_ACC(st_scan, st_t); st_batches += 1; st_kept += pcount;
        // ---- prefix over cells: thread t owns cells [t*KPT, (t+1)*KPT) with their per-wavefront counters; every cell's
        // run starts at a multiple of 4 (aligned quads of sorted indices) ----
        const int KPT = (nkeys + kCBlock - 1) / kCBlock;                          // 1, 2 or 3
        int tot[3] = {0, 0, 0}, mine = 0, mine_nz = 0;
        for (int k = 0; k < KPT; ++k) {
            const int ky = tid * KPT + k;
            if (ky < nkeys) {
                int t = 0;
#pragma unroll
                for (int w2 = 0; w2 < CW; ++w2) { const unsigned c = cnt[ky * CW + w2]; t += (int)(c & 0xffffu) + (int)(c >> 16); }
                tot[k] = t;
                mine += (t + 3) & ~3; mine_nz += t > 0;
            }
        }
        int incl = mine | (mine_nz << 20);                                        // padded records and non-empty cells in one scan
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) { const int y = __shfl_up(incl, o, kWave); if (lane >= o) incl += y; }
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        int excl = incl - (mine | (mine_nz << 20)), all = 0;
        {
#pragma unroll
            for (int w2 = 0; w2 < kCWaves; ++w2) { const int c = wsum[w2]; if (w2 < wave) excl += c; all += c; }
        }
        int start = excl & 0xfffff, nzpos = excl >> 20;
        for (int k = 0; k < KPT; ++k) {
            const int ky = tid * KPT + k;
            if (ky < nkeys) {
                const int colour = ky / kc;
                if (ky == colour * kc) cstart[colour] = nzpos;
                if (tot[k] > 0) {
                    const int rem = ky - colour * kc, rh = rem / kcw, rw = rem - rh * kcw;
                    const int tch = rh * 2 + (colour >> 1), tcw = rw * 2 + (colour & 1);
                    nz[nzpos++] = make_int2(start, (tot[k] << 16) | (tch << 5) | tcw);
                    unsigned s = (unsigned)start;                                  // counts -> first positions, in place
#pragma unroll
                    for (int w2 = 0; w2 < CW; ++w2) {
                        const unsigned c = cnt[ky * CW + w2];
                        const unsigned lo16 = s; s += c & 0xffffu;
                        const unsigned hi16 = s; s += c >> 16;
                        cnt[ky * CW + w2] = lo16 | (hi16 << 16);
                    }
                    start += (tot[k] + 3) & ~3;
                }
            }
        }
        if (tid == 0) cstart[4] = all >> 20;
        __syncthreads();
        // ---- sorted position -> pending slot ----
        for (int i = tid; i < pcount; i += kCBlock) {
            const unsigned v = pk[i], f = v >> 16;                                 // f = key*kCWaves + wave
            const unsigned word = cnt[(f / kCWaves) * CW + ((f % kCWaves) >> 1)];
            sidx[((word >> ((f & 1) * 16)) & 0xffffu) + (v & 0xffffu)] = (unsigned short)i;
        }
        __syncthreads();
        CELL_ACC(st_sort, st_t);
        // ---- gather, colour by colour; lanes per cell from the mean run length and the number of cells ----
        {
            const int nnz = all >> 20;
            const int mean2 = nnz > 0 ? (2 * pcount) / nnz : 0;                   // 2 x mean points per non-empty cell
            const int per_colour = max(1, (nnz + 3) >> 2);
            int slots = mean2 <= 32 ? 1 : mean2 <= 64 ? 2 : mean2 <= 128 ? 4 : 8;
            // few cells: spread each over more lane groups as long as every group still has a few quads
            while (slots < 8 && per_colour * slots < kCBlock / 8 && mean2 >= 16 * slots) slots <<= 1;
            for (int c = 0; c < 4; ++c) {
                const int e0 = cstart[c], e1 = cstart[c + 1];
                if (slots == 1)      gather_cells<1, VT>(go_base, nz, e0, e1, sidx, wrec, qrec, tile, th, tw);
                else if (slots == 2) gather_cells<2, VT>(go_base, nz, e0, e1, sidx, wrec, qrec, tile, th, tw);
                else if (slots == 4) gather_cells<4, VT>(go_base, nz, e0, e1, sidx, wrec, qrec, tile, th, tw);
                else                 gather_cells<8, VT>(go_base, nz, e0, e1, sidx, wrec, qrec, tile, th, tw);
                __syncthreads();
            }
        }
        CELL_ACC(st_gather, st_t);
        for (int i = tid; i < nkeys * CW; i += kCBlock) cnt[i] = 0;               // next batch (ordered by its round barrier)
        pcount = 0;
    }
    __syncthreads();
    // ---- write the tile: every pixel once ----
    for (int rr = tid >> 3; rr < npx; rr += kCBlock / 8) {
        const int dh = rr / tw, dw = rr - dh * tw;
        const int pix = (h_lo + dh) * lt.Wd + (w_lo + dw);
        const float4 v = reinterpret_cast<const float4 *>(tile)[rr * 8 + (tid & 7)];
        if (lt.C > 1) {
            *reinterpret_cast<float4 *>(slab_pair + ((long long)lt.slab0 + (long long)chunk * lt.HW + pix) * kD + (tid & 7) * 4) = v;
        } else {
            Row<GT>::store(grad_value + ((long long)(b * S + lt.lstart + pix) * M + m) * kD + (tid & 7) * 4, v);
        }
    }
    __syncthreads();                                                               // the LDS arrays are reused by the next item
#ifdef MSDA_STAMPS
    CELL_ACC(st_flush, st_t);
    if (tid == 0 && msda_stamp_buf) {
        unsigned long long *o = msda_stamp_buf + (size_t)(blockIdx.x & 65535) * 8;
        o[0] = st_begin; o[1] = st_t; o[2] += st_scan; o[3] += st_sort; o[4] += st_gather; o[5] += st_flush; o[6] += st_batches; o[7] += st_kept;
    }
#endif
}

// Role-B workgroup `slot` of pair `pr`: loops over the pair's items slot, slot + G, ...
template <typename VT, typename GT>
__device__ __forceinline__ void bwd_cell_body(
    const VT *__restrict__ grad_out, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L, int Lq, int P, int p_shift,
    GT *__restrict__ grad_value, const CellPlan pl, int pr, int slot, unsigned char *smem)
{
    const int NP = Lq * P;
    float *slab_pair = pl.slabs ? pl.slabs + (long long)pr * pl.slab_rows * kD : nullptr;
    int cursor = 0, first = 0;
    // Items are numbered level by level, tiles then chunks (numbering the coarse levels — the longest items of a pair —
    // first was measured: cfg-4 encoder role B 264 -> 410 us; profiles/r02_notes.md).
    for (int l = 0; l < L; ++l) {                                                  // uniform (scalar) walk over the levels
        LevelTiles lt;
        level_tiles(shapes, level_start, l, S, NP, Lq, pl, cursor, lt);
        const int items = lt.ntile * lt.C;
        // items of this level owned by this slot: global item numbers first .. first+items-1, mine = slot mod G
        int it = slot - first % pl.slots;
        if (it < 0) it += pl.slots;
        for (; it < items; it += pl.slots)
            cell_item<VT, GT>(grad_out, loc, attn, S, M, L, Lq, P, p_shift, grad_value, lt, l, it / lt.C, it % lt.C, pr,
                              slab_pair, smem);
        first += items;
    }
    // rows no level covers (inconsistent shapes only): defined, zero
    if (slot == 0) zero_uncovered_rows<GT, kCBlock>(shapes, level_start, S, M, L, grad_value, pr / M, pr % M);
}

// Adds up the C partial slabs of every pixel of the levels that were cut into query chunks, in chunk order.
// grid = pairs x ceil(S / 32) workgroups of 256 threads (8 lanes x float4 per pixel).
template <typename GT>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
                                                          int S, int M, int L, int Lq, int P, GT *__restrict__ grad_value,
                                                          const CellPlan pl, int row_blocks)
{
    const int pr = (int)blockIdx.x / row_blocks, rb = (int)blockIdx.x - pr * row_blocks;
    const int s = rb * 32 + ((int)threadIdx.x >> 3), j = (int)threadIdx.x & 7;
    if (s >= S) return;
    const int b = pr / M, m = pr - b * M, NP = Lq * P;
    const float *slab_pair = pl.slabs + (long long)pr * pl.slab_rows * kD;
    int cursor = 0;
    for (int l = 0; l < L; ++l) {
        LevelTiles lt;
        level_tiles(shapes, level_start, l, S, NP, Lq, pl, cursor, lt);
        if (lt.ntile == 0 || lt.C <= 1 || s < lt.lstart || s >= lt.lstart + lt.HW) continue;
        const int pix = s - lt.lstart;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c < lt.C; ++c)
            add4(acc, *reinterpret_cast<const float4 *>(slab_pair + ((long long)lt.slab0 + (long long)c * lt.HW + pix) * kD + j * 4));
        Row<GT>::store(grad_value + ((long long)(b * S + s) * M + m) * kD + j * 4, acc);
    }
}

}  // namespace msda
