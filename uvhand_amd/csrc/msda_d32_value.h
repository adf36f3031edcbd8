// Role B of the D = 32 backward, first generation (the default path): grad_value from per-TAP records sorted by
// destination row — device code only.  Included by msda_d32.hip after its row / DPP helpers (Row<>, fma4, fdiv,
// xcd_block, MSDA_STAMP); the kernels that call it (bwd_value_d32_kernel, bwd_fused_d32_kernel) and the host plans
// (plan_value, plan_fused) stay there.  Replaces the reference's atomicAdd scatter
// (UVHand models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159, lines 125-152).
//
//   gather_rows       rows in lockstep, SLOTS lane groups per row                (fixed-capacity sort, LDS-tile passes)
//   gather_balanced   sorted records dealt to the 64 lane groups at row boundaries (rows of comparable length)
//   gather_split      64 equal stretches of records, parts combined by the row's owner (a few very long rows)
//   bwd_value_body    scan + histogram + prefix sum + counting sort + gather, one pass or query chunks
//   bwd_value_wide_body   kAccWide: ONE pass sized by the taps a workgroup keeps
//   value_block_to_range  workgroup id -> (pair, level, pixel range)
#pragma once

#ifndef MSDA_GATHER_NR1
#define MSDA_GATHER_NR1 4        // gather_rows, fixed-capacity path: rows in flight when one lane group serves a row (A/B: 2)
#endif

namespace msda {

// ------------------------------------------------------------------------------------------
// backward, role B — destination-major: grad_value with no float atomics at all.
//
// grad_value[b, s, m, :] = sum over the taps that land on pixel s of (bilinear weight * attention
// weight) * grad_out[b, q, m, :] — a sparse-matrix x dense-matrix product whose sparse factor is
// only known at run time.  Float atomics are the wrong tool for it on this chip: global float
// atomics run at ~1.3 TB/s of added bytes chip-wide (MI355X_MICROARCH.md), and LDS float atomics
// are slower still — ds_add_f32 measured ~200 cycles per wavefront instruction whatever the
// address pattern (tools/micro/lds_atomic_bench.hip), i.e. ~0.8 TB/s chip-wide, while LDS
// INTEGER atomics are >6x faster.  So each workgroup sorts instead of scattering:
//
//   a workgroup owns the rows of (batch b, head m, level l, pixel range [px0, px1)) and
//   1. scans the level's Lq*P sampling points of (b, m), one lane per point, and counts the taps
//      that land on each of its rows        (LDS integer atomics: histogram)
//   2. prefix-sums the histogram            (row -> segment of the record array)
//   3. scans again and drops a record {weight, query} into the row's segment
//      (counting sort, LDS integer atomics for the cursor)
//   4. gathers: SLOTS x 8 lanes per row walk the row's segment, each record one coalesced 128-B
//      read of grad_out, accumulate in registers, combine the SLOTS partial sums with cross-lane
//      shuffles, and store the row once (zeros included) — no zero-fill pass, no atomics, every
//      element of grad_value written exactly once by exactly one workgroup.
//
// Every level is cut into the same number W of pixel ranges: all levels receive the same number
// of points, so equal counts of workgroups per level balance the gather work even though a coarse
// level has 4x fewer pixels.  W comes from the host (it only needs S and the batch size); the
// ranges come from spatial_shapes on the device.
// ACC (accumulation mode when Lq*P exceeds one pass's record array and queries go in chunks):
//   kAccNone  single pass: each row is stored once from registers;
//   kAccRmw   fp32 storage: pass 0 stores the rows, later passes read-add-write them in global memory
//             (the rows belong to this workgroup alone, passes are separated by __syncthreads, and
//             they stay in L2) — no LDS tile, so several workgroups fit per CU;
//   kAccTile  bf16 storage: rows accumulate in an fp32 LDS tile and are rounded ONCE at the flush.
//   kAccWide  fp32 grad_value, Lq*P beyond one pass: bwd_value_wide_body below — ONE pass over all Lq*P points
//             whenever the taps this workgroup KEEPS fit the record array (a workgroup owns 1/W of a level, so
//             it keeps ~1/W of the level's taps), chunked kAccRmw passes otherwise.
// ------------------------------------------------------------------------------------------
// DET (MSDA_FLAG_DETERMINISTIC): the same algorithm with a record order that is a pure function of the inputs.  What varies
// from run to run above is only the ORDER of a row's records — the rank an LDS atomic hands out depends on which wavefront
// got there first — and with it the association of the row's floating-point sum.  DET keeps one counter per (row,
// WAVEFRONT), eight 16-bit counters packed in four words per row: a wavefront's taps are ranked among themselves (lanes of
// one atomic instruction in lane order, instructions in program order — both fixed), the prefix sum runs over (row,
// wavefront) pairs, and a row's records end up wavefront-major in point order.  The kept-taps pass lists a wavefront's
// points in that wavefront's own list segment and every later pass has the same wavefront revisit them in the same order.
// It costs LDS (24 instead of 8 bytes per row), not a second algorithm: the reference's atomicAdd scatter
// (ms_deform_im2col_cuda.cuh:125-152) has no deterministic counterpart at all.
constexpr int kSBlock = 512;
constexpr int kSWaves = kSBlock / kWave;
static_assert(kSWaves == 8, "DET packs eight 16-bit per-wavefront counters into four words per row");

// DET counters: word (row * 4 + wave / 2), low half for even wavefronts.  Returns this wavefront's previous count.
__device__ __forceinline__ int det_count(int *cntw, int row, int wave)
{
    const int old = atomicAdd(&cntw[row * 4 + (wave >> 1)], (wave & 1) ? 0x10000 : 1);
    return (wave & 1) ? (int)((unsigned)old >> 16) : (old & 0xffff);
}
__device__ __forceinline__ void det_count_only(int *cntw, int row, int wave)
{
    atomicAdd(&cntw[row * 4 + (wave >> 1)], (wave & 1) ? 0x10000 : 1);
}
// Row `r` of the packed counters: returns the row's total and replaces the eight counts by their exclusive prefix within the row.
__device__ __forceinline__ int det_row_prefix(int *cntw, int r)
{
    int4 c = *reinterpret_cast<int4 *>(cntw + r * 4);
    const int w[4] = {c.x, c.y, c.z, c.w};
    int run = 0, out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lo = w[k] & 0xffff, hi = (int)((unsigned)w[k] >> 16);
        out[k] = run | ((run + lo) << 16);
        run += lo + hi;
    }
    *reinterpret_cast<int4 *>(cntw + r * 4) = make_int4(out[0], out[1], out[2], out[3]);
    return run;
}
// This wavefront's exclusive prefix within row `row` (after det_row_prefix).
__device__ __forceinline__ int det_wave_base(const int *cntw, int row, int wave)
{
    const int v = cntw[row * 4 + (wave >> 1)];
    return (wave & 1) ? (int)((unsigned)v >> 16) : (v & 0xffff);
}
constexpr int kAccNone = 0, kAccRmw = 1, kAccTile = 2, kAccWide = 3;
struct alignas(8) SRec { float w; int q; };
// Fixed-capacity segments (FIXED = true: steps 1-3 in one scan, for problems small enough that every workgroup
// is resident at once and the backward is one latency chain): row d owns the record slots [d*CAP, (d+1)*CAP),
// CAP = the largest power of two with rows*CAP <= the record array, so a tap's record goes straight to
// d*CAP + (its rank from the histogram atomic) — no prefix sum, no second scan, two barriers fewer.  The few
// taps whose row is already full go to a short overflow list that the gather of exactly those rows also walks;
// a workgroup whose overflow list fills up (many taps piled on few pixels) starts over on the prefix-sum path.
// Not used on large problems: there the sort phases hide behind other workgroups' gathers, and clustered
// locations make many rows walk the overflow list (profiles/r01_notes.md).
struct SOvf { float w; int q; int row; };
constexpr int kOvfCap = 256;                // overflow entries per workgroup (3 KB of LDS)

// role-B sizing: single pass while 4*Lq*P records (8 B) fit beside the histogram in 64 KB of LDS
constexpr int kSinglePPT = 3;               // points per thread: 3*512 = 1536 points, 48 KB of records
constexpr int kSingleMaxPoints = kSinglePPT * 512;
constexpr int kSingleMaxRows = 1920;        // 15 KB of histogram + prefix (rows per workgroup)
constexpr int kMultiRows = 256;             // 32 KB LDS tile
constexpr int kMultiPPT = 6;                // 3072 points per pass, 96 KB of records


// Rows of [0, S) that no fitting level covers get zeros (include/msda.h) — only shapes inconsistent with S have any: with
// level_start_index the running sum of H*W, as the reference's callers build it (models/arctic_transformer.py:176-177),
// the walk below finds no gap and stores nothing.  Called by ONE workgroup per (batch, head) pair.  Levels that overlap
// each other are not supported (each level's owner STORES its rows, the last store wins); gaps before, between and after
// the levels — including the rows of a level that was dropped because it does not fit — are handled here.
template <typename GT, int THREADS>
__device__ __forceinline__ void zero_uncovered_rows(const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
                                                    int S, int M, int L, GT *__restrict__ grad_value, int b, int m)
{
    // the usual case first, ONE batch of independent scalar loads: every level fits and starts where the previous one ends
    long long run = 0;
    bool tiled = true;
    for (int k = 0; k < L; ++k) {
        tiled = tiled && level_start[k] == run && level_fits(shapes[2 * k], shapes[2 * k + 1], level_start[k], S);
        run += shapes[2 * k] * shapes[2 * k + 1];
    }
    if (tiled && run == S) return;
    long long cur = 0;                                       // rows below `cur` are dealt with (uniform, scalar unit)
    while (cur < S) {
        long long ns = S, ne = S;                            // the covered interval [ns, ne) that ends after `cur` and starts first
        for (int k = 0; k < L; ++k) {
            if (!level_fits(shapes[2 * k], shapes[2 * k + 1], level_start[k], S)) continue;
            const long long s0 = level_start[k], e0 = s0 + shapes[2 * k] * shapes[2 * k + 1];
            if (e0 > cur && s0 < ns) { ns = s0; ne = e0; }
        }
        for (long long i = (long long)threadIdx.x; i < (ns - cur) * 8; i += THREADS)
            Row<GT>::store(grad_value + (((long long)b * S + cur + (i >> 3)) * M + m) * kD + (i & 7) * 4,
                           make_float4(0.f, 0.f, 0.f, 0.f));
        cur = ne > cur ? ne : S;
    }
}

__device__ __forceinline__ float4 shfl_xor4(const float4 &v, int m)
{
    return make_float4(__shfl_xor(v.x, m, kWave), __shfl_xor(v.y, m, kWave), __shfl_xor(v.z, m, kWave),
                       __shfl_xor(v.w, m, kWave));
}
__device__ __forceinline__ void add4(float4 &a, const float4 &b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// Step 4, rows in lockstep.  SLOTS lane-groups of 8 lanes share one row's segment; 8/SLOTS rows per wavefront trip
// and NR trips' rows in flight.  A lane group takes CH records of each of its NR rows at a time: the record reads,
// then NR*CH = 8 independent 128-B row loads, then the FMAs — one memory round trip per CH*SLOTS records of NR rows.
// (Used by the fixed-capacity sort, the LDS-tile passes and the single-pass path's coarse levels; gather_balanced /
// gather_split below walk contiguous segments instead.)
// Where a gather reads the grad_out row of query q of its (batch, head) pair from (lane j's 16 bytes of it): a BUFFER
// descriptor over the pair's rows [(b, 0, m), (b, Lq-1, m)] with 32-bit byte offsets q * (M * row bytes) + j * 16 — one
// 24-bit multiply-add per record instead of a 64-bit address pair and a slow 32-bit multiply, and a padding slot carries the
// query index Lq, whose offset lies past the descriptor: the hardware returns zeros, no select.  (The SQ counters of round 4
// put these kernels at ~1000 vector instructions per wavefront, most of them address arithmetic and selects.)
template <typename VT> struct GoBuf {
    __amdgpu_buffer_rsrc_t rs; unsigned joff; int stride_b, lq;
    static constexpr bool kPadBySelect = false;
    __device__ __forceinline__ int none() const { return lq; }
    __device__ __forceinline__ float4 load(int q) const { return BufRow<VT>::load(rs, (unsigned)__mul24(q, stride_b) + joff); }
    __device__ __forceinline__ float4 load_at(unsigned row_off) const { return BufRow<VT>::load(rs, row_off + joff); }
};
// (d32_supported() bounds Lq * M * 128 B below 2^31 and Lq below 2^23)
template <typename VT>
__device__ __forceinline__ GoBuf<VT> pair_rows(const VT *grad_out, long long item_base, int Lq, int M, int j)
{
    const long long bytes = (((long long)Lq - 1) * M + 1) * kD * (long long)sizeof(VT);
    return GoBuf<VT>{uniform_rsrc(grad_out + item_base * kD, bytes), (unsigned)(j * 4 * sizeof(VT)), M * kD * (int)sizeof(VT), Lq};
}

#ifndef MSDA_GATHER_CH1
#define MSDA_GATHER_CH1 0        // gather_rows, fixed-capacity path, one lane group per row: records of a row per trip (0: 8 / rows in flight)
#endif
template <int SLOTS, int ACC, typename VT, typename GT, int NR1 = 2, typename GO = GoBuf<VT>, int CH1 = 0>
__device__ __forceinline__ void gather_rows(const GO go, GT *__restrict__ gv_base,
                                            const int *cnt, const int *start, const SRec *rec, float *tile,
                                            int npx, int row_stride, bool first_pass, int cap = -1,
                                            const SOvf *ovf = nullptr, int novf = 0)
{
    // cap > 0: fixed-capacity segments (row d at d * cap, at most cap records there, the rest of a fuller row in
    // ovf[0, novf)); cap < 0: segments from the prefix sum (start[]).  (24-bit multiplies: full rate.)
    constexpr int DPW = 8 / SLOTS;
    // rows in flight per lane group x records of each per trip: 8 loads either way.  NR1 (rows in flight when ONE lane
    // group serves a row) is 4 on the fixed-capacity path: such rows hold a couple of records (cfg-2 decoder's 48x48
    // level: 2), and four rows x 2 keeps the loads real — cfg-2 decoder backward 13.5 -> 13.2 us in kbench.  Elsewhere 2:
    // the wider variant costs the other instantiations 7 VGPRs and cfg-4 decoder 2 us (tools/exp_nr.sh).
    constexpr int NR = SLOTS == 1 ? NR1 : 2;
    constexpr int CH = (SLOTS == 1 && CH1 > 0) ? CH1 : 8 / NR;
    constexpr int RSTEP = kSWaves * DPW;                     // rows per workgroup trip
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int dsub = lane / (SLOTS * 8), slot = (lane >> 3) % SLOTS, j = lane & 7;
    for (int d0 = wave * DPW + dsub; d0 < npx; d0 += NR * RSTEP) {
        int full[NR], n[NR]; const SRec *rp[NR]; float4 acc[NR];
        int nmax = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int d = d0 + r * RSTEP;
            const bool has = d < npx;
            full[r] = has ? cnt[d] : 0;
            n[r] = cap > 0 ? min(full[r], cap) : full[r];
            rp[r] = rec + (!has ? 0 : cap > 0 ? __mul24(d, cap) : start[d]);
            acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            nmax = max(nmax, n[r]);
        }
        for (int i0 = slot; i0 < nmax; i0 += CH * SLOTS) {
            SRec e[NR][CH]; float4 g[NR][CH];
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    // unconditional read + two selects (a slot past the row's records still lies inside the LDS arrays: the
                    // record array is followed by the overflow list); a guarded read costs an exec-mask round trip per slot
                    const SRec t = rp[r][i0 + u * SLOTS];
                    const bool ok = i0 + u * SLOTS < n[r];
                    e[r][u].w = ok ? t.w : 0.f; e[r][u].q = ok ? t.q : go.none();
                }
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float4 t = MSDA_DIAG(2) ? make_float4(1.f, 1.f, 1.f, 1.f) : go.load(e[r][u].q);
                    if (GO::kPadBySelect) {
                        const bool ok = e[r][u].q >= 0;
                        g[r][u] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
                    } else
                        g[r][u] = t;
                }
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int u = 0; u < CH; ++u) fma4(acc[r], e[r][u].w, g[r][u]);
        }
        if (novf > 0) {                                                 // rare: one of this lane group's rows overflowed
            bool any = false;
#pragma unroll
            for (int r = 0; r < NR; ++r) any = any || full[r] > n[r];
            if (any) {
                for (int i = slot; i < novf; i += SLOTS) {
                    const SOvf o = ovf[i];
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (o.row == d0 + r * RSTEP && d0 + r * RSTEP < npx)
                            fma4(acc[r], o.w, go.load(o.q));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (SLOTS >= 2) add4(acc[r], shfl_xor4(acc[r], 8));
            if (SLOTS >= 4) add4(acc[r], shfl_xor4(acc[r], 16));
            if (SLOTS >= 8) add4(acc[r], shfl_xor4(acc[r], 32));
        }
        if (slot == 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int d = d0 + r * RSTEP;
                if (d >= npx) continue;
                if (ACC == kAccTile) {
                    float4 *t = reinterpret_cast<float4 *>(tile) + d * 8 + j;
                    if (first_pass) *t = acc[r]; else { float4 o = *t; add4(o, acc[r]); *t = o; }
                } else {
                    GT *pr = gv_base + (long long)__mul24(d, row_stride);      // (rows per workgroup and M*32 are both far below 2^23)
                    if (ACC == kAccRmw && !first_pass) add4(acc[r], Row<GT>::load(pr));
                    Row<GT>::store(pr, acc[r]);
                }
            }
        }
    }
}

// How a gather reads record i: {weight, element offset of its grad_out row}.  RecAos: the 8-byte {weight, query}
// records of the chunked / single-pass paths.  RecSoa (kAccWide): a float weight array and a 16-bit array of queries
// relative to the chunk's first query — 6 bytes per record, so a third more records fit the same LDS and batch-heavy
// shapes need fewer pixel ranges per level (cfg-4 encoder: 2 instead of 3, each range re-scans the level).
// (`off`: the row's byte offset inside the pair's descriptor, GoBuf::load_at; stride_b = M * row bytes, 24-bit multiplies)
struct RecAos {
    const SRec *r; int stride_b;
    __device__ __forceinline__ void get(int i, float &w, unsigned &off) const { const SRec e = r[i]; w = e.w; off = (unsigned)__mul24(e.q, stride_b); }
};
struct RecSoa {
    const float *w; const uint16_t *q; int qbase, stride_b;
    __device__ __forceinline__ void get(int i, float &wt, unsigned &off) const { wt = w[i]; off = (unsigned)__mul24(qbase + (int)q[i], stride_b); }
};

// ENDS: start[r] is the END of row r's segment (kAccWide's scatter cursor), else its beginning.
template <typename VT, typename GT, bool ENDS, typename RV>
__device__ __forceinline__ void gather_balanced(const GoBuf<VT> go, GT *__restrict__ gv_base,
                                                const int *cnt, const int *start, const RV rec, int *firsts, int npx,
                                                int row_stride, int total, int first_pass)
{
    constexpr int G = kSBlock / 8, CH = 8;
    const int tid = threadIdx.x, g = tid >> 3;
    struct { const int *cnt, *start; __device__ int operator[](int r) const { return ENDS ? start[r] : start[r] + cnt[r]; } } endv{cnt, start};
    // first row of this lane group: rows are dealt by weight = records + 2 (a row costs a store and a few
    // instructions even when empty: without the "+ 2" the lane group after a cluster would inherit every empty row
    // behind it), i.e. the smallest r with begin[r] + 2 r >= g * (total + 2 npx) / G
    const int lo = (g * (total + 2 * npx)) / G;              // total <= 12 288 records, npx <= 1920 rows: no overflow
    int a = 0, b = npx;
    while (a < b) {
        const int mid = (a + b) >> 1;
        if ((mid ? endv[mid - 1] : 0) + 2 * mid >= lo) b = mid; else a = mid + 1;
    }
    if ((tid & 7) == 0) firsts[g] = a;
    if (tid == 0) firsts[G] = npx;
    __syncthreads();
    int r = a;
    const int r_stop = firsts[g + 1];
    if (r >= r_stop) return;                                 // (no barrier below)
    int i = r ? endv[r - 1] : 0;
    const int i_stop = endv[r_stop - 1];
    // A flush is issued under the execution mask of the lane groups whose row just ended — 8 groups at different records, so
    // some group flushes at almost every other record and the flush's instructions weigh like the loop's own (round 4: ~25
    // of them, with a prefetched "end of the row after next" and a 64-bit row counter; now the store, the pointer, the
    // accumulator and ONE LDS read of the next row's end — endv[r_stop] is always a readable LDS word, and unused).
    const int fp = __builtin_amdgcn_readfirstlane(first_pass);      // a scalar, not a lane mask: one s_cmp per flush
    int row_end = endv[r];
    GT *prow = gv_base + (long long)r * row_stride;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto flush = [&]() {                                     // row r is complete (possibly empty)
        if (!fp) add4(acc, Row<GT>::load(prow));
        Row<GT>::store(prow, acc);
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        prow += row_stride;
        ++r;
        row_end = endv[r];
    };
    for (; i + CH <= i_stop; i += CH) {
        float ew[CH]; unsigned eo[CH]; float4 gl[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) rec.get(i + u, ew[u], eo[u]);
#pragma unroll
        for (int u = 0; u < CH; ++u) gl[u] = MSDA_DIAG(2) ? make_float4(1.f, 1.f, 1.f, 1.f) : go.load_at(eo[u]);
#pragma unroll
        for (int u = 0; u < CH; ++u) { while (i + u >= row_end) flush(); fma4(acc, ew[u], gl[u]); }   // (i + u < i_stop: r stays below r_stop)
    }
    if (i < i_stop) {                                        // last, partial batch
        float ew[CH]; unsigned eo[CH]; float4 gl[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (i + u < i_stop) { rec.get(i + u, ew[u], eo[u]); gl[u] = go.load_at(eo[u]); }
#pragma unroll
        for (int u = 0; u < CH; ++u) if (i + u < i_stop) { while (i + u >= row_end) flush(); fma4(acc, ew[u], gl[u]); }
    }
    while (r < r_stop) flush();                              // the row in progress and the empty rows after it
}

// gather_balanced for a record array whose rows are padded to multiples of 8 records (zero weight, a query past the
// descriptor; bwd_value_wide_body): a batch of 8 never straddles two rows, so a row's end is looked for once per batch
// instead of once per record — the per-record test and its execution-mask round trip were a quarter of the walk's
// instructions (~17 per record step, 8 lanes groups at a time).  ends[r] = end of row r's (padded) segment.
template <typename VT, typename GT, typename RV>
__device__ __forceinline__ void gather_pad8(const GoBuf<VT> go, GT *__restrict__ gv_base, const int *ends, const RV rec, int *firsts,
                                            int npx, int row_stride, int total, int first_pass)
{
    constexpr int G = kSBlock / 8, CH = 8;
    const int tid = threadIdx.x, g = tid >> 3;
    const int lo = (g * (total + 2 * npx)) / G;              // rows dealt by weight = slots + 2, as gather_balanced does
    int a = 0, b = npx;
    while (a < b) {
        const int mid = (a + b) >> 1;
        if ((mid ? ends[mid - 1] : 0) + 2 * mid >= lo) b = mid; else a = mid + 1;
    }
    if ((tid & 7) == 0) firsts[g] = a;
    if (tid == 0) firsts[G] = npx;
    __syncthreads();
    int r = a;
    const int r_stop = firsts[g + 1];
    if (r >= r_stop) return;                                 // (no barrier below)
    int i = r ? ends[r - 1] : 0;
    const int i_stop = ends[r_stop - 1];
    const int fp = __builtin_amdgcn_readfirstlane(first_pass);
    int row_end = ends[r];
    GT *prow = gv_base + (long long)r * row_stride;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto flush = [&]() {                                     // row r is complete (possibly empty); ends[r_stop] is a readable word
        if (!fp) add4(acc, Row<GT>::load(prow));
        Row<GT>::store(prow, acc);
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        prow += row_stride;
        ++r;
        row_end = ends[r];
    };
    for (; i < i_stop; i += CH) {                            // (i, row ends and i_stop are multiples of 8)
        float ew[CH]; unsigned eo[CH]; float4 gl[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) rec.get(i + u, ew[u], eo[u]);
#pragma unroll
        for (int u = 0; u < CH; ++u) gl[u] = MSDA_DIAG(2) ? make_float4(1.f, 1.f, 1.f, 1.f) : go.load_at(eo[u]);
        while (i >= row_end) flush();                        // the rows that ended before this batch (i < i_stop: r stays below r_stop)
#pragma unroll
        for (int u = 0; u < CH; ++u) fma4(acc, ew[u], gl[u]);
    }
    while (r < r_stop) flush();                              // the row in progress and the empty rows after it
}

// Step 4 for rows of very different lengths (a coarse level: five rows of 1400 records): the sorted records are cut
// into 64 EQUAL stretches whatever the rows are.  A lane group walks its stretch eight records at a time; rows that
// end inside it are stored (the group that holds a row's FIRST record owns the row); the part of a row that began in
// an earlier stretch goes to part[g] in LDS (a lane group has at most one: only its first row can have begun earlier),
// and after one barrier the owner adds the parts of the groups behind it in group order — fixed association, no
// atomics.  Rows without records are stored as zeros up front.  gather_rows gives such rows one wavefront each
// (five of eight busy, 8 slots x 4 loads in flight): gather of a workgroup of cfg-2 encoder's 6x6 level 20.2 -> 11.3 us,
// cfg-4 encoder backward 301 -> 272 us (half its workgroups belong to coarse levels); profiles/r02_notes.md §10.
template <typename VT, typename GT, bool ENDS, typename RV>
__device__ __forceinline__ void gather_split(const GoBuf<VT> go, GT *__restrict__ gv_base, const int *cnt,
                                             const int *start, const RV rec, float4 *part, int npx, int row_stride,
                                             int total, int first_pass)
{
    constexpr int G = kSBlock / 8, CH = 8;
    const int tid = threadIdx.x, g = tid >> 3, j = tid & 7;
    struct { const int *cnt, *start; __device__ int operator[](int r) const { return ENDS ? start[r] : start[r] + cnt[r]; } } endv{cnt, start};
    auto put = [&](int r, float4 v) {
        GT *p = gv_base + (long long)r * row_stride;
        if (!first_pass) add4(v, Row<GT>::load(p));
        Row<GT>::store(p, v);
    };
    for (int r = g; r < npx; r += G)                          // rows nobody will visit
        if (cnt[r] == 0 && first_pass) Row<GT>::store(gv_base + (long long)r * row_stride, make_float4(0.f, 0.f, 0.f, 0.f));
    const int lo = (int)(((long long)g * total) / G), hi = (int)(((long long)(g + 1) * total) / G);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int r = 0, own_end = 0;                                   // own_end > 0: this group owns row r, which runs on to record own_end
    if (lo < hi) {
        int a = 0, b = npx - 1;                               // the row of record lo: smallest r with endv[r] > lo
        while (a < b) { const int mid = (a + b) >> 1; if (endv[mid] > lo) b = mid; else a = mid + 1; }
        r = a;
        bool foreign = (r ? endv[r - 1] : 0) < lo;            // the row began in an earlier stretch
        int row_end = endv[r];
        auto close_row = [&]() {                              // row r's records in this stretch are all in acc
            if (foreign) part[g * 8 + j] = acc; else put(r, acc);
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            foreign = false;
            do { ++r; } while (r < npx && endv[r] == row_end);    // skip rows without records
            row_end = r < npx ? endv[r] : 0x7fffffff;
        };
        int i = lo;
        for (; i + CH <= hi; i += CH) {
            float ew[CH]; unsigned eo[CH]; float4 gl[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) rec.get(i + u, ew[u], eo[u]);
#pragma unroll
            for (int u = 0; u < CH; ++u) gl[u] = go.load_at(eo[u]);
#pragma unroll
            for (int u = 0; u < CH; ++u) { if (i + u >= row_end) close_row(); fma4(acc, ew[u], gl[u]); }
        }
        if (i < hi) {
            float ew[CH]; unsigned eo[CH]; float4 gl[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (i + u < hi) { rec.get(i + u, ew[u], eo[u]); gl[u] = go.load_at(eo[u]); }
#pragma unroll
            for (int u = 0; u < CH; ++u) if (i + u < hi) { if (i + u >= row_end) close_row(); fma4(acc, ew[u], gl[u]); }
        }
        // the row in progress: complete if it ends with the stretch, else it runs on into later stretches
        if (row_end <= hi) close_row();
        else if (foreign) part[g * 8 + j] = acc;              // the whole stretch lies inside a row owned further up
        else own_end = row_end;
    }
    __syncthreads();
    if (own_end > 0) {
        for (int g2 = g + 1; g2 < G && (int)(((long long)g2 * total) / G) < own_end; ++g2) {
            // a stretch may be empty (total < 64): it wrote nothing
            if ((int)(((long long)g2 * total) / G) < (int)(((long long)(g2 + 1) * total) / G)) add4(acc, part[g2 * 8 + j]);
        }
        put(r, acc);
    }
}

template <typename VT, typename GT, bool DET>
__device__ __forceinline__ void bwd_value_wide_body(const VT *__restrict__, const int64_t *__restrict__, const int64_t *__restrict__,
                                                    const float *__restrict__, const float *__restrict__, int, int, int, int, int,
                                                    int, int, GT *__restrict__, int, int, int, int, unsigned char *, const MaskIn);   // kAccWide, below

// PPT = sampling points per thread per pass: all of a thread's points are loaded up front (2*PPT
// independent global loads in flight), their taps and histogram ranks stay in registers between
// step 1 and step 3, so loc / attn are read exactly once and step 3 needs no atomics.
// VT = storage type of grad_out, GT = storage type of grad_value (the same, or float for bf16 rows with an
// fp32 grad_value: no rounding between passes, kAccRmw instead of the LDS tile).
template <int ACC, int PPT, typename VT, typename GT = VT, bool FIXED = false, bool DET = false>
__device__ __forceinline__ void bwd_value_body(
    const VT *__restrict__ grad_out, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc,
    const float *__restrict__ attn, int S, int M, int L, int Lq, int P, int p_shift, int tp_cap,
    GT *__restrict__ grad_value, int ti, int W, int l, int pr, unsigned char *smem,
    const MaskIn mi = MaskIn{nullptr, nullptr})
{
    static_assert(!(FIXED && DET), "the fixed-capacity segments take ranks in arrival order");
    if constexpr (ACC == kAccWide) {
        bwd_value_wide_body<VT, GT, DET>(grad_out, shapes, level_start, loc, attn, S, M, L, Lq, P, p_shift, tp_cap, grad_value,
                                         ti, W, l, pr, smem, mi);
        return;
    }
    constexpr int NPC = PPT * kSBlock;                       // points per pass
    // LDS: [tile: tp_cap*32 floats if kAccTile] [cnt tp_cap (DET: 4 words per row)] [start tp_cap] [DET: row totals tp_cap]
    //      [wsum 32] [rec] [ovf kOvfCap]
    constexpr int CW = DET ? 4 : 1;
    float *tile = reinterpret_cast<float *>(smem);
    int *cnt = reinterpret_cast<int *>(smem + (ACC == kAccTile ? (size_t)tp_cap * kD * 4 : 0));
    int *start = cnt + tp_cap * CW;
    int *tot = DET ? start + tp_cap : cnt;                   // records per row, as the gathers read them
    int *wsum = DET ? tot + tp_cap : start + tp_cap;
    SRec *rec = reinterpret_cast<SRec *>(wsum + 32);
    int *novf_p = wsum + 8, *total_p = wsum + 9;             // wsum[0..7]: per-wavefront sums of the prefix scan, [16..23]: longest row

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    MSDA_STAMP(0);
    // (ti of W ranges, level l, pair pr = b*M + m): all uniform, scalar unit; 32-bit on purpose (the
    // host checks S*W < 2^31).
    const int H = (int)shapes[2 * l], Wd = (int)shapes[2 * l + 1], lstart = (int)level_start[l];
    const int HW = H * Wd;
    const int px0 = (int)((unsigned)(ti * HW) / (unsigned)W), px1 = (int)((unsigned)((ti + 1) * HW) / (unsigned)W);
    const int npx = px1 - px0;
    const int b = pr / M, m = pr - b * M;
    if (l == 0 && ti == 0) zero_uncovered_rows<GT, kSBlock>(shapes, level_start, S, M, L, grad_value, b, m);
    // empty range, or a level that does not fit in S (then nothing of it is touched).  A level that fits has
    // npx <= tp_cap by construction of the ranges (plan_value: tp_cap = ceil(S / W) >= ceil(H*W / W_l)).
    if (npx <= 0 || npx > tp_cap || !level_fits(shapes[2 * l], shapes[2 * l + 1], level_start[l], S)) return;
    const int NP = Lq * P;
    const long long item_base = (long long)b * Lq * M + m;                   // item(q) = item_base + q*M
    const int row_stride = M * kD;
    const int rec_cap = 4 * min(NP, NPC);                                    // records the host sized `rec` for
    SOvf *ovf = reinterpret_cast<SOvf *>(rec + rec_cap);
    // fixed-capacity segments need a few slots per row (uniform; small maps with few queries go the prefix way)
    const int cap_shift = (FIXED && rec_cap >= 4 * npx) ? 31 - __clz(rec_cap / npx) : -1;
    const GoBuf<VT> go = pair_rows<VT>(grad_out, item_base, Lq, M, lane & 7);
    GT *gv_base = grad_value + ((long long)(b * S + lstart + px0) * M + m) * kD + (lane & 7) * 4;

    // single-pass kernels (the host only picks them when NP <= NPC): a visible trip count of one lets the
    // compiler keep the per-pass arrays out of the loop-carried state (103 -> 90 VGPRs)
    const int np_loop = ACC == kAccNone ? min(NP, NPC) : NP;
    for (int c0 = 0; c0 < np_loop; c0 += NPC) {
        // ---- loads of this pass's points first: they overlap the histogram reset ----
        float2 xy[PPT]; float at[PPT]; int qq[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int idx = c0 + tid + k * kSBlock;
            qq[k] = -1; xy[k] = make_float2(-8.f, -8.f); at[k] = 0.f;
            if (idx < NP) {
                const int q = fdiv(idx, P, p_shift), p = idx - q * P;
                const long long pi = ((item_base + (long long)q * M) * L + l) * P + p;
                xy[k] = reinterpret_cast<const float2 *>(loc)[pi];
                at[k] = attn[pi];
                qq[k] = q;
            }
        }
        for (int i = tid; i < npx * CW; i += kSBlock) cnt[i] = 0;
        if (FIXED && tid == 0) { *novf_p = 0; *total_p = 0; }
        __syncthreads();
        MSDA_STAMP(1);
        // ---- 1. taps of each point ----
        int dest[PPT][4]; float tw[PPT][4];
        const bool first = (c0 == 0);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const PointGeom<float> g = point_geom<float>(xy[k].x, xy[k].y, H, Wd);
            const int pix = g.h0 * Wd + g.w0 - px0;                          // range-local index of tap (h0, w0)
            const int p01 = pix + 1, p10 = pix + Wd, p11 = pix + Wd + 1;
            const bool live = g.inside && qq[k] >= 0;
            dest[k][0] = (live && g.ok00 && pix >= 0 && pix < npx) ? pix : -1;
            dest[k][1] = (live && g.ok01 && p01 >= 0 && p01 < npx) ? p01 : -1;
            dest[k][2] = (live && g.ok10 && p10 >= 0 && p10 < npx) ? p10 : -1;
            dest[k][3] = (live && g.ok11 && p11 >= 0 && p11 < npx) ? p11 : -1;
            const float hh = 1.f - g.lh, hw = 1.f - g.lw;
            tw[k][0] = hh * hw * at[k]; tw[k][1] = hh * g.lw * at[k];
            tw[k][2] = g.lh * hw * at[k]; tw[k][3] = g.lh * g.lw * at[k];
        }
        // ---- 1-3 the short way: fixed-capacity segments, a tap's record goes straight to its slot ----
        if (FIXED && cap_shift >= 0) {
            // all the histogram atomics first (they pipeline: nothing waits on a returned rank yet), then the
            // record writes
            int mine_taps = 0, rk[PPT][4];
#pragma unroll
            for (int k = 0; k < PPT; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) { rk[k][t] = 0; if (dest[k][t] >= 0) { rk[k][t] = atomicAdd(&cnt[dest[k][t]], 1); ++mine_taps; } }
#pragma unroll
            for (int k = 0; k < PPT; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (dest[k][t] >= 0) {
                        const int r = rk[k][t];
                        if (r < (1 << cap_shift)) { SRec e; e.w = tw[k][t]; e.q = qq[k]; rec[(dest[k][t] << cap_shift) + r] = e; }
                        else {
                            const int o = atomicAdd(novf_p, 1);
                            if (o < kOvfCap) { SOvf e; e.w = tw[k][t]; e.q = qq[k]; e.row = dest[k][t]; ovf[o] = e; }
                        }
                    }
            mine_taps = wave_sum(mine_taps);
            if (lane == 0) atomicAdd(total_p, mine_taps);
            __syncthreads();
            const int novf = *novf_p;
            if (novf <= kOvfCap) {
                MSDA_STAMP(2); MSDA_STAMP(3); MSDA_STAMP(4);
                const int mean2f = (2 * *total_p) / npx;
                if (mean2f <= 8)       gather_rows<1, ACC, VT, GT, MSDA_GATHER_NR1>(go, gv_base, cnt, start, rec, tile, npx, row_stride, first, 1 << cap_shift, ovf, novf);
                else if (mean2f <= 16) gather_rows<2, ACC, VT, GT>(go, gv_base, cnt, start, rec, tile, npx, row_stride, first, 1 << cap_shift, ovf, novf);
                else if (mean2f <= 32) gather_rows<4, ACC, VT, GT>(go, gv_base, cnt, start, rec, tile, npx, row_stride, first, 1 << cap_shift, ovf, novf);
                else                   gather_rows<8, ACC, VT, GT>(go, gv_base, cnt, start, rec, tile, npx, row_stride, first, 1 << cap_shift, ovf, novf);
                if (ACC != kAccNone) __syncthreads();
                MSDA_STAMP(5);
                continue;
            }
            // the overflow list filled up (taps piled on few pixels): start over on the prefix-sum path
            for (int i = tid; i < npx; i += kSBlock) cnt[i] = 0;
            __syncthreads();
        }
        // ---- 1. histogram rank of each tap on its row ----
        int rank[PPT][4];
#pragma unroll
        for (int k = 0; k < PPT; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                rank[k][t] = 0;
                if (dest[k][t] >= 0) rank[k][t] = DET ? det_count(cnt, dest[k][t], wave) : atomicAdd(&cnt[dest[k][t]], 1);
            }
        __syncthreads();
        MSDA_STAMP(2);
        // ---- 2. exclusive prefix sum over the rows (512 threads x CH consecutive rows) ----
        const int CH = (npx + kSBlock - 1) / kSBlock;
        const int r0 = tid * CH;
        int mine = 0, big = 0;
        if (DET) {                                           // row totals; the packed counts become within-row prefixes
            for (int k = 0; k < CH; ++k) if (r0 + k < npx) { const int t = det_row_prefix(cnt, r0 + k); tot[r0 + k] = t; mine += t; big = max(big, t); }
        } else
        for (int k = 0; k < CH; ++k) if (r0 + k < npx) { mine += cnt[r0 + k]; big = max(big, cnt[r0 + k]); }
        int incl = mine;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) { const int y = __shfl_up(incl, o, kWave); if (lane >= o) incl += y; }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) big = max(big, __shfl_xor(big, o, kWave));
        if (lane == kWave - 1) { wsum[wave] = incl; wsum[16 + wave] = big; }
        __syncthreads();
        int excl = incl - mine, total = 0, longest = 0;
        {
            const int4 wa = *reinterpret_cast<const int4 *>(wsum), wb = *reinterpret_cast<const int4 *>(wsum + 4);
            const int ws[kSWaves] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int w2 = 0; w2 < kSWaves; ++w2) { if (w2 < wave) excl += ws[w2]; total += ws[w2]; }
            const int4 ma = *reinterpret_cast<const int4 *>(wsum + 16), mb = *reinterpret_cast<const int4 *>(wsum + 20);
            longest = max(max(max(ma.x, ma.y), max(ma.z, ma.w)), max(max(mb.x, mb.y), max(mb.z, mb.w)));
        }
        for (int k = 0; k < CH; ++k) if (r0 + k < npx) { start[r0 + k] = excl; excl += tot[r0 + k]; }
        __syncthreads();
        MSDA_STAMP(3);
        // ---- 3. counting sort: each tap's record goes to start[row] (+ DET: this wavefront's share of the row) + rank ----
#pragma unroll
        for (int k = 0; k < PPT; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (dest[k][t] >= 0) {
                    SRec r; r.w = tw[k][t]; r.q = qq[k];
                    rec[start[dest[k][t]] + (DET ? det_wave_base(cnt, dest[k][t], wave) : 0) + rank[k][t]] = r;
                }
        __syncthreads();
        MSDA_STAMP(4);
        // ---- 4. gather; lanes per row chosen from the mean segment length (uniform) ----
        // rows of comparable length (the longest no more than one lane group's share): the balanced walk
        if (ACC != kAccTile && longest * 64 <= max(total, 1024)) {
            gather_balanced<VT, GT, false>(go, gv_base, tot, start, RecAos{rec, go.stride_b}, reinterpret_cast<int *>(ovf), npx,
                                           row_stride, total, ACC == kAccNone || first);
            if (ACC != kAccNone) __syncthreads();
            MSDA_STAMP(5);
            continue;
        }
        const int mean2 = (2 * total) / npx;                                  // 2 x mean records per row
        if (mean2 <= 8)       gather_rows<1, ACC, VT, GT>(go, gv_base, tot, start, rec, tile, npx, row_stride, first);
        else if (mean2 <= 16) gather_rows<2, ACC, VT, GT>(go, gv_base, tot, start, rec, tile, npx, row_stride, first);
        else if (mean2 <= 32) gather_rows<4, ACC, VT, GT>(go, gv_base, tot, start, rec, tile, npx, row_stride, first);
        else                  gather_rows<8, ACC, VT, GT>(go, gv_base, tot, start, rec, tile, npx, row_stride, first);
        if (ACC != kAccNone) __syncthreads();                // next pass reuses the LDS arrays / re-reads rows
        MSDA_STAMP(5);
    }

    if (ACC == kAccTile) {
        // ---- flush the LDS tile: each row once, coalesced ----
        for (int i = tid; i < npx * 8; i += kSBlock) {
            const int d = i >> 3, jj = i & 7;
            const float4 v = NP > 0 ? reinterpret_cast<const float4 *>(tile)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            Row<GT>::store(grad_value + ((long long)(b * S + lstart + px0 + d) * M + m) * kD + jj * 4, v);
        }
    }
}


// ---- small problems: every workgroup of the launch resident at once (plan_fused) -------------------------------------
// The 300-query decoder shape: 406 workgroups on 256 CUs, ONE role-B workgroup per CU, every phase of it a latency chain, and
// — by the SQ counters and in-kernel stamps of round 4 (profiles/r04_notes.md) — role B alone sets the launch's length.
// bwd_value_small_body is role B for exactly that regime (one pass, Lq*P <= kSingleMaxPoints, not deterministic):
//   * the points come from the level-major table the FORWARD of the same autograd node left in the caller's scratch
//     (PointEntry, msda_forward_ws_*): 16 B per point, coalesced, geometry already done — instead of a strided scan of
//     sampling_loc / attn_weight that touches one 128-B line per query for 32 useful bytes (without a table: that scan);
//   * COMPACTION before the histogram: a workgroup owns 1/W of its level's rows, so ~3/4 of its points have no tap there,
//     and an LDS atomic costs ~30 cycles per wavefront instruction however few lanes take part — twelve mostly-empty atomic
//     instructions per wavefront were the longest phase.  Each wavefront lists the points that have a tap on its rows in its
//     own LDS segment (ballot + prefix count: no atomics, no barrier — a wavefront's LDS operations execute in order) and
//     then works through the list with dense lanes: four atomics and four record writes per 64 listed points;
//   * fixed-capacity record slots (kSmallRecCap / rows per row; the rank from the histogram atomic is the slot), a 256-entry
//     overflow list;
//   * taps piled on few pixels (the list fills up), or fewer than four slots per row: the workgroup starts over in
//     bwd_value_body's prefix-sum sort (the launch's LDS is sized for both layouts).
// (Measured and not kept, profiles/r04_notes.md: the pair's grad_out rows staged in LDS for the gather — 1.1 us more in the
// load phase for 1.0 us less in the gather; a flat per-lane-group walk over compacted records instead of rows in lockstep —
// half the loads, 1.2 us slower.)
constexpr int kSmallRecCap = 4096;                  // record slots (32 KB): 8 per row of a 48x48 level's fifth, 256+ of a 6x6 level's third
constexpr int kSmallListCap = 96;                   // listed points per wavefront (of its 192); the rest take the sparse path

// (det: four counter words per row + the row totals instead of one counter per row, and two words of 16-bit ranks per listed point)
__host__ __device__ inline size_t small_lds_bytes(int tp_cap, bool det = false)
{
    return ((size_t)(det ? 5 : 2) * tp_cap + 32) * 4 + (size_t)kSmallRecCap * sizeof(SRec) + (size_t)kOvfCap * sizeof(SOvf) +
           (size_t)kSWaves * kSmallListCap * (sizeof(PointEntry) + 4 + (det ? 8 : 0));
}

// The four taps of a point on this workgroup's rows [px0, px0 + npx) of a W-wide level: range-local destinations (-1 = not
// mine), from a table entry; small_weights: their weights (bilinear x attention).
__device__ __forceinline__ void small_dests(const PointEntry &e, int Wd, int px0, int npx, int (&dest)[4])
{
    const int okb = e.cell >> 24, pix = (e.cell & 0xffffff) - (Wd + 1) - px0;
    const int p01 = pix + 1, p10 = pix + Wd, p11 = pix + Wd + 1;
    dest[0] = ((okb & 1) && pix >= 0 && pix < npx) ? pix : -1;
    dest[1] = ((okb & 2) && p01 >= 0 && p01 < npx) ? p01 : -1;
    dest[2] = ((okb & 4) && p10 >= 0 && p10 < npx) ? p10 : -1;
    dest[3] = ((okb & 8) && p11 >= 0 && p11 < npx) ? p11 : -1;
}
__device__ __forceinline__ void small_weights(const PointEntry &e, float (&tw)[4])
{
    const float hh = 1.f - e.lh, hw = 1.f - e.lw;
    tw[0] = hh * hw * e.a; tw[1] = hh * e.lw * e.a; tw[2] = e.lh * hw * e.a; tw[3] = e.lh * e.lw * e.a;
}

// DET (MSDA_FLAG_DETERMINISTIC, round 5): the same body with a record order that is a pure function of the inputs — one
// 16-bit counter per (row, wavefront) (det_count), ranks taken in list order, one more barrier for the rows' totals and
// per-wavefront bases (det_row_prefix), records written behind it from the list (ranks kept in a word per listed point);
// a row with more records than its fixed-capacity segment holds sends the workgroup to the general deterministic body.
template <typename VT, typename GT, bool DET = false>
__device__ __forceinline__ void bwd_value_small_body(
    const VT *__restrict__ grad_out, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, const PointEntry *__restrict__ table,
    const RangeHeader *__restrict__ header, int S, int M, int L, int Lq, int P, int p_shift, int tp_cap, GT *__restrict__ grad_value,
    int bid, int W_plan, unsigned char *smem)
{
    // LDS: [cnt tp_cap (DET: 4 words per row)] [start tp_cap (DET: the rows' totals)] [wsum 32] [rec kSmallRecCap] [ovf kOvfCap]
    //      [listed points: 8 x kSmallListCap entries] [their queries] [DET: their ranks]
    constexpr int CW = DET ? 4 : 1;
    int *cnt = reinterpret_cast<int *>(smem);
    int *start = cnt + tp_cap * CW;
    int *wsum = start + tp_cap;
    SRec *rec = reinterpret_cast<SRec *>(wsum + 32);
    SOvf *ovf = reinterpret_cast<SOvf *>(rec + kSmallRecCap);
    PointEntry *list = reinterpret_cast<PointEntry *>(ovf + kOvfCap);
    int *list_q = reinterpret_cast<int *>(list + kSWaves * kSmallListCap);
    int *list_rk = list_q + kSWaves * kSmallListCap;                         // (DET only; 8-byte aligned: the lists are multiples of 8 entries)
    int *novf_p = wsum + 8, *total_p = wsum + 9;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = tid & 7;
    MSDA_STAMP(0);
    const int NP = Lq * P;                                                   // <= kSingleMaxPoints (the host's plan)
    // ---- which rows are mine: from the header the forward left behind the point table (one 32-byte entry), or worked out here ----
    int l, H, Wd, lstart, px0, npx, cap;
    const int T = W_plan * L;
    // slot decode without an integer division: bid < 2^16 and T <= 254 on this path, so floor((bid + 0.5) / T) is exact in float
    const int pr = __builtin_amdgcn_readfirstlane((int)(((float)bid + 0.5f) * __builtin_amdgcn_rcpf((float)T)));
    const int slot = bid - pr * T;
    const int b = __builtin_amdgcn_readfirstlane((int)(((float)pr + 0.5f) * __builtin_amdgcn_rcpf((float)M))), m = pr - b * M;
    // (the header's magic is the forward's "table written" stamp: without it neither the entries nor the points are trusted)
    const bool stamped = header && header->magic == kRangeMagic;
    if (stamped && T <= kMaxRangeEntries && header->W == W_plan && header->L == L && header->tp_cap == tp_cap) {
        const RangeEntry e = reinterpret_cast<const RangeEntry *>(header + 1)[slot];
        l = e.l; H = e.H; Wd = e.Wd; lstart = e.lstart; px0 = e.px0; npx = e.npx; cap = e.cap;
        if (slot == 0 && !header->tiled) zero_uncovered_rows<GT, kSBlock>(shapes, level_start, S, M, L, grad_value, b, m);
        if (npx <= 0) return;
    } else {
        int pr2, ti, Wl;
        value_block_to_range(bid, W_plan, L, shapes, pr2, l, ti, Wl, true);
        H = (int)shapes[2 * l]; Wd = (int)shapes[2 * l + 1]; lstart = (int)level_start[l];
        const int HW = H * Wd;
        px0 = (int)((unsigned)(ti * HW) / (unsigned)Wl);
        npx = (int)((unsigned)((ti + 1) * HW) / (unsigned)Wl) - px0;
        if (l == 0 && ti == 0) zero_uncovered_rows<GT, kSBlock>(shapes, level_start, S, M, L, grad_value, b, m);
        const bool level_ok = level_fits(shapes[2 * l], shapes[2 * l + 1], level_start[l], S);
        if (npx <= 0 || npx > tp_cap || !level_ok) return;
        cap = kSmallRecCap / npx;
    }
    const long long item_base = (long long)b * Lq * M + m;                   // item(q) = item_base + q*M
    const int row_stride = M * kD;
    const GoBuf<VT> go = pair_rows<VT>(grad_out, item_base, Lq, M, j);
    GT *gv_base = grad_value + ((long long)(b * S + lstart + px0) * M + m) * kD + j * 4;
    const PointEntry *tab = (table && stamped) ? table + ((long long)pr * L + l) * NP : nullptr;       // (uniform)

    // ---- this pass's points: table entries, or entries made from sampling_loc / attn_weight ----
    constexpr int PPT = kSinglePPT;
    PointEntry pe[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int idx = tid + k * kSBlock;
        pe[k].cell = 0; pe[k].lh = pe[k].lw = pe[k].a = 0.f;
        if (idx < NP) {
            if (tab) pe[k] = tab[idx];
            else {
                const int q = fdiv(idx, P, p_shift), p = idx - q * P;
                const long long pi = ((item_base + (long long)q * M) * L + l) * P + p;
                const float2 xy = reinterpret_cast<const float2 *>(loc)[pi];
                pe[k] = point_entry(xy.x, xy.y, attn[pi], H, Wd, true);
            }
        }
    }
    for (int i = tid; i < npx * CW; i += kSBlock) cnt[i] = 0;
    if (tid == 0) { *novf_p = 0; *total_p = 0; }
    __syncthreads();
    MSDA_STAMP(1);

    if constexpr (DET) {
        if (cap >= 4) {                                                      // slots per row (uniform); below four: the general body
            PointEntry *mylist = list + wave * kSmallListCap;
            int *myq = list_q + wave * kSmallListCap;
            int2 *myrk = reinterpret_cast<int2 *>(list_rk) + wave * kSmallListCap;     // four 16-bit ranks per listed point
            // ---- list the points with a tap on my rows: ballot order (point slot k, then lane) — a pure function of the inputs ----
            int listed = 0;
            bool late[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                int dest[4];
                small_dests(pe[k], Wd, px0, npx, dest);
                const bool keep = (dest[0] & dest[1] & dest[2] & dest[3]) >= 0;
                const unsigned long long mask = __ballot(keep);
                const int pos = listed + __popcll(mask & ((1ull << lane) - 1ull));
                late[k] = keep && pos >= kSmallListCap;
                if (keep && pos < kSmallListCap) { mylist[pos] = pe[k]; myq[pos] = fdiv(tid + k * kSBlock, P, p_shift); }
                listed += __popcll(mask);
            }
            __builtin_amdgcn_wave_barrier();
            const int n_list = min(listed, kSmallListCap);
            // ranks among THIS wavefront's taps of a row: list order (trips in order, lanes of one atomic in lane order), then
            // the points beyond the list's capacity in slot order
            auto ranks_of = [&](const int (&dest)[4]) -> int2 {
                int r[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) r[t] = dest[t] >= 0 ? det_count(cnt, dest[t], wave) : 0;
                return make_int2(r[0] | (r[1] << 16), r[2] | (r[3] << 16));
            };
            for (int i0 = 0; i0 < n_list; i0 += kWave) {
                const int i = i0 + lane;
                PointEntry e; e.cell = 0; e.lh = e.lw = e.a = 0.f;
                if (i < n_list) e = mylist[i];
                int dest[4];
                small_dests(e, Wd, px0, npx, dest);
                const int2 rk = ranks_of(dest);
                if (i < n_list) myrk[i] = rk;
            }
            int2 late_rk[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                late_rk[k] = make_int2(0, 0);
                if (listed > kSmallListCap) {                                // (uniform per wavefront; rare)
                    int dest[4];
                    small_dests(pe[k], Wd, px0, npx, dest);
                    if (!late[k]) dest[0] = dest[1] = dest[2] = dest[3] = -1;
                    late_rk[k] = ranks_of(dest);
                }
            }
            __syncthreads();
            // ---- rows' totals; the packed counts become each wavefront's base inside its row ----
            for (int r = tid; r < npx; r += kSBlock) start[r] = det_row_prefix(cnt, r);
            __syncthreads();
            MSDA_STAMP(2); MSDA_STAMP(3);
            {
                // a record's place in its row is (wavefront base + rank): below the segment's capacity it goes there, beyond it
                // to the overflow list together with that place — the list is sorted by (row, place) below, so a row's overflow
                // records are added in a fixed order too
                auto place = [&](const PointEntry &e, int q, int2 rk, const int (&dest)[4]) {
                    float tw[4];
                    small_weights(e, tw);
                    const int rks[4] = {rk.x & 0xffff, (int)((unsigned)rk.x >> 16), rk.y & 0xffff, (int)((unsigned)rk.y >> 16)};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (dest[t] >= 0) {
                            const int pos = det_wave_base(cnt, dest[t], wave) + rks[t];
                            if (pos < cap) { SRec r; r.w = tw[t]; r.q = q; rec[__mul24(dest[t], cap) + pos] = r; }
                            else {
                                const int o = atomicAdd(novf_p, 1);
                                if (o < kOvfCap) { SOvf r; r.w = tw[t]; r.q = q; r.row = (dest[t] << 16) | pos; ovf[o] = r; }
                            }
                        }
                };
                int taps = 0;
                for (int i0 = 0; i0 < n_list; i0 += kWave) {
                    const int i = i0 + lane;
                    if (i < n_list) {
                        const PointEntry e = mylist[i];
                        int dest[4];
                        small_dests(e, Wd, px0, npx, dest);
                        place(e, myq[i], myrk[i], dest);
                    }
                }
                if (listed > kSmallListCap) {
#pragma unroll
                    for (int k = 0; k < PPT; ++k)
                        if (late[k]) { int dest[4]; small_dests(pe[k], Wd, px0, npx, dest); place(pe[k], fdiv(tid + k * kSBlock, P, p_shift), late_rk[k], dest); }
                }
                for (int r = tid; r < npx; r += kSBlock) taps += start[r];
                taps = wave_sum(taps);
                if (lane == 0) atomicAdd(total_p, taps);
                __syncthreads();
                MSDA_STAMP(4);
                const int novf = *novf_p;                                    // (uniform)
                if (novf <= kOvfCap) {
                    if (novf > 0) {                                          // rare, a handful of entries: rank sort by (row, place)
                        SOvf mine; mine.w = 0.f; mine.q = 0; mine.row = 0;
                        int rank = 0;
                        if (tid < novf) { mine = ovf[tid]; for (int jj = 0; jj < novf; ++jj) rank += (int)(ovf[jj].row < mine.row); }
                        __syncthreads();
                        if (tid < novf) { mine.row = (int)((unsigned)mine.row >> 16); ovf[rank] = mine; }
                        __syncthreads();
                    }
                    const int total = *total_p;
                    if (total <= 4 * npx)       gather_rows<1, kAccNone, VT, GT, MSDA_GATHER_NR1>(go, gv_base, start, start, rec, nullptr, npx, row_stride, true, cap, ovf, novf);
                    else if (total <= 8 * npx)  gather_rows<2, kAccNone, VT, GT, 2>(go, gv_base, start, start, rec, nullptr, npx, row_stride, true, cap, ovf, novf);
                    else if (total <= 16 * npx) gather_rows<4, kAccNone, VT, GT, 2>(go, gv_base, start, start, rec, nullptr, npx, row_stride, true, cap, ovf, novf);
                    else                        gather_rows<8, kAccNone, VT, GT, 2>(go, gv_base, start, start, rec, nullptr, npx, row_stride, true, cap, ovf, novf);
                    MSDA_STAMP(5);
                    return;
                }
            }
        }
        // ---- the overflow list filled up (taps piled on few pixels), or a level too fine for four slots per row: the general
        // deterministic single-pass body from the top (the launch's LDS covers both layouts) ----
        __syncthreads();
        {
            int pr2, l2, ti2, Wl2;
            value_block_to_range(bid, W_plan, L, shapes, pr2, l2, ti2, Wl2, true);
            bwd_value_body<kAccNone, kSinglePPT, VT, GT, false, true>(grad_out, shapes, level_start, loc, attn, S, M, L, Lq, P, p_shift, tp_cap,
                                                                      grad_value, ti2, Wl2, l2, pr2, smem);
        }
        return;
    }

    if (cap >= 4) {                                                          // slots per row (uniform); below four: the general body
        // one point's taps -> ranks from the histogram atomics -> records in their slots
        int mine_taps = 0;
        auto sort_point = [&](const PointEntry &e, int q, const int (&dest)[4]) {
            int rk[4]; float tw[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { rk[t] = 0; if (dest[t] >= 0) { rk[t] = MSDA_DIAG(0) ? 0 : atomicAdd(&cnt[dest[t]], 1); ++mine_taps; } }
            if (MSDA_DIAG(1)) return;
            small_weights(e, tw);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (dest[t] >= 0) {
                    if (rk[t] < cap) { SRec r; r.w = tw[t]; r.q = q; rec[__mul24(dest[t], cap) + rk[t]] = r; }
                    else {
                        const int o = atomicAdd(novf_p, 1);
                        if (o < kOvfCap) { SOvf r; r.w = tw[t]; r.q = q; r.row = dest[t]; ovf[o] = r; }
                    }
                }
        };
        // ---- list the points that have a tap on this workgroup's rows (wavefront-local: ballots, no atomics, no barrier) ----
        PointEntry *mylist = list + wave * kSmallListCap;
        int *myq = list_q + wave * kSmallListCap;
        int listed = 0;
        bool late[PPT];                                                      // kept points beyond the list's capacity: sparse path below
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            int dest[4];
            small_dests(pe[k], Wd, px0, npx, dest);
            const bool keep = (dest[0] & dest[1] & dest[2] & dest[3]) >= 0;     // some tap is mine
            const unsigned long long mask = __ballot(keep);
            const int pos = listed + __popcll(mask & ((1ull << lane) - 1ull));
            late[k] = keep && pos >= kSmallListCap;
            if (keep && pos < kSmallListCap) { mylist[pos] = pe[k]; myq[pos] = fdiv(tid + k * kSBlock, P, p_shift); }
            listed += __popcll(mask);
        }
        __builtin_amdgcn_wave_barrier();
        const int n_list = MSDA_DIAG(3) ? 0 : min(listed, kSmallListCap);
        for (int i0 = 0; i0 < n_list; i0 += kWave) {                         // dense lanes: ~1 trip (fine levels) .. 2 (coarse)
            const int i = i0 + lane;
            PointEntry e; e.cell = 0; e.lh = e.lw = e.a = 0.f;
            int q = 0;
            if (i < n_list) { e = mylist[i]; q = myq[i]; }
            int dest[4];
            small_dests(e, Wd, px0, npx, dest);
            sort_point(e, q, dest);
        }
        if (listed > kSmallListCap || MSDA_DIAG(3)) {                        // (uniform per wavefront; rare: more than half its points kept)
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                int dest[4];
                small_dests(pe[k], Wd, px0, npx, dest);
                if (!late[k] && !MSDA_DIAG(3)) dest[0] = dest[1] = dest[2] = dest[3] = -1;
                sort_point(pe[k], fdiv(tid + k * kSBlock, P, p_shift), dest);
            }
        }
        mine_taps = wave_sum(mine_taps);
        if (lane == 0) atomicAdd(total_p, mine_taps);
        __syncthreads();
        MSDA_STAMP(2); MSDA_STAMP(3); MSDA_STAMP(4);
        const int novf = *novf_p;
        if (novf <= kOvfCap) {
            if (MSDA_DIAG(4)) return;
            const int taps = *total_p;                       // lanes per row from the mean records per row (no division: 2*taps/npx <= k)
            if (taps <= 4 * npx)       gather_rows<1, kAccNone, VT, GT, MSDA_GATHER_NR1, GoBuf<VT>, MSDA_GATHER_CH1>(go, gv_base, cnt, start, rec, nullptr, npx, row_stride, true, cap, ovf, novf);
            else if (taps <= 8 * npx)  gather_rows<2, kAccNone, VT, GT, 2>(go, gv_base, cnt, start, rec, nullptr, npx, row_stride, true, cap, ovf, novf);
            else if (taps <= 16 * npx) gather_rows<4, kAccNone, VT, GT, 2>(go, gv_base, cnt, start, rec, nullptr, npx, row_stride, true, cap, ovf, novf);
            else                   gather_rows<8, kAccNone, VT, GT, 2>(go, gv_base, cnt, start, rec, nullptr, npx, row_stride, true, cap, ovf, novf);
            MSDA_STAMP(5);
            return;
        }
    }

    // ---- taps piled on few pixels (the overflow list filled up), or a level too fine for four slots per row: the general
    // single-pass body (prefix-sum sort over a 4*Lq*P record array; the launch's LDS covers both layouts) from the top ----
    __syncthreads();
    {
        int pr2, l2, ti2, Wl2;
        value_block_to_range(bid, W_plan, L, shapes, pr2, l2, ti2, Wl2, true);
        bwd_value_body<kAccNone, kSinglePPT, VT, GT, false, false>(grad_out, shapes, level_start, loc, attn, S, M, L, Lq, P, p_shift, tp_cap,
                                                                   grad_value, ti2, Wl2, l2, pr2, smem);
    }
}

// ---- kAccWide: single pass by KEPT taps ----------------------------------------------------------------------
// The chunked passes above size a pass by the points it SCANS (4 records per point), although a workgroup that
// owns 1/W of a level keeps ~1/W of them: cfg-2 encoder took 8 passes of 1536 points, each a chain of five
// barrier-separated phases that kept ~770 records (profiles/r02_notes.md §10).  Here a workgroup
//   1. scans ALL Lq*P points of its (batch, head, level) once: histogram of the taps that land on its rows
//      (LDS integer atomics, no return value) and a compact list of the points that have such a tap
//      (wavefront ballot + one LDS atomic per wavefront);
//   2. prefix-sums the histogram;
//   3. revisits only the listed points (~1/W of them), recomputes their taps and drops the records into the
//      rows' segments (the prefix array is the cursor: it ends up holding segment ENDS);
//   4. gathers as before and stores every row once.
// If the kept taps exceed the record array or the list (locations piled on this workgroup's rows), it starts
// over in chunks whose taps always fit, accumulating like kAccRmw.
constexpr int kWideMaxStep = 65528;                 // points per attempt (what a 16-bit list entry / relative query can address; 0xffff = no query)
#ifndef MSDA_WIDE_PAD8
#define MSDA_WIDE_PAD8 1
#endif
constexpr bool kWidePad8 = MSDA_WIDE_PAD8 != 0;      // rows padded to 8 records when the record array has room (gather_pad8)
constexpr int kWideLdsBudget = 80 * 1024;           // two workgroups per CU
// Capacities of a kAccWide workgroup, the same on the host (plan_value) and on the device: what is left of the LDS budget
// after the row arrays goes to the list (16-bit entries, one per listed point; at least the 8 KB the gathers use as
// scratch once the list is dead) and to the 6-byte records.
__host__ __device__ inline void wide_caps(int tp_cap, int NP, int &rec_cap, int &list_cap, bool det = false)
{
    const int fixed = (det ? 24 : 8) * tp_cap + 128;                          // cnt (DET: 4 words per row + row totals), start, wsum
    int lc = ((NP < kWideMaxStep ? NP : kWideMaxStep) + 7) & ~7;
    const int list_bytes = 2 * lc > 8192 ? 2 * lc : 8192;
    int rc = (kWideLdsBudget - fixed - list_bytes) / 6;
    if (lc > rc) { rc = (kWideLdsBudget - fixed) / 8; lc = rc; }              // a listed point has at least one record
    rec_cap = rc & ~7;
    list_cap = (lc < rec_cap ? lc : rec_cap) & ~7;
}
__host__ __device__ inline int wide_lds_bytes(int tp_cap, int NP, bool det = false)
{
    int rc, lc;
    wide_caps(tp_cap, NP, rc, lc, det);
    return (det ? 24 : 8) * tp_cap + 128 + (2 * lc > 8192 ? 2 * lc : 8192) + 6 * rc;
}

template <typename VT, typename GT, bool DET>
__device__ __forceinline__ void bwd_value_wide_body(
    const VT *__restrict__ grad_out, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc,
    const float *__restrict__ attn, int S, int M, int L, int Lq, int P, int p_shift, int tp_cap,
    GT *__restrict__ grad_value, int ti, int W, int l, int pr, unsigned char *smem, const MaskIn mi)
{
    // LDS: [cnt tp_cap (DET: 4 words per row)] [start tp_cap] [DET: row totals tp_cap] [wsum 32] [list: list_cap x u16, at least
    //      8 KB] [weights rec_cap x f32] [queries rec_cap x u16]
    int rec_cap, list_cap;
    wide_caps(tp_cap, Lq * P, rec_cap, list_cap, DET);
    constexpr int CW = DET ? 4 : 1;
    int *cnt = reinterpret_cast<int *>(smem);
    int *start = cnt + tp_cap * CW;
    int *tot = DET ? start + tp_cap : cnt;                   // records per row, as the gathers read them
    int *wsum = DET ? tot + tp_cap : start + tp_cap;
    // DET: every wavefront lists its points in its own eighth of the list (wsum[24 + w] entries) and is the one that revisits them
    const int seg = DET ? (list_cap / kSWaves) & ~1 : list_cap;
    int *kept_w = wsum + 24;
    uint16_t *list = reinterpret_cast<uint16_t *>(wsum + 32);             // wsum[16..23]: per-wavefront longest row
    float *rw = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(list) + (2 * list_cap > 8192 ? 2 * list_cap : 8192));
    uint16_t *rq = reinterpret_cast<uint16_t *>(rw + rec_cap);
    const int wide_chunk = min(rec_cap / 4, DET ? (list_cap / kSWaves) & ~1 : list_cap);      // points whose taps and list entries always fit
    int *kept_p = wsum + 8;                                  // wsum[0..7]: per-wavefront sums of the prefix scan

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    MSDA_STAMP(0);
    const int H = (int)shapes[2 * l], Wd = (int)shapes[2 * l + 1], lstart = (int)level_start[l];
    const int HW = H * Wd;
    const int px0 = (int)((unsigned)(ti * HW) / (unsigned)W), px1 = (int)((unsigned)((ti + 1) * HW) / (unsigned)W);
    const int npx = px1 - px0;
    const int b = pr / M, m = pr - b * M;
    if (l == 0 && ti == 0) zero_uncovered_rows<GT, kSBlock>(shapes, level_start, S, M, L, grad_value, b, m);
    if (npx <= 0 || npx > tp_cap || !level_fits(shapes[2 * l], shapes[2 * l + 1], level_start[l], S)) return;
    const int NP = Lq * P;
    const long long item_base = (long long)b * Lq * M + m;
    const int row_stride = M * kD;
    GT *gv_base = grad_value + ((long long)(b * S + lstart + px0) * M + m) * kD + (lane & 7) * 4;
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    // Everything below is written for INSTRUCTION COUNT: a CU issues about one wavefront instruction per clock
    // (4 SIMDs x 4 cycles per wave64 op; 32-bit integer multiplies take four times that), and the first version of
    // this scan — 64-bit index arithmetic and the full tap geometry for every point — spent 18 us on 12 240 points.
    // 32-bit indices throughout: d32_supported() bounds N*Lq*M*L*P*2 and N*S*M*D below 2^31 and S below 2^19.
    const int MLP = M * L * P;
    const unsigned off0 = (unsigned)((item_base * L + l) * P);                 // float2 index of (b, q = 0, m, l, p = 0)
    const float Hf = (float)H, Wf = (float)Wd, px0f = (float)px0, npxf = (float)npx;

    // range-local destinations of a point's four taps (-1: outside the map or not one of this workgroup's rows)
    auto taps_of = [&](const float2 xy, int (&dest)[4], PointGeom<float> &g) {
        g = point_geom<float>(xy.x, xy.y, H, Wd);
        const int pix = g.h0 * Wd + g.w0 - px0;
        const int p01 = pix + 1, p10 = pix + Wd, p11 = pix + Wd + 1;
        dest[0] = (g.ok00 && pix >= 0 && pix < npx) ? pix : -1;
        dest[1] = (g.ok01 && p01 >= 0 && p01 < npx) ? p01 : -1;
        dest[2] = (g.ok10 && p10 >= 0 && p10 < npx) ? p10 : -1;
        dest[3] = (g.ok11 && p11 >= 0 && p11 < npx) ? p11 : -1;
    };
    // float2 index of point idx = q*P + p, and q
    auto point_of = [&](int idx, int &q) -> unsigned {
        q = fdiv(idx, P, p_shift);
        return off0 + (unsigned)(q * MLP + (idx - q * P));
    };

    bool pad8 = false;                                       // layout of the record array for the current attempt (count_points)
    // steps 1-2 for the points [p0, p1); returns the number of record slots they take
    // The forward's per-point range masks (MaskHeader, msda_d32.hip), if the caller handed them over and they were written for
    // exactly this plan: then the first attempt (all points at once) finds its candidates there instead of in sampling_loc.
    bool have_masks = false;
    if constexpr (!DET) {
        if (mi.masks) {                                      // (uniform: kernel argument; the header comes through the scalar cache)
            const MaskHeader *h = mi.hdr;
            have_masks = h->magic == kMaskMagic && h->W == W && h->L == L && h->NP == Lq * P && pr < h->pairs && (Lq * P & 3) == 0;
        }
    }
    auto count_points = [&](int p0, int p1) -> int {
        for (int i = tid; i < npx * CW; i += kSBlock) cnt[i] = 0;
        if (tid == 0) *kept_p = 0;
        if (DET && tid < kSWaves) kept_w[tid] = 0;
        __syncthreads();
        const bool from_masks = !DET && have_masks && p0 == 0 && p1 == Lq * P;      // (uniform)
        if (from_masks) {
            // 1a'. candidates from the masks: Lq*P bytes of this (pair, level), four points per 32-bit load, fully coalesced;
            // a point is listed iff bit `ti` of its byte is set.  Same list, same order of work as the scan below, minus the
            // strided float2 loads and the float range test.
            const unsigned *mw = reinterpret_cast<const unsigned *>(mi.masks + ((size_t)pr * L + l) * (size_t)(Lq * P));
            const int nd = (Lq * P) >> 2;
            constexpr int UM = 4;                                             // loads in flight per thread: 8192 points per trip
            for (int base = 0; base < nd; base += kSBlock * UM) {
                unsigned wv[UM];
#pragma unroll
                for (int u = 0; u < UM; ++u) {
                    const int d = base + u * kSBlock + tid;
                    wv[u] = d < nd ? (mw[d] >> ti) & 0x01010101u : 0u;          // bit 8j: point 4d + j is a candidate
                }
                unsigned long long mask[UM][4];
                int n = 0;
#pragma unroll
                for (int u = 0; u < UM; ++u)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        mask[u][jj] = 0;
                        if (base + u * kSBlock >= nd) continue;                 // uniform: no point in this slot
                        mask[u][jj] = __ballot((wv[u] >> (8 * jj)) & 1u);
                        n += __popcll(mask[u][jj]);
                    }
                if (n == 0) continue;                                           // uniform
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(kept_p, n);
                wbase = __shfl(wbase, 0, kWave);
#pragma unroll
                for (int u = 0; u < UM; ++u)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int pos = wbase + __popcll(mask[u][jj] & ((1ull << lane) - 1ull));
                        if (((mask[u][jj] >> lane) & 1ull) && pos < seg) list[pos] = (uint16_t)(4 * (base + u * kSBlock + tid) + jj);
                        wbase += __popcll(mask[u][jj]);
                    }
            }
        } else {
        // 1a. the points that MAY have a tap on this workgroup's rows -> list.  No histogram yet (an LDS atomic costs
        // ~30 cycles per wavefront instruction however few lanes take part, and here ~1/W of them would); the test is
        // a superset — inside the map and one of the two tap pairs meets [0, npx) — in float arithmetic (exact: S < 2^24);
        // the exact taps are worked out for the listed points only.  The thread's point index advances by 512 per
        // step: (q, p) and the address follow by additions.
        constexpr int U = 9;                        // loads in flight per thread (4180 points = ONE trip); 10 crosses 128 VGPRs
        const int dq = fdiv(kSBlock, P, p_shift), dp = kSBlock - dq * P;        // 512 = dq*P + dp (uniform)
        const int doff = dq * MLP + dp, wrap = MLP - P;
        int q0;
        unsigned off = point_of(p0 + tid, q0);
        int p = p0 + tid - q0 * P;
        for (int base = p0; base < p1; base += kSBlock * U) {
            float2 xy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xy[u] = make_float2(-8.f, -8.f);                                // outside every map
                if (base + u * kSBlock + tid < p1) xy[u] = loc2[off];
                p += dp; off += doff;
                if (p >= P) { p -= P; off += wrap; }
            }
            unsigned long long mask[U];
            int n = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                mask[u] = 0;
                if (base + u * kSBlock >= p1) continue;                         // uniform: no point in this slot
                const float h_im = xy[u].y * Hf - 0.5f, w_im = xy[u].x * Wf - 0.5f;
                const float pixf = fmaf(floorf(h_im), Wf, floorf(w_im)) - px0f;  // tap (h0, w0), range-local
                // taps sit at pix, pix + 1 and pix + W, pix + W + 1 (border validity ignored: a superset)
                const bool keep = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf &&
                                  ((pixf + 1.f >= 0.f && pixf < npxf) || (pixf + Wf + 1.f >= 0.f && pixf + Wf < npxf));
                mask[u] = __ballot(keep);
                n += __popcll(mask[u]);
            }
            if (n == 0) continue;                                               // uniform
            int wbase = 0;
            if (DET) {                                                          // this wavefront's own cursor: no atomic, no race
                wbase = kept_w[wave];
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) { kept_w[wave] = wbase + n; atomicAdd(kept_p, n); }
            } else {
                if (lane == 0) wbase = atomicAdd(kept_p, n);
                wbase = __shfl(wbase, 0, kWave);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pos = wbase + __popcll(mask[u] & ((1ull << lane) - 1ull));
                if (((mask[u] >> lane) & 1ull) && pos < seg) list[(DET ? wave * seg : 0) + pos] = (uint16_t)(base - p0 + u * kSBlock + tid);
                wbase += __popcll(mask[u]);
            }
        }
        }                                                    // (scan)
        __syncthreads();
        MSDA_STAMP(1);
        if (MSDA_DIAG(6)) return 0;
        // 1b. histogram of the listed points' taps (dense lanes: ~4 atomics per 64 listed points)
        {
            // DET: the wavefront walks its own list segment (64 entries per step); else all threads share one list
            const int kept = DET ? min(kept_w[wave], seg) : min(*kept_p, list_cap);
            const uint16_t *mylist = list + (DET ? wave * seg : 0);
            const int me = DET ? lane : tid, stride = DET ? kWave : kSBlock;
            // up to 8 loads in flight per thread (a coarse level lists several thousand points); the slots past the
            // list's end are skipped as a whole (uniform test)
            constexpr int UH = 8;
            for (int base = 0; base < kept; base += stride * UH) {
                float2 xy[UH];
#pragma unroll
                for (int u = 0; u < UH; ++u) {
                    const int i = base + u * stride + me;
                    xy[u] = make_float2(-8.f, -8.f);
                    if (i < kept) { int q; xy[u] = loc2[point_of(p0 + mylist[i], q)]; }
                }
#pragma unroll
                for (int u = 0; u < UH; ++u) {
                    if (base + u * stride >= kept) break;
                    int dest[4]; PointGeom<float> g;
                    taps_of(xy[u], dest, g);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (dest[t] >= 0) { if (DET) det_count_only(cnt, dest[t], wave); else atomicAdd(&cnt[dest[t]], 1); }
                }
            }
        }
        __syncthreads();
        MSDA_STAMP(2);
        // exclusive prefix sum over the rows (512 threads x CH consecutive rows)
        const int CH = (npx + kSBlock - 1) / kSBlock;
        const int r0 = tid * CH;
        // One scan for two layouts of the record array: rows back to back (low 16 bits), or every row rounded up to a multiple
        // of 8 records (high 16 bits; totals stay below 2^14) — the padded layout if it fits: gather_pad8 then walks a row's
        // records 8 at a time without a boundary test per record (kWidePad8).
        auto padded = [](int c) { return c + (((c + 7) & ~7) << 16); };
        int mine = 0, big = 0;
        if (DET) {                                           // row totals; the packed counts become within-row prefixes (cursors)
            for (int k = 0; k < CH; ++k) if (r0 + k < npx) { const int t = det_row_prefix(cnt, r0 + k); tot[r0 + k] = t; mine += padded(t); big = max(big, t); }
        } else
        for (int k = 0; k < CH; ++k) if (r0 + k < npx) { mine += padded(cnt[r0 + k]); big = max(big, cnt[r0 + k]); }
        int incl = mine;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) { const int y = __shfl_up(incl, o, kWave); if (lane >= o) incl += y; }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) big = max(big, __shfl_xor(big, o, kWave));
        if (lane == kWave - 1) { wsum[wave] = incl; wsum[16 + wave] = big; }
        __syncthreads();
        int excl = incl - mine, total = 0;
        {
            const int4 wa = *reinterpret_cast<const int4 *>(wsum), wb = *reinterpret_cast<const int4 *>(wsum + 4);
            const int ws[kSWaves] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int w2 = 0; w2 < kSWaves; ++w2) { if (w2 < wave) excl += ws[w2]; total += ws[w2]; }
        }
        pad8 = kWidePad8 && (total >> 16) <= rec_cap;        // (uniform)
        const int sh = pad8 ? 16 : 0;
        for (int k = 0; k < CH; ++k) if (r0 + k < npx) { start[r0 + k] = (excl >> sh) & 0xffff; excl += padded(tot[r0 + k]); }
        __syncthreads();
        MSDA_STAMP(3);
        return (total >> sh) & 0xffff;
    };
    auto longest_row = [&]() {
        const int4 wa = *reinterpret_cast<const int4 *>(wsum + 16), wb = *reinterpret_cast<const int4 *>(wsum + 20);
        return max(max(max(wa.x, wa.y), max(wa.z, wa.w)), max(max(wb.x, wb.y), max(wb.z, wb.w)));
    };

    // steps 3-4 for the listed points.  A record = {weight, query relative to the chunk's first query (16 bits)}.
    auto scatter_and_gather = [&](int p0, int p1, int total, int first) {
        if (MSDA_DIAG(5) || MSDA_DIAG(6)) return;
        const int kept = DET ? kept_w[wave] : *kept_p;
        const uint16_t *mylist = list + (DET ? wave * seg : 0);
        const int me = DET ? lane : tid, stride = DET ? kWave : kSBlock;
        const int qbase = fdiv(p0, P, p_shift), q_count = fdiv(p1 - 1, P, p_shift) - qbase + 1;
        constexpr int U = 8;
        for (int base = 0; base < kept; base += stride * U) {
            float2 xy[U]; float at[U]; int qq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * stride + me;
                xy[u] = make_float2(-8.f, -8.f); at[u] = 0.f; qq[u] = 0;
                if (i < kept) {
                    const unsigned pi = point_of(p0 + mylist[i], qq[u]);
                    xy[u] = loc2[pi]; at[u] = attn[pi];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (base + u * stride >= kept) break;                            // uniform
                int dest[4]; PointGeom<float> g;
                taps_of(xy[u], dest, g);
                const float hh = 1.f - g.lh, hw = 1.f - g.lw;
                const float tw[4] = {hh * hw * at[u], hh * g.lw * at[u], g.lh * hw * at[u], g.lh * g.lw * at[u]};
                const uint16_t qrel = (uint16_t)(qq[u] - qbase);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (dest[t] >= 0) {
                        // DET: the cursor is this wavefront's 16-bit share of the row (start[] stays the row's beginning)
                        const int pos = DET ? start[dest[t]] + det_count(cnt, dest[t], wave) : atomicAdd(&start[dest[t]], 1);
                        rw[pos] = tw[t]; rq[pos] = qrel;
                    }
            }
        }
        __syncthreads();
        MSDA_STAMP(4);
        if (MSDA_DIAG(4)) return;
        if (DET || pad8) {
            // start[] becomes the rows' ENDS (the deterministic variant's stayed their beginnings, the other's cursor is there
            // already); padded layout: the slots up to the next multiple of 8 get weight 0 and a query past the descriptor
            for (int i = tid; i < npx; i += kSBlock) {
                int e = start[i] + (DET ? tot[i] : 0);
                if (pad8) {
                    const int fill = (-tot[i]) & 7;
                    for (int k = 0; k < fill; ++k) { rw[e + k] = 0.f; rq[e + k] = (uint16_t)0xffffu; }
                    e += fill;
                }
                start[i] = e;
            }
            __syncthreads();
        }
        // the records' queries are relative to qbase: so is the descriptor the gathers read grad_out through (one add less per record)
        // (... and it ends with the attempt's last query — at most kWideMaxStep of them — so that the padding slots' query
        // 0xffff lies past it and reads zeros)
        const GoBuf<VT> go = pair_rows<VT>(grad_out, item_base + (long long)qbase * M, min(Lq - qbase, q_count), M, lane & 7);
        // rows of comparable length (the longest no more than one lane group's share): the balanced walk
        if (longest_row() * 64 <= max(total, 1024)) {
            if (pad8) gather_pad8<VT, GT>(go, gv_base, start, RecSoa{rw, rq, 0, go.stride_b}, reinterpret_cast<int *>(list), npx, row_stride, total, first);
            else
            gather_balanced<VT, GT, true>(go, gv_base, tot, start, RecSoa{rw, rq, 0, go.stride_b}, reinterpret_cast<int *>(list), npx,
                                          row_stride, total, first);
            MSDA_STAMP(5);
            return;
        }
        // rows of very different lengths (coarse levels): equal stretches of records, parts combined by the row's owner.
        // (Measured against it: the split gather for every row mix — same at cfg-2 / cfg-4 encoder, but on the single-pass
        // path 784 six-record rows cost 19.5 instead of 14.5 us; gather_rows here: 20.2 instead of 11.3 us per workgroup.)
        gather_split<VT, GT, true>(go, gv_base, tot, start, RecSoa{rw, rq, 0, go.stride_b}, reinterpret_cast<float4 *>(list), npx,
                                   row_stride, total, first);
        MSDA_STAMP(5);
    };

    // All points at once; if this workgroup's rows receive more taps than the record array holds, a chunk sized
    // from the count just taken (7/8 full if the points spread evenly), halved while it still does not fit.
    // A chunk of wide_chunk points always fits (4 taps per point), so the loop ends.
    int a = 0, step = min(NP, kWideMaxStep);
    int first = 1;
    while (a < NP) {
        const int a1 = min(NP, a + step);
        __syncthreads();                                     // the previous attempt / gather still reads the LDS arrays
        const int t = count_points(a, a1);
        bool list_full = *kept_p > list_cap;
        if (DET) {
#pragma unroll
            for (int w2 = 0; w2 < kSWaves; ++w2) list_full = list_full || kept_w[w2] > seg;
        }
        if (t > rec_cap || list_full) {                      // uniform: LDS values read after a barrier
            const long long even = (long long)(a1 - a) * (rec_cap - rec_cap / 8) / max(t, 1);
            step = max(wide_chunk, (int)min(even, (long long)(a1 - a) / 2));
            continue;
        }
        scatter_and_gather(a, a1, t, first);
        first = 0;
        a = a1;
    }
}


// Role-B workgroup id -> (pair, level, range, ranges of that level).  A (batch, head) pair has W*L workgroups.
// They are dealt W per level, except that on a pyramid (the level with the most pixels has >= 4x the pixels
// of the one with the fewest; W >= 2) the largest level takes one range from the smallest: every level
// receives the same number of taps, but a fine level's gather also walks 4-64x more rows, and the launch ends
// with its slowest workgroup (cfg-2 decoder: level 0 gather 5.9 us vs 3.6 us for level 3 with W = 4 each).
// Uniform (scalar) arithmetic; the shapes come from the scalar cache.
__device__ __forceinline__ void value_block_to_range(int bid, int W, int L, const int64_t *__restrict__ shapes,
                                                     int &pr, int &l, int &ti, int &Wl, bool may_skew = true)
{
    const int T = W * L;
    pr = bid / T;
    const int s = bid - pr * T;
    int lmax = 0, lmin = 0;
    long long rmax = shapes[0] * shapes[1], rmin = rmax;
    for (int k = 1; k < L; ++k) {
        const long long r = shapes[2 * k] * shapes[2 * k + 1];
        if (r > rmax) { rmax = r; lmax = k; }
        if (r <= rmin) { rmin = r; lmin = k; }
    }
    // (kAccWide deals W to every level: there the RECORDS a workgroup keeps are what must stay equal)
    const bool skew = may_skew && L >= 2 && W >= 2 && lmax != lmin && rmax >= 4 * rmin;
    int base = 0;
    l = 0; ti = 0; Wl = W;
    for (int k = 0; k < L; ++k) {
        const int wk = W + (skew ? (int)(k == lmax) - (int)(k == lmin) : 0);
        if (s < base + wk) { l = k; ti = s - base; Wl = wk; return; }
        base += wk;
    }
}

}  // namespace msda
