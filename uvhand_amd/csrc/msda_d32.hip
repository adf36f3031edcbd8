// D = 32 multi-scale deformable attention for MI355X (gfx950) — the shape the UVHand transformers
// use (d_model 256 / 8 heads; util/settings.py:102-120): fp32 or bf16 rows, fp32 arithmetic; forward,
// backward (role A + role B in one launch), and the fused-prologue variants of both.
//
// Work decomposition (not the reference's one-thread-per-channel / 32-thread blocks,
// ms_deform_im2col_cuda.cuh:237-299, :301-403):
//
//   * item  = one (batch, query, head); its output row is 32 floats = 128 B.
//   * octet = 8 consecutive items = 64 lanes: 8 lanes per item, one float4 (16 B) per
//     lane, so every tap is one 128-B line per item and a wavefront's store of its
//     octet is one contiguous 1-KiB write (for M = 8 an octet is exactly one query).
//   * a 256-thread workgroup first turns the sampling locations of its items into
//     "tap records" in LDS — ONE lane per sampling point does the floor / bounds /
//     bilinear-weight arithmetic once, reading loc and attn coalesced from HBM —
//     and then the gather lanes only read records (broadcast ds_read_b128) and rows.
//     The reference recomputes that arithmetic in every one of the 32 channel threads.
//   * SPLIT = 4: the 4 wavefronts of a workgroup share ONE octet and split its L*P
//     points, so that small problems (the 300-query decoder shape has only 600
//     octets) still put ~2400 wavefronts with 16 row loads each in flight;
//     SPLIT = 1: one octet per wavefront, no cross-wave reduction (encoder regime,
//     large batches).
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "msda_common.h"
#include "msda_launch.h"

// This file is compiled three times by the Makefile, one object per storage-type combination, so that the three sets of
// kernel instantiations build side by side (-DMSDA_D32_PART=0: fp32 rows, 1: bf16 rows with a bf16 grad_value, 2: bf16
// rows with an fp32 grad_value); without the macro (tools/micro/kbench.cpp's unity build) it holds all of them.
#ifndef MSDA_D32_PART
#define MSDA_D32_PART -1
#endif
#define MSDA_D32_HAS(part) (MSDA_D32_PART < 0 || MSDA_D32_PART == (part))

namespace msda {

// Diagnostic build only (tools/micro/kbench.cpp, -DMSDA_STAMPS): per-workgroup phase timestamps
// (100 MHz s_memrealtime) into a side buffer that no kernel reads.  The shipped library is built
// without it and executes no stamp.
#ifdef MSDA_STAMPS
__device__ unsigned long long *msda_stamp_buf = nullptr;              // [blocks][8]
#define MSDA_STAMP_AT(region, i)                                                                   \
    do { if (threadIdx.x == 0 && msda_stamp_buf)                                                   \
             msda_stamp_buf[((size_t)(region) * 65536 + (blockIdx.x & 65535)) * 8 + (i)] =         \
                 __builtin_amdgcn_s_memrealtime(); } while (0)
#define MSDA_STAMP(i) MSDA_STAMP_AT(0, i)
__device__ int msda_skip_role = 0;                                    // diagnostic: 1 = role B's workgroups exit at once, 2 = role A's
#define MSDA_SKIP_ROLE(is_b) do { if (msda_skip_role == ((is_b) ? 1 : 2)) return; } while (0)
__device__ int msda_diag = 0;                                         // diagnostic bit flags (timing only: results are then wrong)
#define MSDA_DIAG(bit) ((msda_diag >> (bit)) & 1)
#else
#define MSDA_SKIP_ROLE(is_b) do { } while (0)
#define MSDA_DIAG(bit) 0
#define MSDA_STAMP(i) do { } while (0)
#define MSDA_STAMP_AT(region, i) do { } while (0)
#endif

constexpr int kD = 32;
constexpr int kBlock = 256;
constexpr int kRecBytes = 32;      // one tap record
constexpr int kItemPad = 16;       // bytes: shifts consecutive items by one 16-B bank slot so the
                                   // 8 per-item broadcast reads of a wavefront do not collide

struct alignas(16) LevelInfo { int H, W, start, pad; };

// forward record: byte offsets of the 4 taps' rows inside the workgroup's window of `value` (kBufOob = tap outside
// the map: the buffer load returns zeros) and their weights already multiplied by the attention weight.
struct alignas(16) FwdRec { int off[4]; float w[4]; };
// backward record: same offsets, the two fractions and the attention weight.
struct alignas(16) BwdRec { int off[4]; float lh, lw, a, pad; };

// One sampling point as the FORWARD can leave it for the backward of the same autograd node (msda_forward_ws_*), level-major:
// entry ((b*M + m)*L + l) * Lq*P + q*P + p.  Role B of a small problem reads its level's entries coalesced instead of
// re-deriving them from a strided scan of sampling_loc / attn_weight (msda_d32_value.h: bwd_value_small_body).
struct alignas(16) PointEntry { int cell; float lh, lw, a; };   // cell: tap validity bits << 24 | (h0 * W + w0 + W + 1); 0 = no tap
// Behind the point entries the table carries a small HEADER, written by one workgroup of the forward: for every role-B
// workgroup slot s of a (batch, head) pair (s = level-and-range, value_block_to_range) the level geometry and pixel range that
// workgroup would otherwise derive from spatial_shapes / level_start_index itself — ~550 scalar instructions per wavefront
// (64-bit products, six integer divisions, the level walk) at the head of every role-B wavefront's life, a serial chain of
// more than a microsecond (SQ counters with role B alone and its phases switched off, profiles/r04_notes.md).
struct alignas(32) RangeEntry { int l, H, Wd, lstart, px0, npx, cap, pad; };   // npx = 0: nothing to do (the level does not fit, or an empty range)
constexpr int kMaxRangeEntries = 254;                                          // W * L of a small problem's plan (else: no header)
struct alignas(32) RangeHeader { int magic, W, L, tiled, tp_cap, pad[3]; };    // entries follow
constexpr int kRangeMagic = 0x4d534441;
__host__ __device__ inline size_t range_header_bytes() { return sizeof(RangeHeader) + (size_t)kMaxRangeEntries * sizeof(RangeEntry); }

// ---- LARGE problems (LDS-stage forward, kept-taps role B): per-point RANGE MASKS ------------------------------------------------
// Role B of a multi-pass problem cuts every level into W pixel ranges, and each range's workgroup used to re-scan ALL Lq*P
// sampling points of its (batch, head, level) — a strided float2 read and a float range test per point — for the ~1/W that have
// a tap on its rows (cfg-2 encoder: 6 x 12 240 points per level and pair, 12 us of a 53 us workgroup; profiles/r04_notes.md
// section 4).  The FORWARD of the same autograd node visits every point anyway (fwd_d32_lds_kernel: one lane per point):
// with a mask buffer it also stores ONE BYTE per point, level-major — bit t set iff one of the point's taps may land in range
// t of its level — and role B's scan becomes a coalesced read of Lq*P bytes and a bit test.  The masks are a SUPERSET (all
// ranges from the first tap pixel's to the last one's, border validity ignored): role B works out the exact taps of the
// points it keeps as it does after its own scan, so results do not depend on where the candidates came from.
// (First version, measured and replaced: per-range LISTS of 16-bit point indices appended through LDS cursors — role B's
// scan gone just the same, but the scattered 2-byte stores cost the forward 4-10 us, all the backward gained.)
//   layout: [MaskHeader 64 B] [masks: u8 [pairs][L][Lq*P]]        (W <= 8, Lq*P a multiple of 4)
struct alignas(64) MaskHeader { int magic, W, L, NP, pairs, pad[11]; };
constexpr int kMaskMagic = 0x4d53444d;
constexpr int kMaxMaskRanges = 8;                                               // W: one bit per range
constexpr int kMinMaskRanges = 4;                                               // fewer ranges: the scan is cheaper than the forward's bytes (plan_masks)
struct MaskOut { MaskHeader *hdr; uint8_t *masks; int W; };                    // forward side (masks = null: none)
struct MaskIn { const MaskHeader *hdr; const uint8_t *masks; };                // role B side

// The table entry of a point from its geometry (the same values the tap records are made of, so that the table and a scan
// of sampling_loc / attn_weight give role B bit-identical records).
__device__ __forceinline__ PointEntry entry_of(const PointGeom<float> &g, float a, int Wd, bool level_ok)
{
    PointEntry e;
    const int okb = (int)g.ok00 | ((int)g.ok01 << 1) | ((int)g.ok10 << 2) | ((int)g.ok11 << 3);
    e.cell = (g.inside && level_ok) ? ((okb << 24) | (g.h0 * Wd + g.w0 + Wd + 1)) : 0;
    e.lh = g.lh; e.lw = g.lw; e.a = a;
    return e;
}
__device__ __forceinline__ PointEntry point_entry(float x, float y, float a, int H, int Wd, bool level_ok)
{
    return entry_of(point_geom<float>(x, y, H, Wd), a, Wd, level_ok);
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Storage type of the value-like tensors (value, out, grad_out, grad_value): float, or bfloat16
// bits (uint16_t) with all arithmetic and accumulation in fp32 and ONE rounding at the final store.
using bf16_t = uint16_t;

// A lane's 4 consecutive channels of a row.
template <typename VT> struct Row;
template <> struct Row<float> {
    static __device__ __forceinline__ float4 load(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void store(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
    static __device__ __forceinline__ void store1(float *p, float v) { *p = v; }
};
template <> struct Row<bf16_t> {
    static __device__ __forceinline__ float4 load(const bf16_t *p)
    {
        const uint2 u = *reinterpret_cast<const uint2 *>(p);                 // 4 x bf16 = 8 B per lane
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi)
    {
        // plain casts: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
        const __bf16 a = static_cast<__bf16>(lo), b = static_cast<__bf16>(hi);
        return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
    }
    static __device__ __forceinline__ void store(bf16_t *p, const float4 &v)
    {
        *reinterpret_cast<uint2 *>(p) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
    }
    static __device__ __forceinline__ void store1(bf16_t *p, float v)
    {
        *p = __builtin_bit_cast(unsigned short, static_cast<__bf16>(v));
    }
};

// Row loads of `value` go through a BUFFER descriptor with a 32-bit byte offset: no 64-bit address pair per load, and a tap
// that is absent (outside the map) carries an offset past the descriptor, for which the hardware returns zeros without
// touching memory — so a gather lane spends ONE add per tap (its 16-byte column) and no compare / clamp / select, all sixteen
// loads of a trip are in flight together, and an Inf / NaN in an unsampled row can never leak in.  (The first tiled kernels
// loaded through 64-bit pointers with the result discarded by four selects per tap: ~10 vector instructions per tap of a
// kernel that the SQ counters show to be bound by instruction issue — profiles/r04_notes.md.)
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
constexpr int kBufDword3 = 0x00020000;             // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
constexpr unsigned kBufOob = 0x80000000u;          // >= any descriptor size used here (slices are checked < 2^31 bytes)

template <typename VT> struct BufRow;
template <> struct BufRow<float> {
    static __device__ __forceinline__ float4 load(__amdgpu_buffer_rsrc_t rs, unsigned off)
    {
        const v4u_t u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
    }
};
template <> struct BufRow<bf16_t> {
    static __device__ __forceinline__ float4 load(__amdgpu_buffer_rsrc_t rs, unsigned off)
    {
        const v2u_t u = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
};

// The tiled kernels' descriptor: the batch elements [b0, b1] a workgroup's items belong to (b0 = the first item's; a
// workgroup rarely straddles a boundary).  d32_supported() admits only geometries whose widest such window stays below 2^31
// bytes, so record offsets (relative to batch element b0) are plain 32-bit numbers.
// A descriptor lives in scalar registers.  Its base and size here are uniform by construction (functions of the workgroup id
// and kernel arguments) but reach the compiler through integer divisions and 64-bit multiplies it evaluates on the vector
// unit — and a descriptor it cannot PROVE uniform makes every buffer load a "waterfall" loop (v_readfirstlane x4, v_cmp_eq_u64 x2,
// s_and_saveexec, the load, s_cbranch: twelve instructions around each of sixteen loads — found in the ISA of round 4's first
// buffer-load kernels, where it ate the whole gain).  uniform_rsrc() says so explicitly.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *base, long long bytes)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int n = __builtin_amdgcn_readfirstlane((int)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, n, kBufDword3);
}

template <typename VT>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t value_window(const VT *value, int b0, int b1, int S, int M)
{
    const long long slice = (long long)S * M * kD;                         // elements per batch element
    return uniform_rsrc(value + (long long)b0 * slice, (long long)(b1 - b0 + 1) * slice * (long long)sizeof(VT));
}

// byte offsets of a point's four tap rows inside that window (db = the item's batch element minus b0), kBufOob = absent
template <typename VT>
__device__ __forceinline__ void tap_offsets_b(const PointGeom<float> &g, const LevelInfo &lv, int db, int m, int S, int M, unsigned off[4])
{
    constexpr int RB = kD * (int)sizeof(VT);                               // bytes per row
    const int row = M * RB;                                                // next pixel
    const int base = ((db * S + lv.start) * M + m) * RB + (g.h0 * lv.W + g.w0) * row;
    off[0] = g.ok00 ? (unsigned)base : kBufOob;
    off[1] = g.ok01 ? (unsigned)(base + row) : kBufOob;
    off[2] = g.ok10 ? (unsigned)(base + lv.W * row) : kBufOob;
    off[3] = g.ok11 ? (unsigned)(base + lv.W * row + row) : kBufOob;
}
__device__ __forceinline__ void fma4(float4 &acc, float w, const float4 &v)
{
    acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y);
    acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
}
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b)
{
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// The three per-tap-set sums of role A from the four <grad_out, tap row> dot products of a lane's channels
// (ms_deform_im2col_cuda.cuh:116-158: grad_attn, and the pixel-space location gradients before the W / H scaling).
// Explicit fmaf chains: every role-A kernel (this file and msda_d32_lds.h) gets the same contraction, bit for bit.
__device__ __forceinline__ void tap_sums(float lh, float lw, float a, float d1, float d2, float d3, float d4, float &s_a,
                                         float &s_x, float &s_y)
{
    const float hh = 1.f - lh, hw = 1.f - lw;
    s_a = fmaf(lh * lw, d4, fmaf(lh * hw, d3, fmaf(hh * lw, d2, (hh * hw) * d1)));
    s_x = a * fmaf(lh, d4 - d3, hh * (d2 - d1));
    s_y = a * fmaf(lw, d4 - d2, hw * (d3 - d1));
}

// Sum over the 8 lanes that share an item (lanes 8k..8k+7), on the VALU's DPP path — no LDS
// crossbar round trips: quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror (lane j of
// each 8-lane half-row reads lane 7-j, which by then holds the other quad's sum).
template <int CTRL>
__device__ __forceinline__ float dpp_read(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float octlane_sum(float x)
{
    x += dpp_read<0xB1>(x);
    x += dpp_read<0x4E>(x);
    x += dpp_read<0x141>(x);
    return x;
}

// Workgroup ids are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  xcd_block() renumbers
// them so that every XCD works on ONE contiguous range of the logical blocks (bijection for any count).
__device__ __forceinline__ int xcd_block(int bid, int nb)
{
    const int x = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
    return x * q + (x < r ? x : r) + idx;
}

// n / d for a divisor that is usually a power of two (shift >= 0), exact otherwise.
__device__ __forceinline__ int fdiv(int n, int d, int shift) { return shift >= 0 ? (n >> shift) : n / d; }

// (batch, head) of the il-th item after item0, from the uniform decomposition of item0:
// b0 = item0 / (Lq*M), r0 = item0 % (Lq*M), m0 = item0 % M.
__device__ __forceinline__ void item_bm(int il, int b0, int r0, int m0, int LqM, int M, int m_shift, int &b, int &m)
{
    const int t = r0 + il;
    b = b0 + (t >= LqM ? t / LqM : 0);                     // a workgroup rarely straddles a batch boundary
    m = m_shift >= 0 ? ((m0 + il) & (M - 1)) : (m0 + il) % M;
}

// ---- fused prologue: the module's softmax over the L*P logits of an item and its
// location arithmetic  loc = reference_point + offset / (W_l, H_l)  (models/ops/modules/
// ms_deform_attn.py:101-108) done by the point lanes of the prepass instead of 3 elementwise launches
// over the [N,Lq,M,L,P,*] tensors.  The L*P points of an item sit on L*P consecutive lanes (L*P a power
// of two <= 64), so max and sum are xor-shuffles inside that lane group.
// The raw offsets / logits (and their gradients) may be column blocks of a wider row-major matrix — the
// output of ONE projection GEMM for both (f1) — so each carries the distance between consecutive query
// rows beyond its own M*L*P points: off_pad in float2 units, log_pad in floats (0 = dense tensors).
struct PrologueIn  { const float *ref; float *loc_out; float *attn_out; int off_pad, log_pad; };   // ref[N,Lq,L,2]
struct PrologueOut { float *grad_ref; int off_pad, log_pad; };                                       // grad_ref[N,Lq,L,2]

__device__ __forceinline__ float group_max(float x, int width)
{
    for (int m = 1; m < width; m <<= 1) x = fmaxf(x, __shfl_xor(x, m, kWave));
    return x;
}
__device__ __forceinline__ float group_sum(float x, int width)
{
    for (int m = 1; m < width; m <<= 1) x += __shfl_xor(x, m, kWave);
    return x;
}
// (defined in msda_d32_value.h, role B's workgroup id -> level / range dealing)
__device__ __forceinline__ void value_block_to_range(int bid, int W, int L, const int64_t *__restrict__ shapes, int &pr, int &l, int &ti,
                                                     int &Wl, bool may_skew);

// The header behind the point table: thread s < W*L works out slot s (one thread per slot: the level walk runs once per
// launch, in parallel, instead of once per role-B wavefront).
__device__ __forceinline__ void write_range_header(PointEntry *table, long long points, const int64_t *__restrict__ shapes,
                                                   const int64_t *__restrict__ level_start, int S, int L, int W, int tp_cap, int rec_cap)
{
    RangeHeader *hdr = reinterpret_cast<RangeHeader *>(table + points);
    RangeEntry *ent = reinterpret_cast<RangeEntry *>(hdr + 1);
    const int s = threadIdx.x, T = W * L;
    if (s == 0) {
        long long run = 0;
        bool tiled = true;                                   // every level fits and starts where the previous one ends, up to S
        for (int k = 0; k < L; ++k) {
            tiled = tiled && level_start[k] == run && level_fits(shapes[2 * k], shapes[2 * k + 1], level_start[k], S);
            run += shapes[2 * k] * shapes[2 * k + 1];
        }
        RangeHeader h;
        h.magic = kRangeMagic; h.W = W; h.L = L; h.tiled = (tiled && run == S) ? 1 : 0; h.tp_cap = tp_cap;
        h.pad[0] = h.pad[1] = h.pad[2] = 0;
        *hdr = h;
    }
    if (s < T && T <= kMaxRangeEntries) {
        int pr, l, ti, Wl;
        value_block_to_range(s, W, L, shapes, pr, l, ti, Wl, true);
        RangeEntry e;
        e.l = l; e.H = e.Wd = e.lstart = e.px0 = e.npx = e.cap = e.pad = 0;
        if (level_fits(shapes[2 * l], shapes[2 * l + 1], level_start[l], S)) {
            const int H = (int)shapes[2 * l], Wd = (int)shapes[2 * l + 1], HW = H * Wd;
            const int px0 = (int)((unsigned)(ti * HW) / (unsigned)Wl), px1 = (int)((unsigned)((ti + 1) * HW) / (unsigned)Wl);
            e.H = H; e.Wd = Wd; e.lstart = (int)level_start[l]; e.px0 = px0;
            e.npx = (px1 - px0 > 0 && px1 - px0 <= tp_cap) ? px1 - px0 : 0;
            e.cap = e.npx > 0 ? rec_cap / e.npx : 0;
        }
        ent[s] = e;
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// FUSED: `loc` holds the raw sampling offsets and `attn` the attention logits; see PrologueIn.
template <int SPLIT, typename VT, bool FUSED = false>
__global__ __launch_bounds__(kBlock) void fwd_d32_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc,
    const float *__restrict__ attn, int S, int M, int L, int Lq, int P, int items, int p_shift,
    int lp_shift, int m_shift, VT *__restrict__ out, const PrologueIn pro = PrologueIn{nullptr, nullptr, nullptr, 0, 0},
    int xcd = 0, PointEntry *__restrict__ table = nullptr, int hdr_W = 0, int hdr_tp_cap = 0, int hdr_rec_cap = 0)
{
    constexpr int IPW = 32 / SPLIT;                       // items per workgroup
    // with a header to write the launch has ONE EXTRA workgroup that does only that: beside the others (it is done long before
    // they are), not in front of one of them — as a prefix of workgroup 0 it made that workgroup, and with it the launch, 0.9 us longer
    const int n_blocks = (int)gridDim.x - (hdr_W > 0 ? 1 : 0);
    if (hdr_W > 0 && (int)blockIdx.x == n_blocks) {
        write_range_header(table, (long long)items * L * P, shapes, level_start, S, L, hdr_W, hdr_tp_cap, hdr_rec_cap);
        return;
    }
    constexpr int OPW = 4 / SPLIT;                        // octets per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *recs = smem;
    const int LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    const int tid = threadIdx.x;
    const int item0 = (xcd ? xcd_block((int)blockIdx.x, n_blocks) : (int)blockIdx.x) * IPW;
    MSDA_STAMP_AT(2, 0);
    const int LqM = Lq * M;
    const int b0 = item0 / LqM, r0 = item0 - b0 * LqM, m0 = item0 % M;      // uniform (scalar unit)

    // ---- one lane per sampling point: loc/attn coalesced from HBM -> tap records in LDS.
    // The level's (H, W, start) come straight from global memory (a handful of addresses, served by
    // the L1/scalar caches) in the same batch of loads as loc/attn: no table stage, no extra barrier.
    for (int idx = tid; idx < IPW * LP; idx += kBlock) {
        const int il = fdiv(idx, LP, lp_shift), pt = idx - il * LP;
        const int item = item0 + il;
        FwdRec r;
        r.off[0] = r.off[1] = r.off[2] = r.off[3] = (int)kBufOob;
        r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0.f;
        if (item < items) {
            const int l = fdiv(pt, P, p_shift);
            const long long e = (long long)item0 * LP + idx;
            const long long row = FUSED ? fdiv(item, M, m_shift) : 0;           // b*Lq + q
            float2 xy = reinterpret_cast<const float2 *>(loc)[FUSED ? e + row * pro.off_pad : e];
            float a = attn[FUSED ? e + row * pro.log_pad : e];
            LevelInfo li;
            li.H = (int)shapes[2 * l]; li.W = (int)shapes[2 * l + 1]; li.start = (int)level_start[l]; li.pad = 0;
            int b, m;
            item_bm(il, b0, r0, m0, LqM, M, m_shift, b, m);
            if (FUSED) {
                const float2 rp = reinterpret_cast<const float2 *>(pro.ref)[row * L + l];
                xy = make_float2(rp.x + xy.x / (float)li.W, rp.y + xy.y / (float)li.H);
                const float ex = expf(a - group_max(a, LP));
                a = ex / group_sum(ex, LP);
                reinterpret_cast<float2 *>(pro.loc_out)[e] = xy;
                pro.attn_out[e] = a;
            }
            const PointGeom<float> g = point_geom<float>(xy.x, xy.y, li.H, li.W);
            const bool level_ok = level_fits(shapes[2 * l], shapes[2 * l + 1], level_start[l], S);
            if (g.inside && level_ok) {
                tap_offsets_b<VT>(g, li, b - b0, m, S, M, reinterpret_cast<unsigned *>(r.off));
                const float hh = 1.f - g.lh, hw = 1.f - g.lw;
                r.w[0] = hh * hw * a; r.w[1] = hh * g.lw * a;
                r.w[2] = g.lh * hw * a; r.w[3] = g.lh * g.lw * a;
            }
            if (table) {                                     // (uniform) the point, level-major, for the backward of this node
                const int q = fdiv(r0 + il - (b - b0) * LqM, M, m_shift);
                table[(((long long)b * M + m) * L + l) * ((long long)Lq * P) + q * P + (pt - l * P)] = entry_of(g, a, li.W, level_ok);
            }
        }
        *reinterpret_cast<FwdRec *>(recs + il * item_stride + pt * kRecBytes) = r;
    }
    __syncthreads();
    MSDA_STAMP_AT(2, 1);

    // ---- gather: 8 lanes x float4 per item, 8 items per wavefront ----
    const int wave = tid >> 6, lane = tid & 63, grp = lane >> 3, j = lane & 7;
    const int il = (wave / SPLIT) * 8 + grp;
    const unsigned char *rb = recs + il * item_stride;
    const __amdgpu_buffer_rsrc_t vbuf = value_window<VT>(value, b0, (min(item0 + IPW, items) - 1) / LqM, S, M);
    const unsigned joff = (unsigned)(j * 4 * sizeof(VT));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // 4 points per trip: all records, then all 16 row loads, then the FMAs (one memory round trip
    // per trip instead of one per point).
    for (int p0 = wave % SPLIT; p0 < LP; p0 += 4 * SPLIT) {
        uint4 off[4]; float4 w[4]; float4 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * SPLIT;
            off[u] = make_uint4(kBufOob, kBufOob, kBufOob, kBufOob); w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < LP) {
                off[u] = *reinterpret_cast<const uint4 *>(rb + p * kRecBytes);
                w[u] = *reinterpret_cast<const float4 *>(rb + p * kRecBytes + 16);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u][0] = BufRow<VT>::load(vbuf, off[u].x + joff); v[u][1] = BufRow<VT>::load(vbuf, off[u].y + joff);
            v[u][2] = BufRow<VT>::load(vbuf, off[u].z + joff); v[u][3] = BufRow<VT>::load(vbuf, off[u].w + joff);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            fma4(acc, w[u].x, v[u][0]); fma4(acc, w[u].y, v[u][1]);
            fma4(acc, w[u].z, v[u][2]); fma4(acc, w[u].w, v[u][3]);
        }
    }
    MSDA_STAMP_AT(2, 2);

    if (SPLIT == 1) {
        if (item0 + il < items)
            Row<VT>::store(out + (long long)(item0 + il) * kD + j * 4, acc);
    } else {
        // the SPLIT wavefronts of an octet hold partial sums: combine through LDS in a fixed
        // order, then coalesced 1-KiB stores.
        float *red = reinterpret_cast<float *>(recs + IPW * item_stride);
        reinterpret_cast<float4 *>(red)[wave * 64 + lane] = acc;
        __syncthreads();
        for (int o = tid; o < OPW * 256; o += kBlock) {
            const int oct = o >> 8, f = o & 255;
            float sum = red[(oct * SPLIT) * 256 + f];
#pragma unroll
            for (int k = 1; k < SPLIT; ++k) sum += red[(oct * SPLIT + k) * 256 + f];
            if (item0 + (o >> 5) < items) Row<VT>::store1(out + (long long)item0 * kD + o, sum);
        }
    }
    MSDA_STAMP_AT(2, 3);
}

// ------------------------------------------------------------------------------------------
// backward, role A — query-major: grad_sampling_loc and grad_attn_weight.
// Same tiling as the forward; each lane forms the four tap dot products <grad_out, v_k> over its
// 4 channels, the three per-point sums are reduced over the item's 8 lanes and staged in LDS so
// that the workgroup writes both gradient tensors as contiguous runs.
// ATOMIC = true additionally scatters grad_value with global float atomics (the v1 scheme, kept
// for A/B measurements: MSDA_BWD_MODE=atomic); the default leaves grad_value to role B.
// ------------------------------------------------------------------------------------------
// FUSED: the write-out applies the chain rule of the fused prologue — grad_loc becomes the gradient of
// the raw offsets (grad_loc / (W,H)), grad_attn the gradient of the logits (softmax backward over the
// item's L*P lanes), and the location gradients summed over heads and points go to pro.grad_ref.
// Needs L*P and P powers of two and whole queries per workgroup (M | IPW): the host checks.
template <int SPLIT, bool ATOMIC, int THREADS, typename VT, bool FUSED = false>
__device__ __forceinline__ void bwd_query_body(
    const VT *__restrict__ grad_out, const VT *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L, int Lq,
    int P, int items, int p_shift, int lp_shift, int m_shift, VT *__restrict__ grad_value,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn, int block, unsigned char *smem,
    const PrologueOut pro = PrologueOut{nullptr, 0, 0})
{
    constexpr int IPW = (THREADS / kWave) * 8 / SPLIT;      // items per workgroup
    unsigned char *recs = smem;
    const int LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    float4 *res = reinterpret_cast<float4 *>(recs + IPW * item_stride);   // [IPW*LP] (gx, gy, ga, -)
    const int tid = threadIdx.x;
    const int item0 = block * IPW;
    MSDA_STAMP_AT(1, 0);
    const int LqM = Lq * M;
    const int b0 = item0 / LqM, r0 = item0 - b0 * LqM, m0 = item0 % M;      // uniform (scalar unit)

    for (int idx = tid; idx < IPW * LP; idx += THREADS) {
        const int il = fdiv(idx, LP, lp_shift), pt = idx - il * LP;
        const int item = item0 + il;
        BwdRec r;
        r.off[0] = r.off[1] = r.off[2] = r.off[3] = (int)kBufOob;
        r.lh = r.lw = r.a = r.pad = 0.f;
        if (item < items) {
            const int l = fdiv(pt, P, p_shift);
            const float2 xy = reinterpret_cast<const float2 *>(loc)[(long long)item0 * LP + idx];
            const float a = attn[(long long)item0 * LP + idx];
            LevelInfo li;
            li.H = (int)shapes[2 * l]; li.W = (int)shapes[2 * l + 1]; li.start = (int)level_start[l]; li.pad = 0;
            const PointGeom<float> g = point_geom<float>(xy.x, xy.y, li.H, li.W);
            if (g.inside && level_fits(shapes[2 * l], shapes[2 * l + 1], level_start[l], S)) {
                int b, m;
                item_bm(il, b0, r0, m0, LqM, M, m_shift, b, m);
                tap_offsets_b<VT>(g, li, b - b0, m, S, M, reinterpret_cast<unsigned *>(r.off));
                r.lh = g.lh; r.lw = g.lw; r.a = a;
            }
        }
        *reinterpret_cast<BwdRec *>(recs + il * item_stride + pt * kRecBytes) = r;
    }

    const int wave = tid >> 6, lane = tid & 63, grp = lane >> 3, j = lane & 7;
    const int il = (wave / SPLIT) * 8 + grp;
    const unsigned char *rb = recs + il * item_stride;
    const __amdgpu_buffer_rsrc_t vbuf = value_window<VT>(value, b0, (min(item0 + IPW, items) - 1) / LqM, S, M);
    const unsigned joff = (unsigned)(j * 4 * sizeof(VT));
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (item0 + il < items) g4 = Row<VT>::load(grad_out + (long long)(item0 + il) * kD + j * 4);
    __syncthreads();
    MSDA_STAMP_AT(1, 1);

    // 4 points per trip: all records, then all 16 row loads, then the arithmetic — so that one
    // memory round trip covers the trip instead of one per point.
    for (int p0 = wave % SPLIT; p0 < LP; p0 += 4 * SPLIT) {
        uint4 off[4]; float4 f[4]; float4 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * SPLIT;
            off[u] = make_uint4(kBufOob, kBufOob, kBufOob, kBufOob); f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < LP) {
                off[u] = *reinterpret_cast<const uint4 *>(rb + p * kRecBytes);
                f[u] = *reinterpret_cast<const float4 *>(rb + p * kRecBytes + 16);   // lh lw a -
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u][0] = BufRow<VT>::load(vbuf, off[u].x + joff); v[u][1] = BufRow<VT>::load(vbuf, off[u].y + joff);
            v[u][2] = BufRow<VT>::load(vbuf, off[u].z + joff); v[u][3] = BufRow<VT>::load(vbuf, off[u].w + joff);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * SPLIT;
            const float lh = f[u].x, lw = f[u].y, a = f[u].z, hh = 1.f - lh, hw = 1.f - lw;
            const float k1 = hh * hw, k2 = hh * lw, k3 = lh * hw, k4 = lh * lw;
            if constexpr (ATOMIC && sizeof(VT) == 4) {
                // (record offsets are bytes from the workgroup's first batch element, as for `value`)
                unsigned char *gvb = reinterpret_cast<unsigned char *>(reinterpret_cast<float *>(grad_value) + (long long)b0 * S * M * kD + j * 4);
                const unsigned o4[4] = {off[u].x, off[u].y, off[u].z, off[u].w};
                const float k[4] = {k1, k2, k3, k4};
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (o4[t] != kBufOob) { const float c = k[t] * a; float *d = reinterpret_cast<float *>(gvb + o4[t]);
                        atomicAdd(d, c * g4.x); atomicAdd(d + 1, c * g4.y); atomicAdd(d + 2, c * g4.z); atomicAdd(d + 3, c * g4.w); }
            }
            const float d1 = dot4(g4, v[u][0]), d2 = dot4(g4, v[u][1]), d3 = dot4(g4, v[u][2]), d4 = dot4(g4, v[u][3]);
            float s_a, s_x, s_y;
            tap_sums(lh, lw, a, d1, d2, d3, d4, s_a, s_x, s_y);
            s_a = octlane_sum(s_a); s_x = octlane_sum(s_x); s_y = octlane_sum(s_y);
            if (j == 0 && p < LP) res[il * LP + p] = make_float4(s_x, s_y, s_a, 0.f);
        }
    }
    __syncthreads();
    MSDA_STAMP_AT(1, 2);

    // ---- coalesced write-out of grad_sampling_loc / grad_attn_weight for the workgroup's items ----
    float2 *refpart = reinterpret_cast<float2 *>(recs);                 // [IPW][L]: the records are no longer needed
    for (int idx = tid; idx < IPW * LP; idx += THREADS) {
        const int il2 = fdiv(idx, LP, lp_shift), pt = idx - il2 * LP;
        if (item0 + il2 < items) {
            const int l = fdiv(pt, P, p_shift);
            const float4 r = res[idx];
            const float gx = r.x * (float)(int)shapes[2 * l + 1], gy = r.y * (float)(int)shapes[2 * l];
            if (FUSED) {
                // d loc / d offset = 1 / (W, H): the pixel-space sums are the offset gradients
                const long long e = (long long)item0 * LP + idx, row = fdiv(item0 + il2, M, m_shift);
                reinterpret_cast<float2 *>(grad_loc)[e + row * pro.off_pad] = make_float2(r.x, r.y);
                // softmax backward over the item's L*P lanes
                const float a = attn[e];
                const float dot = group_sum(a * r.z, LP);
                grad_attn[e + row * pro.log_pad] = a * (r.z - dot);
                // reference point: sum over the level's P points here, over the heads below
                const float sx = group_sum(gx, P), sy = group_sum(gy, P);
                if ((pt & (P - 1)) == 0) refpart[il2 * L + l] = make_float2(sx, sy);
            } else {
                reinterpret_cast<float2 *>(grad_loc)[(long long)item0 * LP + idx] = make_float2(gx, gy);
                grad_attn[(long long)item0 * LP + idx] = r.z;
            }
        }
    }
    if (FUSED) {
        __syncthreads();
        for (int i = tid; i < (IPW / M) * L; i += THREADS) {           // one thread per (query of this workgroup, level)
            const int qi = i / L, l = i - qi * L;
            const int first = item0 + qi * M;
            if (first < items) {
                float2 acc = make_float2(0.f, 0.f);
                for (int mm = 0; mm < M; ++mm) { const float2 v = refpart[(qi * M + mm) * L + l]; acc.x += v.x; acc.y += v.y; }
                reinterpret_cast<float2 *>(pro.grad_ref)[(long long)(first / M) * L + l] = acc;   // first / M = b*Lq + q
            }
        }
    }
    MSDA_STAMP_AT(1, 3);
}

template <int SPLIT, bool ATOMIC, typename VT, bool FUSED = false>
__global__ __launch_bounds__(kBlock) void bwd_query_d32_kernel(
    const VT *__restrict__ grad_out, const VT *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L, int Lq,
    int P, int items, int p_shift, int lp_shift, int m_shift, VT *__restrict__ grad_value,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn, int xcd, const PrologueOut pro = PrologueOut{nullptr, 0, 0})
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bwd_query_body<SPLIT, ATOMIC, kBlock, VT, FUSED>(grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P, items,
                                                 p_shift, lp_shift, m_shift, grad_value, grad_loc, grad_attn,
                                                 xcd ? xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x, smem, pro);
}

}  // namespace msda
#include "msda_d32_lds.h"        // large problems: forward and role A with the coarse levels served from LDS
#include "msda_d32_value.h"      // role B, per-tap records: gathers, bwd_value_body, bwd_value_wide_body, value_block_to_range
#include "msda_d32_dense.h"      // role B, coarse levels: [pixels x queries] weights x grad_out on the matrix cores
namespace msda {

template <int ACC, int PPT, typename VT, typename GT = VT, bool DET = false>
__global__ __launch_bounds__(kSBlock) void bwd_value_d32_kernel(
    const VT *__restrict__ grad_out, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc,
    const float *__restrict__ attn, int S, int M, int L, int Lq, int P, int p_shift, int tp_cap, int W,
    GT *__restrict__ grad_value, int xcd)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // grid = W ranges x L levels x N*M pairs, range fastest
    const int bid = xcd ? xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    int pr, l, ti, Wl;
    value_block_to_range(bid, W, L, shapes, pr, l, ti, Wl, ACC != kAccWide);
    bwd_value_body<ACC, PPT, VT, GT, false, DET>(grad_out, shapes, level_start, loc, attn, S, M, L, Lq, P, p_shift, tp_cap,
                                                 grad_value, ti, Wl, l, pr, smem);
}

// One launch for the whole backward of a single-pass problem: the first nB workgroups are role B
// (grad_value), the rest role A (grad_sampling_loc / grad_attn_weight).  The two roles share no
// data, so this is plain concurrency inside one grid — it removes a dependent kernel boundary
// (~1.5 us) and lets role A's short workgroups fill the CUs around role B's longer ones.
// (second launch bound: 4 wavefronts per SIMD = two 512-thread workgroups per CU.  The kAccWide instantiations sit at
// 127-128 VGPRs; one more would silently halve the occupancy — measured: cfg-4 encoder 304 -> 442 us.)
template <int SPLIT, int ACC, typename VT, bool FUSED = false, typename GT = VT, bool FIXED = false, bool DET = false>
__global__ __launch_bounds__(kSBlock, 4) void bwd_fused_d32_kernel(
    const VT *__restrict__ grad_out, const VT *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L, int Lq,
    int P, int items, int p_shift, int lp_shift, int m_shift, int tp_cap, int W, int nB,
    GT *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attn,
    const PrologueOut pro = PrologueOut{nullptr, 0, 0}, int xcd = 0, const PointEntry *__restrict__ table = nullptr,
    const RangeHeader *__restrict__ header = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int bid = (int)blockIdx.x;
    MSDA_SKIP_ROLE(bid < nB);
    if (bid < nB) {
        if (xcd) bid = xcd_block(bid, nB);
        if constexpr (FIXED) {                        // small problems (plan_fused): every workgroup of the launch resident
            bwd_value_small_body<VT, GT, DET>(grad_out, shapes, level_start, loc, attn, table, header, S, M, L, Lq, P, p_shift, tp_cap,
                                              grad_value, bid, W, smem);
        } else {
            int pr, l, ti, Wl;
            value_block_to_range(bid, W, L, shapes, pr, l, ti, Wl, ACC != kAccWide);
            bwd_value_body<ACC, kSinglePPT, VT, GT, false, DET>(grad_out, shapes, level_start, loc, attn, S, M, L, Lq, P, p_shift,
                                                                tp_cap, grad_value, ti, Wl, l, pr, smem);
        }
    } else {
        // role A never touches grad_value unless it scatters with atomics (separate kernel, MSDA_BWD_MODE=atomic)
        bwd_query_body<SPLIT, false, kSBlock, VT, FUSED>(grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P,
                                                     items, p_shift, lp_shift, m_shift, static_cast<VT *>(nullptr), grad_loc, grad_attn,
                                                     xcd ? xcd_block(bid - nB, (int)gridDim.x - nB) : bid - nB, smem, pro);
    }
}

// The same single launch for LARGE problems: role B as above, role A on the LDS-stage body of msda_d32_lds.h (chunks of
// one (batch, head) pair's queries, coarse levels served from LDS).  Both roles are 512-thread workgroups, two per CU.
// DENSE: the instantiation that may send coarse levels to the matrix cores (msda_d32_dense.h).  A launch whose plan rules
// that out (more than two ranges per level) takes the other one: the dense body is a real function, and a kernel that contains
// the call reserves scratch for every wavefront — cfg-2 encoder, where no level qualifies, paid 1.5 us of 57 for it.
template <int ACC, typename VT, bool FUSED, typename GT, int NS, bool DET = false, bool DENSE = true>
__global__ __launch_bounds__(kSBlock, 4) void bwd_fused_lds_d32_kernel(
    const VT *__restrict__ grad_out, const VT *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L,
    int Lq, int P, int p_shift, int lp_shift, int tp_cap, int W, int nB, int chunks, int qw, int stage_rows,
    GT *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attn, const PrologueOut pro, int xcd,
    int lds_bytes, const MaskIn mi = MaskIn{nullptr, nullptr})
{
    static_assert(kSBlock == kLBlock, "both roles run in 512-thread workgroups");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int bid = (int)blockIdx.x;
    MSDA_SKIP_ROLE(bid < nB);
    if (bid < nB) {
        if (xcd) bid = xcd_block(bid, nB);
        int pr, l, ti, Wl;
        value_block_to_range(bid, W, L, shapes, pr, l, ti, Wl, ACC != kAccWide);
        // a level of a few dozen pixels: its taps are a dense [pixels x queries] matrix — matrix cores (msda_d32_dense.h)
        if (DENSE && !MSDA_DIAG(7) && dense_level(shapes[2 * l], shapes[2 * l + 1], level_start[l], S, Wl, lds_bytes)) {
            if (l == 0 && ti == 0) zero_uncovered_rows<GT, kSBlock>(shapes, level_start, S, M, L, grad_value, pr / M, pr % M);
            bwd_value_dense_body<VT, GT>(grad_out, shapes, level_start, loc, attn, S, M, L, Lq, P, grad_value, ti, Wl, l, pr, smem);
            return;
        }
        bwd_value_body<ACC, kSinglePPT, VT, GT, false, DET>(grad_out, shapes, level_start, loc, attn, S, M, L, Lq, P, p_shift,
                                                            tp_cap, grad_value, ti, Wl, l, pr, smem, mi);
    } else {
        bwd_query_lds_body<VT, FUSED, NS>(grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P, p_shift, lp_shift, chunks,
                                          qw, stage_rows, grad_loc, grad_attn, pro,
                                          xcd ? xcd_block(bid - nB, (int)gridDim.x - nB) : bid - nB, smem);
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
#if MSDA_D32_HAS(0)
bool d32_supported(int N, int S, int M, int D, int L, int Lq, int P)
{
    if (D != kD || L > kMaxLevels || L * P > 32) return false;
    const long long items = (long long)N * Lq * M;
    if ((long long)N * S * M * kD >= (1LL << 31)) return false;        // int32 element offsets (value / grad_value)
    if (items * kD >= (1LL << 31)) return false;                       // ... and of out / grad_out rows (q * M*32 in role B's gathers)
    if (items * L * P * 2 >= (1LL << 31) || items >= (1LL << 30)) return false;
    if ((long long)N * M > 65535 || S > (1 << 19)) return false;       // role-B workgroup count and S*W stay 32-bit
    // the tiled kernels address `value` with 32-bit BYTE offsets inside the batch elements a workgroup's (at most 64) items
    // span (value_window): that window — two elements unless a batch element has fewer than 64 items — stays below 2^31 bytes
    // role B reads a (batch, head) pair's grad_out rows through a descriptor with 32-bit byte offsets q * M * 128 (24-bit multiplies)
    if ((long long)Lq * M * kD * 4 >= (1LL << 31) || Lq >= (1 << 23) || (long long)M * kD * 4 >= (1 << 24)) return false;
    if (Lq <= 0 || M <= 0) return true;                                 // (degenerate sizes never reach a kernel)
    const long long span = min((long long)N, 2 + 63 / ((long long)Lq * M));
    if (span * S * M * kD * 4 >= (1LL << 31)) return false;
    return true;
}
#endif

// Every MSDA_* knob named in this file is read in DIAGNOSTIC builds only (-DMSDA_TUNING, msda_launch.h); the shipped
// library gets the defaults.
static int env_int(const char *name, int dflt) { return tuning_int(name, dflt); }

static int bwd_target_wgs()
{
    static const int v = [] { int t = env_int("MSDA_BWD_WGS", 256); return t < 1 ? 1 : t; }();
    return v;
}

// coarse levels on the matrix cores (msda_d32_dense.h); MSDA_DENSE=0: every level through the sort + gather bodies (A/B knob)
static int dense_on()
{
    static const int v = env_int("MSDA_DENSE", 1) != 0;
    return v;
}

// XCD-aware workgroup numbering (xcd_block); MSDA_XCD=0 keeps the hardware's round-robin order (A/B knob)
static int xcd_remap()
{
    static const int v = env_int("MSDA_XCD", 1) != 0;
    return v;
}

static int pow2_shift(int x) { return (x > 0 && (x & (x - 1)) == 0) ? __builtin_ctz((unsigned)x) : -1; }

// Wavefronts per octet (SPLIT).  Small problems are latency-bound: more wavefronts with fewer
// points each keep more row loads in flight; large ones want no cross-wave reduction.
// MSDA_SPLIT=1|2|4 overrides (tuning knob, read once).
static int pick_split(int items, int LP)
{
    static const int forced = env_int("MSDA_SPLIT", 0);
    if (forced == 1 || forced == 2 || forced == 4) return (LP >= forced) ? forced : 1;
    const int octets = (items + 7) / 8;
    if (LP >= 4 && octets <= 2048) return 4;
    if (LP >= 2 && octets <= 8192) return 2;
    return 1;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static int allow_lds(const void *fn, size_t bytes);

// Large problems take the LDS-stage kernels of msda_d32_lds.h: `chunks` workgroups per (batch, head) pair, each staging the
// pair's coarse levels once and serving `qw` queries in sub-batches of 64.  At least ~1024 workgroups (two resident per
// CU, 512 threads each) unless that would leave fewer than one sub-batch per workgroup; MSDA_LDS=0/1 overrides (tuning).
struct LdsPlan { bool use; int chunks, qw, stage_rows; size_t lds; };
template <typename VT>
static LdsPlan plan_lds(int N, int S, int M, int L, int Lq, int P, long long role_b_wgs = 0)
{
    static const int mode = env_int("MSDA_LDS", -1);
    static const int target_alone = env_int("MSDA_LDS_WGS", 512);        // two 512-thread workgroups per CU: ONE round
    static const int target_fused = env_int("MSDA_LDS_WGS_BWD", 0);      // role A inside the fused backward launch (0: rule below)
    // Inside the fused backward role B's workgroups are dispatched first.  If they leave slots free (a launch of fewer than
    // the 512 resident workgroups), role A gets one round in exactly those slots; behind several rounds of role B it is cut
    // finer, so that the launch does not end on a few long role-A workgroups (cfg-2 encoder: 384 role-B workgroups -> 128
    // for role A, 59.4 -> 57.0 us; cfg-4 encoder: 2048 -> 1024, 219 -> 215 us; profiles/r03_notes.md).
    int target = target_alone;
    // (fewer than 128 free slots make role A's few workgroups the tail instead: deterministic cfg-2 encoder, 448 role-B
    // workgroups -> 64 for role A 88 us, 1024 62 us)
    if (role_b_wgs > 0) target = target_fused > 0 ? target_fused : role_b_wgs <= 512 - 128 ? (int)(512 - role_b_wgs) : 1024;
    LdsPlan pl;
    const long long pairs = (long long)N * M, items = pairs * Lq;
    pl.stage_rows = min(S, kLStageBytes / (int)(kD * sizeof(VT)));
    // chunks per pair: about `target` workgroups in all, at least one octet per wavefront — and at least two per wavefront
    // when cutting finer than one resident round (10 octets on 8 wavefronts leave six of them idle half the time:
    // cfg-4 decoder 76 -> 82 us)
    const int octets = ceil_div(Lq, 8);
    auto octets_per_wg = [&](int tgt) {
        const int want_chunks = (int)max(1LL, min((long long)ceil_div(octets, kLWaves), (tgt + pairs - 1) / pairs));
        return ceil_div(octets, want_chunks);
    };
    int opw = octets_per_wg(target);
    if (target > 512 && opw < 2 * kLWaves) opw = octets_per_wg(512);
    pl.qw = 8 * opw;
    pl.chunks = ceil_div(Lq, pl.qw);
    pl.lds = lds_variant_bytes<VT>(pl.stage_rows, L * P);
    const bool large = items >= 32768 && Lq >= kLItems;                       // below that the launch is latency-bound
    // (a batch element's slice of `value` is addressed through a buffer descriptor with 32-bit byte offsets)
    // L*P <= 16: two point slots per lane and octet (with more, role A's look-ahead registers spill)
    pl.use = (mode < 0 ? large : mode != 0) && pairs * pl.chunks <= 0x7fffffffLL && Lq > 0 && L * P <= 16 &&
             (long long)S * M * kD * (long long)sizeof(VT) < (1LL << 31);
    return pl;
}

template <typename VT, bool FUSED>
static int launch_fwd_lds(const LdsPlan &lp, const VT *value, const int64_t *shapes, const int64_t *level_start, const float *loc,
                          const float *attn, int N, int S, int M, int L, int Lq, int P, VT *out, const PrologueIn &pro,
                          hipStream_t stream, const MaskOut mo = MaskOut{nullptr, nullptr, 0})
{
    const dim3 grid((unsigned)(N * M * lp.chunks));
    // (with masks to write: the ranges' first pixels [L][W + 1] and a float scale per level behind the records)
    const size_t lds = lp.lds + (mo.masks ? (size_t)(L * (mo.W + 1) + L) * sizeof(int) : 0);
#define MSDA_LAUNCH_FL(NS_)                                                                            \
    do { if (int rc = allow_lds(reinterpret_cast<const void *>(fwd_d32_lds_kernel<VT, FUSED, NS_>), lds)) return rc;               \
         hipLaunchKernelGGL((fwd_d32_lds_kernel<VT, FUSED, NS_>), grid, dim3(kLBlock), lds, stream, value, shapes, level_start,    \
                            loc, attn, S, M, L, Lq, P, pow2_shift(P), pow2_shift(L * P), lp.chunks, lp.qw, lp.stage_rows, out,   \
                            pro, xcd_remap(), mo); } while (0)
    if (8 * L * P <= kWave) MSDA_LAUNCH_FL(1); else MSDA_LAUNCH_FL(2);             // plan_lds: L*P <= 16
#undef MSDA_LAUNCH_FL
    return check_launch("msda forward (d32, LDS stage)");
}

template <typename VT, bool FUSED>
static int launch_query_lds(const LdsPlan &lp, const VT *grad_out, const VT *value, const int64_t *shapes,
                            const int64_t *level_start, const float *loc, const float *attn, int N, int S, int M, int L, int Lq,
                            int P, float *grad_loc, float *grad_attn, const PrologueOut &pro, hipStream_t stream)
{
    const dim3 grid((unsigned)(N * M * lp.chunks));
#define MSDA_LAUNCH_QL(NS_)                                                                            \
    do { if (int rc = allow_lds(reinterpret_cast<const void *>(bwd_query_d32_lds_kernel<VT, FUSED, NS_>), lp.lds)) return rc;      \
         hipLaunchKernelGGL((bwd_query_d32_lds_kernel<VT, FUSED, NS_>), grid, dim3(kLBlock), lp.lds, stream, grad_out, value,      \
                            shapes, level_start, loc, attn, S, M, L, Lq, P, pow2_shift(P), pow2_shift(L * P), lp.chunks, lp.qw,  \
                            lp.stage_rows, grad_loc, grad_attn, pro, xcd_remap()); } while (0)
    if (8 * L * P <= kWave) MSDA_LAUNCH_QL(1); else MSDA_LAUNCH_QL(2);             // plan_lds: L*P <= 16
#undef MSDA_LAUNCH_QL
    return check_launch("msda backward (d32, query-major, LDS stage)");
}

// ranges per level / rows per range of the backward's role-B plan, for the header the forward writes behind the point table
static void table_header_plan(int N, int S, int M, int L, int Lq, int P, int &W, int &tp_cap);
// the header behind the point entries of a table (null: no table)
static const RangeHeader *table_header_of(const PointEntry *table, int N, int M, int L, int Lq, int P, int W)
{
    if (!table) return nullptr;
    return reinterpret_cast<const RangeHeader *>(table + (long long)N * M * L * Lq * P);
}

// per-point range masks of large problems (MaskHeader): plan_masks below
static MaskOut mask_out(void *table, int N, int S, int M, int L, int Lq, int P);

template <typename VT>
static int launch_fwd_d32_t(const VT *value, const int64_t *shapes, const int64_t *level_start,
                            const float *loc, const float *attn, int N, int S, int M, int L, int Lq, int P,
                            VT *out, hipStream_t stream, PointEntry *table = nullptr)
{
    const LdsPlan lp = plan_lds<VT>(N, S, M, L, Lq, P);
    if (lp.use)
        return launch_fwd_lds<VT, false>(lp, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, out,
                                         PrologueIn{nullptr, nullptr, nullptr, 0, 0}, stream, mask_out(table, N, S, M, L, Lq, P));
    const int items = N * Lq * M, LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    const int split = pick_split(items, LP);
    const int ipw = 32 / split;
    const size_t lds = (size_t)ipw * item_stride + (split > 1 ? 4096 : 0);
    int hdr_W = 0, hdr_tp_cap = 0;
    if (table) table_header_plan(N, S, M, L, Lq, P, hdr_W, hdr_tp_cap);
    const dim3 grid((items + ipw - 1) / ipw + (hdr_W > 0 ? 1 : 0)), block(kBlock);      // (+ the header's workgroup)
#define MSDA_LAUNCH_FWD(SP)                                                                            \
    hipLaunchKernelGGL((fwd_d32_kernel<SP, VT>), grid, block, lds, stream, value, shapes, level_start, loc, attn, \
                       S, M, L, Lq, P, items, pow2_shift(P), pow2_shift(LP), pow2_shift(M), out,                  \
                       PrologueIn{nullptr, nullptr, nullptr, 0, 0}, xcd, table, hdr_W, hdr_tp_cap, kSmallRecCap)
    const int xcd = xcd_remap();
    if (split == 4) MSDA_LAUNCH_FWD(4); else if (split == 2) MSDA_LAUNCH_FWD(2); else MSDA_LAUNCH_FWD(1);
#undef MSDA_LAUNCH_FWD
    return check_launch("msda forward (d32)");
}

// Role-B plan (host side; only S, the batch and Lq*P are known here — the level geometry lives on
// the device): W ranges per level, PPT points per thread per pass, accumulation mode.
struct ValuePlan { int W, tp_cap, ppt, acc; size_t lds; };

template <typename VT>
static ValuePlan plan_value(int N, int S, int M, int L, int Lq, int P, int target_wgs, bool det = false)
{
    ValuePlan pl;
    const int NP = Lq * P, pairs_levels = N * M * L;
    const bool multipass = NP > kSingleMaxPoints;
    static const int wide = env_int("MSDA_BWD_WIDE", 1);             // A/B knob: 0 = the chunked kAccRmw passes
    pl.acc = !multipass ? kAccNone : (sizeof(VT) == 4 ? (wide ? kAccWide : kAccRmw) : kAccTile);
    if (pl.acc == kAccWide) {
        // enough ranges per level that the taps a workgroup keeps (4*NP/W on average) fit its record array with
        // 15 % to spare, and at least one workgroup per CU.  No more than that: every range re-scans its level
        // (cfg-2 encoder: 67 us with 6 ranges = 384 workgroups, 73 with 8 = 512, 84 with 12; role A's workgroups take
        // the free slots from the start)
        static const int wide_wgs = env_int("MSDA_WIDE_WGS", 1);     // x target_wgs
        const int w_max = max(1, S / 16);
        pl.W = max(1, min(max(ceil_div(S, kSingleMaxRows), ceil_div(wide_wgs * target_wgs, pairs_levels)), w_max));
        for (;; ++pl.W) {                                            // the record capacity depends on the rows per range
            pl.tp_cap = (ceil_div(S, pl.W) + 3) & ~3;
            int rec_cap, list_cap;
            wide_caps(pl.tp_cap, NP, rec_cap, list_cap, det);
            if (pl.W >= w_max || (long long)4 * NP * 115 / 100 <= (long long)rec_cap * pl.W) break;
        }
        pl.ppt = kSinglePPT;
        pl.lds = (size_t)wide_lds_bytes(pl.tp_cap, NP, det);
        return pl;
    }
    if (pl.acc == kAccTile) {
        pl.tp_cap = kMultiRows;
        pl.W = ceil_div(S, pl.tp_cap);
        pl.ppt = kMultiPPT;
    } else {
        // multi-pass problems: twice the workgroups (each pass is a latency chain; more, smaller
        // workgroups overlap better) and the smaller record array, so role A still shares the launch
        // (measured: cfg-2 encoder 108 vs 120 us; cfg-4 encoder unchanged)
        pl.W = max(ceil_div(S, kSingleMaxRows), ceil_div(multipass ? 2 * target_wgs : target_wgs, pairs_levels));
        pl.W = max(1, min(pl.W, max(1, S / 16)));
        pl.tp_cap = ceil_div(S, pl.W);
        static const int force_ppt = env_int("MSDA_BWD_PPT", 0);         // tuning knob: kSinglePPT or kMultiPPT
        pl.ppt = (multipass && force_ppt == kMultiPPT) ? kMultiPPT : kSinglePPT;
    }
    pl.tp_cap = (pl.tp_cap + 3) & ~3;                                 // keeps the LDS arrays 16-B aligned
    const int pass_points = min(NP, pl.ppt * kSBlock);
    // (DET: four counter words per row and the row totals besides the segment starts)
    pl.lds = (pl.acc == kAccTile ? (size_t)pl.tp_cap * kD * 4 : 0) + ((det ? 6 : 2) * (size_t)pl.tp_cap + 32) * 4 +
             (size_t)4 * pass_points * sizeof(SRec) + (size_t)kOvfCap * sizeof(SOvf);
    return pl;
}

static ValuePlan plan_value_f32(int N, int S, int M, int L, int Lq, int P) { return plan_value<float>(N, S, M, L, Lq, P, bwd_target_wgs(), false); }

// Per-point range masks (MaskHeader): where the forward is the LDS-stage kernel and role B the kept-taps pass with W <= 8
// ranges per level — the same plan functions the launchers run, so the forward's writer and the backward's reader agree
// (the header repeats the plan and role B checks it).  Row storage does not matter.
struct MaskPlan { bool use; int W; size_t bytes; };
static MaskPlan plan_masks(int N, int S, int M, int L, int Lq, int P)
{
    static const int enabled = env_int("MSDA_MASKS", 1);                // A/B knob (tuning build): 0 = role B scans sampling_loc
    MaskPlan mp{false, 0, 0};
    if (!enabled || !plan_lds<float>(N, S, M, L, Lq, P).use) return mp;
    const ValuePlan pl = plan_value_f32(N, S, M, L, Lq, P);
    const long long NP = (long long)Lq * P;
    // W >= kMinMaskRanges: the byte costs the forward ~35 instructions per point lane (+10 % of its time), and each range's
    // workgroup saves the strided float scan of all Lq*P points — a trade that pays from four ranges per level on (cfg-2
    // encoder, W = 6: forward +1.7 us, backward -2.5 ... -4 us) and loses with two (cfg-4 encoder: forward +4.5 us, backward
    // +2 us — its scan reads every point twice either way; profiles/r05_notes.md section 1)
    static const int min_w = env_int("MSDA_MASKS_MIN_W", kMinMaskRanges);
    if (pl.acc != kAccWide || pl.ppt != kSinglePPT || pl.W > kMaxMaskRanges || pl.W < min_w || (NP & 3) || (long long)L * NP >= (1LL << 31)) return mp;
    mp.use = true; mp.W = pl.W;
    mp.bytes = sizeof(MaskHeader) + (size_t)N * M * L * (size_t)NP;
    return mp;
}
static MaskOut mask_out(void *table, int N, int S, int M, int L, int Lq, int P)
{
    MaskOut mo{nullptr, nullptr, 0};
    if (!table) return mo;
    const MaskPlan mp = plan_masks(N, S, M, L, Lq, P);
    if (!mp.use) return mo;
    unsigned char *b = static_cast<unsigned char *>(table);
    return MaskOut{reinterpret_cast<MaskHeader *>(b), b + sizeof(MaskHeader), mp.W};
}
// (role B reads them only where its own plan is the kept-taps pass with the same W; never under the deterministic flag)
static MaskIn mask_in(const void *table, int N, int S, int M, int L, int Lq, int P, int acc, int W, bool deterministic)
{
    MaskIn mi{nullptr, nullptr};
    if (!table || deterministic || acc != kAccWide) return mi;
    const MaskPlan mp = plan_masks(N, S, M, L, Lq, P);
    if (!mp.use || mp.W != W) return mi;
    const unsigned char *b = static_cast<const unsigned char *>(table);
    return MaskIn{reinterpret_cast<const MaskHeader *>(b), b + sizeof(MaskHeader)};
}

static int allow_lds(const void *fn, size_t bytes)
{
    if (bytes <= 64 * 1024) return MSDA_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? MSDA_OK : set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
}

// The fused backward launch of a SMALL problem.  The fused kernel keeps two 512-thread workgroups per CU (its
// ~100 VGPRs allow 4 wavefronts per SIMD), i.e. 512 resident at once.  When role B's and role A's workgroups
// do not all fit, the last role-A workgroups start only after a slot frees up and become the tail of the
// launch; halving their number (role A's SPLIT 4 -> 2) lets everything start at once.  If everything is resident
// the backward is one latency chain and role B takes the short sort (FIXED, see SOvf); bigger problems keep
// the prefix-sum sort, whose phases hide behind other workgroups' gathers.  MSDA_BWD_FIXED=0 disables both.
struct FusedPlan { int split; bool fixed; };
static FusedPlan plan_fused(int items, int LP, int split, long long nB, int acc, int whole_queries_of = 0, bool det = false)
{
    // whole_queries_of = M when role A must hold whole queries per workgroup (fused prologue), else 0
    static const int enabled = env_int("MSDA_BWD_FIXED", 1);
    constexpr long long kResident = 512;
    FusedPlan fp{split, false};
    if (!enabled || acc != kAccNone) return fp;
    auto n_a = [&](int sp) { const int ipw = 64 / sp; return (long long)((items + ipw - 1) / ipw); };
    if (nB + n_a(split) > kResident && split == 4 && LP >= 2 && nB + n_a(2) <= kResident &&
        (whole_queries_of == 0 || (64 / 2) % whole_queries_of == 0))
        fp.split = 2;
    fp.fixed = nB + n_a(fp.split) <= kResident;              // (det: the small body's per-wavefront-counter variant, round 5)
    return fp;
}

// per-head reference-point gradients of the fused-prologue backward on large problems, padded to 16 bytes
static size_t prologue_heads_bytes(int N, int M, int L, int Lq) { return ((size_t)N * Lq * M * L * sizeof(float2) + 15) & ~(size_t)15; }

static void table_header_plan(int N, int S, int M, int L, int Lq, int P, int &W, int &tp_cap)
{
    // (W >= 1 always: the header's magic is the table's "written" stamp; a plan with more slots than the header holds gets
    // the stamp and no entries — write_range_header)
    const ValuePlan pl = plan_value<float>(N, S, M, L, Lq, P, bwd_target_wgs(), false);
    W = pl.W;
    tp_cap = pl.tp_cap;
}

#if MSDA_D32_HAS(0)
// A forward call that was handed a table buffer but runs kernels that write none (the generic family: rows that are not
// 16-byte aligned) clears the table's stamp, so that the backward can never mistake an earlier call's contents — the caching
// allocator may hand the same block out again — for this call's.
int invalidate_forward_table(void *table, int N, int S, int M, int L, int Lq, int P, hipStream_t stream)
{
    const size_t off = plan_lds<float>(N, S, M, L, Lq, P).use ? 0 : (size_t)N * M * L * Lq * P * sizeof(PointEntry);
    static_assert(offsetof(RangeHeader, magic) == 0 && offsetof(MaskHeader, magic) == 0, "the stamp is the header's first word");
    const hipError_t e = hipMemsetAsync(static_cast<unsigned char *>(table) + off, 0, sizeof(int), stream);
    return e == hipSuccess ? MSDA_OK : set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
}

// Bytes of the point table a forward of this geometry can leave for its backward (msda_forward_workspace_bytes): non-zero
// exactly where the backward would take the small-problem role B (plan_fused.fixed: single pass, every workgroup resident,
// no LDS stage) — the same plan functions the launchers run.  prologue: the fused-prologue entry points (whole queries per
// role-A workgroup change the split).
size_t forward_table_bytes(int N, int S, int M, int D, int L, int Lq, int P, bool prologue)
{
    if (!d32_supported(N, S, M, D, L, Lq, P) || (prologue && !prologue_supported(N, S, M, D, L, Lq, P))) return 0;
    if (plan_lds<float>(N, S, M, L, Lq, P).use) {                       // large problems: per-point range masks, if role B reads them
        const MaskPlan mm = plan_masks(N, S, M, L, Lq, P);
        return mm.use ? mm.bytes : 0;
    }
    const ValuePlan pl = plan_value<float>(N, S, M, L, Lq, P, bwd_target_wgs(), false);
    if (pl.acc != kAccNone || pl.ppt != kSinglePPT) return 0;
    const FusedPlan fp = plan_fused(N * Lq * M, L * P, pick_split(N * Lq * M, L * P), (long long)pl.W * N * M * L, pl.acc, prologue ? M : 0, false);
    return fp.fixed ? (size_t)N * M * L * Lq * P * sizeof(PointEntry) + range_header_bytes() : 0;
}


size_t backward_workspace_bytes(int N, int S, int M, int D, int L, int Lq, int P, unsigned flags)
{
    if (!d32_supported(N, S, M, D, L, Lq, P)) return 0;
    // The fused-prologue backward on large problems: role A sees one head per workgroup and leaves the reference-point
    // gradient per head ([N, Lq, M, L, 2]) for ref_heads_reduce_kernel.  Nothing else needs scratch: the deterministic
    // kernels (MSDA_FLAG_DETERMINISTIC) order a row's records through per-wavefront counters in LDS.
    if ((flags & MSDA_FLAG_PROLOGUE) && plan_lds<float>(N, S, M, L, Lq, P).use) return prologue_heads_bytes(N, M, L, Lq);
    return 0;
}
#endif

// VT = storage of value / grad_out, GT = storage of grad_value (VT, or float for bf16 rows)
template <typename VT, typename GT = VT>
static int launch_bwd_d32_t(const VT *grad_out, const VT *value, const int64_t *shapes,
                            const int64_t *level_start, const float *loc, const float *attn, int N, int S,
                            int M, int L, int Lq, int P, GT *grad_value, float *grad_loc, float *grad_attn,
                            hipStream_t stream, void *workspace = nullptr, size_t ws_bytes = 0, bool deterministic = false,
                            const PointEntry *table = nullptr, bool no_dense = false)
{
    const int items = N * Lq * M, LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    // deterministic (MSDA_FLAG_DETERMINISTIC): the same launches with role B's DET instantiations (per-wavefront counters,
    // msda_d32_value.h) and without the short sort, which ranks a row's records in arrival order
    // tuning / A-B knobs, read once per process: MSDA_BWD_MODE=atomic selects the v1 global-atomic
    // scatter (fp32 only), =split launches role B and role A as two kernels; MSDA_BWD_WGS is the number
    // of role-B workgroups to aim for on small problems.
    static const int bwd_mode_env = [] {                   // 0 fused (default), 1 split launches, 2 v1 atomics
        const char *v = tuning_str("MSDA_BWD_MODE");
        return (v && !strcmp(v, "atomic")) ? 2 : (v && !strcmp(v, "split")) ? 1 : 0; }();
    const int bwd_mode = (bwd_mode_env == 2 && (sizeof(VT) != 4 || deterministic)) ? 0 : bwd_mode_env;
    const int target_wgs = bwd_target_wgs();
    const int split = pick_split(items, LP);
    const int ps = pow2_shift(P), lps = pow2_shift(LP), ms = pow2_shift(M);
    const int xcd = xcd_remap();
    // large problems: role A as its own launch with the coarse levels in LDS (msda_d32_lds.h), role B as its own launch
    LdsPlan lds_a = plan_lds<VT>(N, S, M, L, Lq, P);
    if (bwd_mode == 2) lds_a.use = false;

#ifdef MSDA_TUNING
    if (bwd_mode == 2) {
        hipError_t e = hipMemsetAsync(grad_value, 0, sizeof(GT) * (size_t)N * S * M * kD, stream);
        if (e != hipSuccess) return set_error(MSDA_ERR_LAUNCH, hipGetErrorString(e));
    } else
#endif
    {
        const ValuePlan pl = plan_value<GT>(N, S, M, L, Lq, P, target_wgs, deterministic);
        const long long nB = (long long)pl.W * N * M * L;
        // ---- large problems: one launch, role A on the LDS-stage body ----
        if (lds_a.use) lds_a = plan_lds<VT>(N, S, M, L, Lq, P, nB);      // cut role A for the slots role B leaves
        if (bwd_mode == 0 && lds_a.use && pl.ppt == kSinglePPT && (pl.acc == kAccNone || pl.acc == kAccWide) &&
            nB + (long long)N * M * lds_a.chunks <= 0x7fffffffLL) {
            const dim3 fgrid((unsigned)(nB + (long long)N * M * lds_a.chunks));
            // (at least what the dense coarse-level body needs: two such workgroups still share a CU)
            const size_t flds = max(max(pl.lds, lds_a.lds), (size_t)dense_lds_bytes());
            const bool dense = dense_on() && pl.W <= MSDA_DENSE_MAX_W && !no_dense;
            const MaskIn masks = mask_in(table, N, S, M, L, Lq, P, pl.acc, pl.W, deterministic);
#define MSDA_LAUNCH_FLD__(AC, NS_, DT, DN)                                                             \
            do { if (int rc = allow_lds(reinterpret_cast<const void *>(bwd_fused_lds_d32_kernel<AC, VT, false, GT, NS_, DT, DN>), flds)) return rc; \
            hipLaunchKernelGGL((bwd_fused_lds_d32_kernel<AC, VT, false, GT, NS_, DT, DN>), fgrid, dim3(kSBlock), flds, stream, grad_out,  \
                               value, shapes, level_start, loc, attn, S, M, L, Lq, P, ps, lps, pl.tp_cap, pl.W, (int)nB,          \
                               lds_a.chunks, lds_a.qw, lds_a.stage_rows, grad_value, grad_loc, grad_attn,                        \
                               PrologueOut{nullptr, 0, 0}, xcd, (DN) ? (int)flds : 0, masks); } while (0)
#define MSDA_LAUNCH_FLD_(AC, NS_, DT) do { if (dense) MSDA_LAUNCH_FLD__(AC, NS_, DT, true); else MSDA_LAUNCH_FLD__(AC, NS_, DT, false); } while (0)
#define MSDA_LAUNCH_FLD(AC, NS_) do { if (deterministic) MSDA_LAUNCH_FLD_(AC, NS_, true); else MSDA_LAUNCH_FLD_(AC, NS_, false); } while (0)
            const bool one_slot = 8 * LP <= kWave;
            if (pl.acc == kAccNone) { if (one_slot) MSDA_LAUNCH_FLD(kAccNone, 1); else MSDA_LAUNCH_FLD(kAccNone, 2); }
            else                    { if (one_slot) MSDA_LAUNCH_FLD(kAccWide, 1); else MSDA_LAUNCH_FLD(kAccWide, 2); }
#undef MSDA_LAUNCH_FLD
#undef MSDA_LAUNCH_FLD_
#undef MSDA_LAUNCH_FLD__
            return check_launch("msda backward (d32, fused, LDS stage)");
        }
        // ---- whole backward in one launch when role A's workgroups can share the CUs (LDS) ----
        if (bwd_mode == 0 && pl.ppt == kSinglePPT && pl.acc != kAccTile && !lds_a.use) {
            const FusedPlan fp = plan_fused(items, LP, split, nB, pl.acc, 0, deterministic);
            const int ipw_f = 64 / fp.split;                            // 512-thread role-A workgroups
            const size_t lds_a = (size_t)ipw_f * item_stride + (size_t)ipw_f * LP * 16;
            const long long nA = (items + ipw_f - 1) / ipw_f;
            if (nB + nA <= 0x7fffffffLL) {
                const dim3 fgrid((unsigned)(nB + nA));
                size_t flds = pl.lds > lds_a ? pl.lds : lds_a;
                if (fp.fixed) flds = max(flds, small_lds_bytes(pl.tp_cap, deterministic));
#define MSDA_LAUNCH_F_(SP, AC, FX, DT)                                                                 \
                do { if (int rc = allow_lds(reinterpret_cast<const void *>(bwd_fused_d32_kernel<SP, AC, VT, false, GT, FX, DT>), flds)) return rc; \
                hipLaunchKernelGGL((bwd_fused_d32_kernel<SP, AC, VT, false, GT, FX, DT>), fgrid, dim3(kSBlock), flds, stream,  \
                                   grad_out, value, shapes, level_start, loc, attn, S, M, L, Lq, P, items, ps, lps, ms,  \
                                   pl.tp_cap, pl.W, (int)nB, grad_value, grad_loc, grad_attn, PrologueOut{nullptr, 0, 0}, xcd,  \
                                   (FX) ? table : nullptr, (FX) ? table_header_of(table, N, M, L, Lq, P, pl.W) : nullptr); } while (0)
#define MSDA_LAUNCH_F(SP, AC, FX) do { if (deterministic) MSDA_LAUNCH_F_(SP, AC, FX, true); else MSDA_LAUNCH_F_(SP, AC, FX, false); } while (0)
                if (fp.fixed)                { if (fp.split == 4) MSDA_LAUNCH_F(4, kAccNone, true); else if (fp.split == 2) MSDA_LAUNCH_F(2, kAccNone, true); else MSDA_LAUNCH_F(1, kAccNone, true); }
                else if (pl.acc == kAccNone) { if (fp.split == 4) MSDA_LAUNCH_F(4, kAccNone, false); else if (fp.split == 2) MSDA_LAUNCH_F(2, kAccNone, false); else MSDA_LAUNCH_F(1, kAccNone, false); }
                else if (pl.acc == kAccWide) { if (fp.split == 4) MSDA_LAUNCH_F(4, kAccWide, false); else if (fp.split == 2) MSDA_LAUNCH_F(2, kAccWide, false); else MSDA_LAUNCH_F(1, kAccWide, false); }
#ifdef MSDA_TUNING                                                       // kAccRmw is only ever planned with MSDA_BWD_WIDE=0
                else                         { if (fp.split == 4) MSDA_LAUNCH_F(4, kAccRmw, false); else if (fp.split == 2) MSDA_LAUNCH_F(2, kAccRmw, false); else MSDA_LAUNCH_F(1, kAccRmw, false); }
#else
                else return set_error(MSDA_ERR_LAUNCH, "msda backward (d32): unplanned accumulation mode");
#endif
#undef MSDA_LAUNCH_F
#undef MSDA_LAUNCH_F_
                return check_launch("msda backward (d32, fused)");
            }
        }
        // ---- role B as its own launch (large record arrays / bf16 tile / A-B knob) ----
        if (nB > 0x7fffffffLL) return set_error(MSDA_ERR_ARGUMENT, "msda backward (d32): too many grad_value workgroups");
        const dim3 grid((unsigned)nB);
#define MSDA_LAUNCH_B_(AC, PPT_, DT)                                                                   \
        do { if (int rc = allow_lds(reinterpret_cast<const void *>(bwd_value_d32_kernel<AC, PPT_, VT, GT, DT>), pl.lds)) return rc; \
             hipLaunchKernelGGL((bwd_value_d32_kernel<AC, PPT_, VT, GT, DT>), grid, dim3(kSBlock), pl.lds, stream, grad_out,    \
                                shapes, level_start, loc, attn, S, M, L, Lq, P, ps, pl.tp_cap, pl.W, grad_value, xcd); } while (0)
#define MSDA_LAUNCH_B(AC, PPT_) do { if (deterministic) MSDA_LAUNCH_B_(AC, PPT_, true); else MSDA_LAUNCH_B_(AC, PPT_, false); } while (0)
        if (pl.acc == kAccNone) MSDA_LAUNCH_B(kAccNone, kSinglePPT);
        else if (pl.acc == kAccTile) MSDA_LAUNCH_B(kAccTile, kMultiPPT);
        else if (pl.acc == kAccWide) MSDA_LAUNCH_B(kAccWide, kSinglePPT);
#ifdef MSDA_TUNING
        else if (pl.ppt == kMultiPPT) MSDA_LAUNCH_B(kAccRmw, kMultiPPT);
        else MSDA_LAUNCH_B(kAccRmw, kSinglePPT);
#else
        else return set_error(MSDA_ERR_LAUNCH, "msda backward (d32): unplanned accumulation mode");
#endif
#undef MSDA_LAUNCH_B
#undef MSDA_LAUNCH_B_
        if (int rc = check_launch("msda backward (d32, grad_value sort+gather)")) return rc;
    }
    if (lds_a.use)
        return launch_query_lds<VT, false>(lds_a, grad_out, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, grad_loc,
                                           grad_attn, PrologueOut{nullptr, 0, 0}, stream);
    {
        const int ipw = 32 / split;
        const size_t lds = (size_t)ipw * item_stride + (size_t)ipw * LP * 16;
        const dim3 grid((items + ipw - 1) / ipw), block(kBlock);
#define MSDA_LAUNCH_A(SP, AT)                                                                          \
        hipLaunchKernelGGL((bwd_query_d32_kernel<SP, AT, VT>), grid, block, lds, stream, grad_out, value, shapes, \
                           level_start, loc, attn, S, M, L, Lq, P, items, ps, lps, ms, gv_atomic, grad_loc, grad_attn, xcd)
        VT *gv_atomic = nullptr;                                        // only the v1 atomic scatter writes grad_value here
        if constexpr (sizeof(VT) == sizeof(GT)) gv_atomic = grad_value;
#ifdef MSDA_TUNING
        if (bwd_mode == 2) { if (split == 4) MSDA_LAUNCH_A(4, true); else if (split == 2) MSDA_LAUNCH_A(2, true); else MSDA_LAUNCH_A(1, true); }
        else
#endif
        { if (split == 4) MSDA_LAUNCH_A(4, false); else if (split == 2) MSDA_LAUNCH_A(2, false); else MSDA_LAUNCH_A(1, false); }
#undef MSDA_LAUNCH_A
    }
    return check_launch("msda backward (d32, query-major)");
}

// ---- fused prologue (fp32 only): see PrologueIn / PrologueOut ----
#if MSDA_D32_HAS(0)
bool prologue_supported(int N, int S, int M, int D, int L, int Lq, int P)
{
    if (!d32_supported(N, S, M, D, L, Lq, P)) return false;
    const int LP = L * P;
    if (pow2_shift(LP) < 0 || pow2_shift(P) < 0 || LP > 64) return false;     // lane-group reductions
    const int split = pick_split(N * Lq * M, LP);
    if ((kSWaves * 8 / split) % M != 0) return false;                           // whole queries per role-A workgroup
    static const bool plain_modes = tuning_str("MSDA_BWD_MODE") != nullptr;
    return !plain_modes;                                                        // A/B knobs select the unfused kernels
}
#endif

template <typename VT>
static int launch_fwd_prologue_t(const VT *value, const int64_t *shapes, const int64_t *level_start, const float *ref,
                                 const float *offsets, const float *logits, int N, int S, int M, int L, int Lq, int P,
                                 long long ld_offsets, long long ld_logits, VT *out, float *loc_out, float *attn_out,
                                 hipStream_t stream, PointEntry *table = nullptr)
{
    const int items = N * Lq * M, LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    const int split = pick_split(items, LP);
    const int ipw = 32 / split;
    const size_t lds = (size_t)ipw * item_stride + (split > 1 ? 4096 : 0);
    int hdr_W = 0, hdr_tp_cap = 0;
    if (table) table_header_plan(N, S, M, L, Lq, P, hdr_W, hdr_tp_cap);
    const dim3 grid((items + ipw - 1) / ipw + (hdr_W > 0 ? 1 : 0)), block(kBlock);      // (+ the header's workgroup)
    const PrologueIn pro{ref, loc_out, attn_out, (int)((ld_offsets - 2LL * M * LP) / 2), (int)(ld_logits - (long long)M * LP)};
    const LdsPlan lp = plan_lds<VT>(N, S, M, L, Lq, P);
    if (lp.use)
        return launch_fwd_lds<VT, true>(lp, value, shapes, level_start, offsets, logits, N, S, M, L, Lq, P, out, pro, stream,
                                        mask_out(table, N, S, M, L, Lq, P));
    const int xcd = xcd_remap();
#define MSDA_LAUNCH_FP(SP)                                                                             \
    hipLaunchKernelGGL((fwd_d32_kernel<SP, VT, true>), grid, block, lds, stream, value, shapes, level_start, offsets, \
                       logits, S, M, L, Lq, P, items, pow2_shift(P), pow2_shift(LP), pow2_shift(M), out, pro, xcd, table,   \
                       hdr_W, hdr_tp_cap, kSmallRecCap)
    if (split == 4) MSDA_LAUNCH_FP(4); else if (split == 2) MSDA_LAUNCH_FP(2); else MSDA_LAUNCH_FP(1);
#undef MSDA_LAUNCH_FP
    return check_launch("msda forward (d32, fused prologue)");
}

// VT = storage of grad_out / value; grad_value is fp32 for both (bf16 rows: nothing rounded between passes)
template <typename VT>
static int launch_bwd_prologue_t(const VT *grad_out, const VT *value, const int64_t *shapes, const int64_t *level_start,
                                 const float *loc, const float *attn, int N, int S, int M, int L, int Lq, int P, float *grad_value,
                                 long long ld_grad_offsets, long long ld_grad_logits, float *grad_offsets, float *grad_logits,
                                 float *grad_ref, hipStream_t stream, void *workspace, size_t ws_bytes, bool deterministic,
                                 const PointEntry *table = nullptr, bool no_dense = false)
{
    const int items = N * Lq * M, LP = L * P;
    const int item_stride = LP * kRecBytes + kItemPad;
    const int split = pick_split(items, LP);
    const int ps = pow2_shift(P), lps = pow2_shift(LP), ms = pow2_shift(M);
    const ValuePlan pl = plan_value<float>(N, S, M, L, Lq, P, bwd_target_wgs(), deterministic);
    const long long nB = (long long)pl.W * N * M * L;
    const FusedPlan fp = plan_fused(items, LP, split, nB, pl.acc, M, deterministic);
    const int ipw_f = kSWaves * 8 / fp.split;
    const size_t lds_a = (size_t)ipw_f * item_stride + (size_t)ipw_f * LP * 16;
    const long long nA = (items + ipw_f - 1) / ipw_f;
    if (pl.ppt != kSinglePPT || pl.acc == kAccTile || nB + nA > 0x7fffffffLL)
        return set_error(MSDA_ERR_ARGUMENT, "msda backward (fused prologue): geometry not supported");
    const PrologueOut pro{grad_ref, (int)((ld_grad_offsets - 2LL * M * LP) / 2), (int)(ld_grad_logits - (long long)M * LP)};
    // ---- large problems: role A on the LDS-stage body; it leaves grad_ref per head in the caller's scratch ----
    const LdsPlan lq = plan_lds<VT>(N, S, M, L, Lq, P, nB);
    const size_t heads_bytes = prologue_heads_bytes(N, M, L, Lq);
    if (lq.use && workspace != nullptr && ws_bytes >= heads_bytes && ((uintptr_t)workspace & 7) == 0 &&
        (pl.acc == kAccNone || pl.acc == kAccWide) && nB + (long long)N * M * lq.chunks <= 0x7fffffffLL) {
        const PrologueOut pro_h{static_cast<float *>(workspace), pro.off_pad, pro.log_pad};
        const dim3 lgrid((unsigned)(nB + (long long)N * M * lq.chunks));
        const size_t llds = max(max(pl.lds, lq.lds), (size_t)dense_lds_bytes());
#define MSDA_LAUNCH_BPL(AC, NS_) do { if (deterministic) MSDA_LAUNCH_BPL_(AC, NS_, true); else MSDA_LAUNCH_BPL_(AC, NS_, false); } while (0)
        const bool dense = dense_on() && pl.W <= MSDA_DENSE_MAX_W && !no_dense;
        const MaskIn masks = mask_in(table, N, S, M, L, Lq, P, pl.acc, pl.W, deterministic);
#define MSDA_LAUNCH_BPL_(AC, NS_, DT) do { if (dense) MSDA_LAUNCH_BPL__(AC, NS_, DT, true); else MSDA_LAUNCH_BPL__(AC, NS_, DT, false); } while (0)
#define MSDA_LAUNCH_BPL__(AC, NS_, DT, DN)                                                             \
        do { if (int rc = allow_lds(reinterpret_cast<const void *>(bwd_fused_lds_d32_kernel<AC, VT, true, float, NS_, DT, DN>), llds)) return rc; \
        hipLaunchKernelGGL((bwd_fused_lds_d32_kernel<AC, VT, true, float, NS_, DT, DN>), lgrid, dim3(kSBlock), llds, stream, grad_out,   \
                           value, shapes, level_start, loc, attn, S, M, L, Lq, P, ps, lps, pl.tp_cap, pl.W, (int)nB, lq.chunks,   \
                           lq.qw, lq.stage_rows, grad_value, grad_offsets, grad_logits, pro_h, xcd_remap(), (DN) ? (int)llds : 0, masks); } while (0)
        const bool one_slot = 8 * LP <= kWave;
        if (pl.acc == kAccNone) { if (one_slot) MSDA_LAUNCH_BPL(kAccNone, 1); else MSDA_LAUNCH_BPL(kAccNone, 2); }
        else                    { if (one_slot) MSDA_LAUNCH_BPL(kAccWide, 1); else MSDA_LAUNCH_BPL(kAccWide, 2); }
#undef MSDA_LAUNCH_BPL
#undef MSDA_LAUNCH_BPL_
#undef MSDA_LAUNCH_BPL__
        if (int rc = check_launch("msda backward (d32, fused prologue, LDS stage)")) return rc;
        const long long cells = (long long)N * Lq * L;
        hipLaunchKernelGGL(ref_heads_reduce_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream,
                           reinterpret_cast<const float2 *>(workspace), N * Lq, M, L, reinterpret_cast<float2 *>(grad_ref));
        return check_launch("msda backward (d32, reference-point gradient over heads)");
    }
    const dim3 fgrid((unsigned)(nB + nA));
    size_t flds = pl.lds > lds_a ? pl.lds : lds_a;
    if (fp.fixed) flds = max(flds, small_lds_bytes(pl.tp_cap, deterministic));
#define MSDA_LAUNCH_BP(SP, AC, FX) do { if (deterministic) MSDA_LAUNCH_BP_(SP, AC, FX, true); else MSDA_LAUNCH_BP_(SP, AC, FX, false); } while (0)
#define MSDA_LAUNCH_BP_(SP, AC, FX, DT)                                                                \
    do { if (int rc = allow_lds(reinterpret_cast<const void *>(bwd_fused_d32_kernel<SP, AC, VT, true, float, FX, DT>), flds)) return rc; \
    hipLaunchKernelGGL((bwd_fused_d32_kernel<SP, AC, VT, true, float, FX, DT>), fgrid, dim3(kSBlock), flds, stream, grad_out, \
                       value, shapes, level_start, loc, attn, S, M, L, Lq, P, items, ps, lps, ms, pl.tp_cap, pl.W, (int)nB,  \
                       grad_value, grad_offsets, grad_logits, pro, xcd_remap(), (FX) ? table : nullptr,                          \
                       (FX) ? table_header_of(table, N, M, L, Lq, P, pl.W) : nullptr); } while (0)
    if (fp.fixed)                { if (fp.split == 4) MSDA_LAUNCH_BP(4, kAccNone, true); else if (fp.split == 2) MSDA_LAUNCH_BP(2, kAccNone, true); else MSDA_LAUNCH_BP(1, kAccNone, true); }
    else if (pl.acc == kAccNone) { if (fp.split == 4) MSDA_LAUNCH_BP(4, kAccNone, false); else if (fp.split == 2) MSDA_LAUNCH_BP(2, kAccNone, false); else MSDA_LAUNCH_BP(1, kAccNone, false); }
    else if (pl.acc == kAccWide) { if (fp.split == 4) MSDA_LAUNCH_BP(4, kAccWide, false); else if (fp.split == 2) MSDA_LAUNCH_BP(2, kAccWide, false); else MSDA_LAUNCH_BP(1, kAccWide, false); }
#ifdef MSDA_TUNING
    else                         { if (fp.split == 4) MSDA_LAUNCH_BP(4, kAccRmw, false); else if (fp.split == 2) MSDA_LAUNCH_BP(2, kAccRmw, false); else MSDA_LAUNCH_BP(1, kAccRmw, false); }
#else
    else return set_error(MSDA_ERR_LAUNCH, "msda backward (d32, fused prologue): unplanned accumulation mode");
#endif
#undef MSDA_LAUNCH_BP
#undef MSDA_LAUNCH_BP_
    return check_launch("msda backward (d32, fused prologue)");
}

#if MSDA_D32_HAS(0)
int launch_fwd_prologue(const float *value, const int64_t *shapes, const int64_t *level_start, const float *ref,
                        const float *offsets, const float *logits, int N, int S, int M, int L, int Lq, int P,
                        long long ld_offsets, long long ld_logits, float *out, float *loc_out, float *attn_out,
                        hipStream_t stream, void *table)
{
    return launch_fwd_prologue_t<float>(value, shapes, level_start, ref, offsets, logits, N, S, M, L, Lq, P, ld_offsets, ld_logits,
                                        out, loc_out, attn_out, stream, static_cast<PointEntry *>(table));
}
#endif
#if MSDA_D32_HAS(1)
int launch_fwd_prologue_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *level_start, const float *ref,
                             const float *offsets, const float *logits, int N, int S, int M, int L, int Lq, int P,
                             long long ld_offsets, long long ld_logits, uint16_t *out, float *loc_out, float *attn_out,
                             hipStream_t stream, void *table)
{
    return launch_fwd_prologue_t<bf16_t>(value, shapes, level_start, ref, offsets, logits, N, S, M, L, Lq, P, ld_offsets, ld_logits,
                                         out, loc_out, attn_out, stream, static_cast<PointEntry *>(table));
}
#endif
#if MSDA_D32_HAS(0)
int launch_bwd_prologue(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *level_start,
                        const float *loc, const float *attn, int N, int S, int M, int L, int Lq, int P, float *grad_value,
                        long long ld_grad_offsets, long long ld_grad_logits, float *grad_offsets, float *grad_logits,
                        float *grad_ref, hipStream_t stream, void *workspace, size_t ws_bytes, bool deterministic, const void *table,
                        bool no_dense)
{
    return launch_bwd_prologue_t<float>(grad_out, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, grad_value,
                                        ld_grad_offsets, ld_grad_logits, grad_offsets, grad_logits, grad_ref, stream, workspace,
                                        ws_bytes, deterministic, static_cast<const PointEntry *>(table), no_dense);
}
#endif
#if MSDA_D32_HAS(2)
int launch_bwd_prologue_bf16(const uint16_t *grad_out, const uint16_t *value, const int64_t *shapes, const int64_t *level_start,
                             const float *loc, const float *attn, int N, int S, int M, int L, int Lq, int P, float *grad_value,
                             long long ld_grad_offsets, long long ld_grad_logits, float *grad_offsets, float *grad_logits,
                             float *grad_ref, hipStream_t stream, void *workspace, size_t ws_bytes, bool deterministic,
                             const void *table, bool no_dense)
{
    return launch_bwd_prologue_t<bf16_t>(grad_out, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, grad_value,
                                         ld_grad_offsets, ld_grad_logits, grad_offsets, grad_logits, grad_ref, stream, workspace,
                                         ws_bytes, deterministic, static_cast<const PointEntry *>(table), no_dense);
}
#endif

#if MSDA_D32_HAS(0)
int launch_fwd_d32(const float *value, const int64_t *shapes, const int64_t *level_start, const float *loc,
                   const float *attn, int N, int S, int M, int L, int Lq, int P, float *out, hipStream_t stream, void *table)
{
    return launch_fwd_d32_t<float>(value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, out, stream, static_cast<PointEntry *>(table));
}
#endif
#if MSDA_D32_HAS(0)
int launch_bwd_d32(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *level_start,
                   const float *loc, const float *attn, int N, int S, int M, int L, int Lq, int P,
                   float *grad_value, float *grad_loc, float *grad_attn, hipStream_t stream, void *workspace, size_t ws_bytes,
                   bool deterministic, const void *table, bool no_dense)
{
    return launch_bwd_d32_t<float>(grad_out, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, grad_value,
                                   grad_loc, grad_attn, stream, workspace, ws_bytes, deterministic, static_cast<const PointEntry *>(table), no_dense);
}
#endif
#if MSDA_D32_HAS(1)
int launch_fwd_d32_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *level_start, const float *loc,
                        const float *attn, int N, int S, int M, int L, int Lq, int P, uint16_t *out,
                        hipStream_t stream, void *table)
{
    return launch_fwd_d32_t<bf16_t>(value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, out, stream, static_cast<PointEntry *>(table));
}
#endif
#if MSDA_D32_HAS(1)
int launch_bwd_d32_bf16(const uint16_t *grad_out, const uint16_t *value, const int64_t *shapes,
                        const int64_t *level_start, const float *loc, const float *attn, int N, int S, int M, int L,
                        int Lq, int P, uint16_t *grad_value, float *grad_loc, float *grad_attn, hipStream_t stream,
                        void *workspace, size_t ws_bytes, bool deterministic, const void *table, bool no_dense)
{
    return launch_bwd_d32_t<bf16_t>(grad_out, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, grad_value,
                                    grad_loc, grad_attn, stream, workspace, ws_bytes, deterministic, static_cast<const PointEntry *>(table), no_dense);
}
#endif
#if MSDA_D32_HAS(2)
int launch_bwd_d32_bf16_gv32(const uint16_t *grad_out, const uint16_t *value, const int64_t *shapes,
                             const int64_t *level_start, const float *loc, const float *attn, int N, int S, int M, int L,
                             int Lq, int P, float *grad_value, float *grad_loc, float *grad_attn, hipStream_t stream,
                             void *workspace, size_t ws_bytes, bool deterministic, const void *table, bool no_dense)
{
    return launch_bwd_d32_t<bf16_t, float>(grad_out, value, shapes, level_start, loc, attn, N, S, M, L, Lq, P, grad_value,
                                           grad_loc, grad_attn, stream, workspace, ws_bytes, deterministic, static_cast<const PointEntry *>(table), no_dense);
}
#endif

#if MSDA_D32_HAS(0)
int backward_passes(int Lq, int P) { return (Lq * P + kSingleMaxPoints - 1) / kSingleMaxPoints; }

// What the launchers above would do for a geometry, as text (msda_describe_plan, include/msda.h): the same plan functions,
// nothing launched.  row_bytes = 4 (fp32 rows) or 2 (bf16 rows); gv_bytes the same for grad_value.
template <typename VT, typename GT>
static int describe_plan_t(int N, int S, int M, int L, int Lq, int P, bool prologue, bool has_ws, bool det, char *buf, int len, bool no_dense)
{
    const int items = N * Lq * M, LP = L * P;
    int n = 0;
    auto put = [&](const char *fmt, auto... a) { if (n < len) { const int k = snprintf(buf + n, (size_t)(len - n), fmt, a...); n += k > 0 ? k : 0; } };
    const LdsPlan lf = plan_lds<VT>(N, S, M, L, Lq, P);
    if (lf.use) put("fwd=lds(chunks=%d,qw=%d,stage_rows=%d,wgs=%d)", lf.chunks, lf.qw, lf.stage_rows, N * M * lf.chunks);
    else { const int sp = pick_split(items, LP); put("fwd=tiled(split=%d,wgs=%d)", sp, (items + 32 / sp - 1) / (32 / sp)); }
    const ValuePlan pl = plan_value<GT>(N, S, M, L, Lq, P, bwd_target_wgs(), det);
    const long long nB = (long long)pl.W * N * M * L;
    const char *acc = pl.acc == kAccNone ? "single" : pl.acc == kAccWide ? "wide" : pl.acc == kAccTile ? "tile" : "rmw";
    const LdsPlan la = plan_lds<VT>(N, S, M, L, Lq, P, nB);
    const bool lds_ok = la.use && pl.ppt == kSinglePPT && (pl.acc == kAccNone || pl.acc == kAccWide) && (!prologue || has_ws);
    if (lds_ok)
        // dense_px: a level of at most this many pixels (that the launch deals W workgroups) goes to the matrix cores
        // masks: role B finds its points through the per-point range masks a forward with a workspace leaves (msda_forward_ws_*)
        put(" bwd=fused_lds(acc=%s,W=%d,tp_cap=%d,roleB=%lld,roleA=%d,qw=%d,dense_px=%d%s%s%s)", acc, pl.W, pl.tp_cap, nB, N * M * la.chunks, la.qw,
            !dense_on() || no_dense || pl.W > MSDA_DENSE_MAX_W ? 0 : pl.W >= 2 ? kDenseMaxRows : kDensePassRows, det ? ",det" : "", prologue ? ",heads_reduce" : "",
            (!det && pl.acc == kAccWide && plan_masks(N, S, M, L, Lq, P).use && plan_masks(N, S, M, L, Lq, P).W == pl.W) ? ",masks" : "");
    else if (pl.ppt == kSinglePPT && pl.acc != kAccTile) {
        const FusedPlan fp = plan_fused(items, LP, pick_split(items, LP), nB, pl.acc, prologue ? M : 0, det);
        put(" bwd=fused(acc=%s,W=%d,tp_cap=%d,roleB=%lld,roleA=%d,split=%d,%s%s)", acc, pl.W, pl.tp_cap, nB,
            (items + 64 / fp.split - 1) / (64 / fp.split), fp.split, fp.fixed ? "fixed" : "prefix", det ? ",det" : "");
    } else
        put(" bwd=two_launches(acc=%s,W=%d,tp_cap=%d,roleB=%lld,roleA=%s%s)", acc, pl.W, pl.tp_cap, nB, la.use ? "lds" : "tiled", det ? ",det" : "");
    return n;
}
int describe_plan(int row_bytes, int gv_bytes, int N, int S, int M, int L, int Lq, int P, bool prologue, bool has_ws, bool det,
                  char *buf, int len, bool no_dense)
{
    if (row_bytes == 4) return describe_plan_t<float, float>(N, S, M, L, Lq, P, prologue, has_ws, det, buf, len, no_dense);
    if (gv_bytes == 4) return describe_plan_t<bf16_t, float>(N, S, M, L, Lq, P, prologue, has_ws, det, buf, len, no_dense);
    return describe_plan_t<bf16_t, bf16_t>(N, S, M, L, Lq, P, prologue, has_ws, det, buf, len, no_dense);
}
#endif

}  // namespace msda
