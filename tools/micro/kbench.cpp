// Standalone kernel harness for the D=32 kernels (diagnostic; not part of the library).
// Unity build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DMSDA_STAMPS \
//                 -DMSDA_TUNING -Iinclude -Iuvhand_amd/csrc tools/micro/kbench.cpp -o tools/micro/kbench
// Usage: kbench <workload: c2d|c2e|c4d|c4e> [iters]
// Prints HIP-event time per call of forward / backward and, for the grad_value kernel, the
// per-phase breakdown from in-kernel s_memrealtime stamps (100 MHz) of one extra launch.
#include "../../uvhand_amd/csrc/msda_abi.hip"
#include "../../uvhand_amd/csrc/msda_generic.hip"
#include "../../uvhand_amd/csrc/msda_d32.hip"
#include "../../uvhand_amd/csrc/msda_linear.hip"
#include "../../uvhand_amd/csrc/msda_gemm.hip"
#include "../../uvhand_amd/csrc/msda_layernorm.hip"
#include "../../uvhand_amd/csrc/msda_flatten.hip"
#include "../../uvhand_amd/csrc/msda_probe.hip"

#include <algorithm>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    const std::string wl = argc > 1 ? argv[1] : "c2d";
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    int N = 2, Lq = 300; std::vector<std::pair<int, int>> shp = {{48, 48}, {24, 24}, {12, 12}, {6, 6}};
    if (wl == "c2e") { Lq = 3060; }
    if (wl == "c4d" || wl == "c4e") { N = 32; shp = {{28, 28}, {14, 14}, {7, 7}, {4, 4}}; Lq = (wl == "c4d") ? 300 : 1045; }
    const int M = 8, D = 32, L = (int)shp.size(), P = 4;
    int S = 0; std::vector<int64_t> hs, hl;
    for (auto &s : shp) { hl.push_back(S); hs.push_back(s.first); hs.push_back(s.second); S += s.first * s.second; }
    const size_t nv = (size_t)N * S * M * D, nl = (size_t)N * Lq * M * L * P * 2, na = nl / 2, no = (size_t)N * Lq * M * D;
    std::mt19937 rng(1); std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float> hv(nv), hloc(nl), hat(na), hgo(no);
    for (auto &x : hv) x = U(rng) * 0.01f;
    for (auto &x : hloc) x = U(rng);
    for (auto &x : hat) x = U(rng) / 8.f + 1e-5f;
    for (auto &x : hgo) x = U(rng);
    float *v, *loc, *at, *go, *out, *gv, *gl, *ga; int64_t *ds, *dl;
    CK(hipMalloc(&v, nv * 4)); CK(hipMalloc(&loc, nl * 4)); CK(hipMalloc(&at, na * 4)); CK(hipMalloc(&go, no * 4));
    CK(hipMalloc(&out, no * 4)); CK(hipMalloc(&gv, nv * 4)); CK(hipMalloc(&gl, nl * 4)); CK(hipMalloc(&ga, na * 4));
    CK(hipMalloc(&ds, hs.size() * 8)); CK(hipMalloc(&dl, hl.size() * 8));
    CK(hipMemcpy(v, hv.data(), nv * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(loc, hloc.data(), nl * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(at, hat.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(go, hgo.data(), no * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), hs.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dl, hl.data(), hl.size() * 8, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // KB_TABLE=1: the forward leaves its point table and the backward reads it (small problems; what the autograd nodes do)
    const bool use_table = getenv("KB_TABLE") && atoi(getenv("KB_TABLE")) != 0;
    void *tab = nullptr; unsigned long long tab_bytes = use_table ? msda_forward_workspace_bytes(N, S, M, D, L, Lq, P, 0) : 0;
    if (tab_bytes) CK(hipMalloc(&tab, tab_bytes));
    printf("forward table: %.2f MB\n", tab_bytes / 1e6);
    auto fwd = [&] { return msda_forward_ws_f32(v, ds, dl, loc, at, N, S, M, D, L, Lq, P, out, tab, tab_bytes, st); };
    // KB_DET=1: the deterministic backward (per-wavefront counters; role B and role A as two launches)
    const bool det = getenv("KB_DET") && atoi(getenv("KB_DET")) != 0;
    void *ws = nullptr; unsigned long long ws_bytes = 0;
    if (!det) {                                      // default path: scratch for the level-major point table, if the shape uses one
        ws_bytes = msda_backward_workspace_bytes(N, S, M, D, L, Lq, P, 0);
        if (ws_bytes) CK(hipMalloc(&ws, ws_bytes));
        printf("default backward, workspace %.1f MB\n", ws_bytes / 1e6);
    }
    if (det) {
        ws_bytes = msda_backward_workspace_bytes(N, S, M, D, L, Lq, P, MSDA_FLAG_DETERMINISTIC);
        if (ws_bytes) CK(hipMalloc(&ws, ws_bytes));
        printf("deterministic backward, workspace %.1f MB\n", ws_bytes / 1e6);
    }
    auto bwd = [&] { return det ? msda_backward_ws_f32(go, v, ds, dl, loc, at, N, S, M, D, L, Lq, P, gv, gl, ga, ws, ws_bytes,
                                                       MSDA_FLAG_DETERMINISTIC, st)
                                : tab ? msda_backward_ws_f32(go, v, ds, dl, loc, at, N, S, M, D, L, Lq, P, gv, gl, ga, tab, tab_bytes, MSDA_FLAG_FORWARD_TABLE, st)
                                      : msda_backward_ws_f32(go, v, ds, dl, loc, at, N, S, M, D, L, Lq, P, gv, gl, ga, ws, ws_bytes, 0, st); };
    if (getenv("KB_SKIP_ROLE")) {                     // diagnostic: 1 = role B's workgroups exit at once, 2 = role A's (results are then incomplete)
        const int sk = atoi(getenv("KB_SKIP_ROLE"));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(msda::msda_skip_role), &sk, sizeof(sk)));
        printf("skipping role %s\n", sk == 1 ? "B" : sk == 2 ? "A" : "-");
    }
    if (getenv("KB_DIAG")) {                          // timing-only switches of the stamped build (results are then wrong): see msda_diag
        const int dg = atoi(getenv("KB_DIAG"));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(msda::msda_diag), &dg, sizeof(dg)));
        printf("diag flags %d\n", dg);
    }
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 5; ++i) if ((which ? bwd() : fwd()) != 0) { printf("launch failed: %s\n", msda_last_error()); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) which ? bwd() : fwd();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s %s: %.2f us per call (eager, %d calls)\n", wl.c_str(), which ? "bwd" : "fwd", ms * 1e3 / iters, iters);
    }
    // ---- stamps of one forward + one backward ----
    const size_t region = 65536 * 8, total = 3 * region;
    unsigned long long *sb; CK(hipMalloc(&sb, total * 8)); CK(hipMemset(sb, 0, total * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(msda::msda_stamp_buf), &sb, sizeof(sb)));
    fwd(); bwd(); CK(hipStreamSynchronize(st));
    unsigned long long *nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(msda::msda_stamp_buf), &nul, sizeof(nul)));
    std::vector<unsigned long long> hsb(total);
    CK(hipMemcpy(hsb.data(), sb, total * 8, hipMemcpyDeviceToHost));
    struct Reg { const char *name; int nph; const char *ph[5]; };
    const Reg regs[3] = {{"grad_value kernel (role B)", 5, {"loads+zero", "histogram", "prefix sum", "scatter", "gather"}},
                         {"query kernel (role A)", 3, {"prepass", "taps+dots", "write-out", "", ""}},
                         {"forward kernel", 3, {"prepass", "gather", "reduce+store", "", ""}}};
    for (int rg = 0; rg < 3; ++rg) {
        unsigned long long tmin = ~0ull, tmax = 0, smax = 0; size_t nb = 0;
        double ph[5] = {0, 0, 0, 0, 0}, phmax[5] = {0, 0, 0, 0, 0};
        const int last = regs[rg].nph;
        for (size_t b = 0; b < 65536; ++b) {
            const unsigned long long *t = &hsb[rg * region + b * 8];
            if (!t[0] || !t[last]) continue;
            ++nb; tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[last]); smax = std::max(smax, t[0]);
            for (int k = 0; k < last; ++k) { const double d = (double)(t[k + 1] - t[k]) * 0.01; ph[k] += d; phmax[k] = std::max(phmax[k], d); }
        }
        if (rg == 0 && getenv("KB_W")) {                    // role-B phase times by level (xcd numbering assumed on)
            const int W = atoi(getenv("KB_W")), nBb = W * N * M * L;
            const bool skew = !getenv("KB_SKEW") || atoi(getenv("KB_SKEW")) != 0;      // kAccWide deals W to every level
            double gs[16][5] = {{0}}, gm[16] = {0}; int gn[16] = {0};
            for (int b = 0; b < nBb && b < 65536; ++b) {
                const unsigned long long *t = &hsb[b * 8];
                if (!t[0] || !t[1] || !t[2] || !t[3] || !t[4] || !t[5]) continue;   // (a workgroup that had nothing to do stamps only its ends)
                const int x = b & 7, idx = b >> 3, q = nBb >> 3, r = nBb & 7, logical = x * q + (x < r ? x : r) + idx;
                // same dealing as value_block_to_range (msda_d32.hip) for a pyramid whose level 0 is the largest
                const int sl = logical % (W * L); int lvl = 0, base = 0;
                for (int k = 0; k < L; ++k) { const int wk = W + (skew && W >= 2 && L >= 2 ? (k == 0) - (k == L - 1) : 0); if (sl < base + wk) { lvl = k; break; } base += wk; }
                for (int k = 0; k < 5; ++k) gs[lvl][k] += (double)(t[k + 1] - t[k]) * 0.01;
                gm[lvl] = std::max(gm[lvl], (double)(t[5] - t[0]) * 0.01); ++gn[lvl];
            }
            for (int l = 0; l < L; ++l) if (gn[l]) printf("  level %d: phases %.2f %.2f %.2f %.2f %.2f us, lifetime max %.2f us (%d blocks)\n", l, gs[l][0] / gn[l],
                                                           gs[l][1] / gn[l], gs[l][2] / gn[l], gs[l][3] / gn[l], gs[l][4] / gn[l], gm[l], gn[l]);
        }
        if (!nb) continue;
        printf("%s: %zu sampled blocks, first start -> last end %.2f us, last block start +%.2f us\n", regs[rg].name, nb,
               (double)(tmax - tmin) * 0.01, (double)(smax - tmin) * 0.01);
        for (int k = 0; k < last; ++k) printf("  %-12s mean %.2f us  max %.2f us\n", regs[rg].ph[k], ph[k] / nb, phmax[k]);
    }
    return 0;
}
