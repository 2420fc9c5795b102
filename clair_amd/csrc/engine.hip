// C ABI of the MI355X engine (include/clair_amd.h): host side -- weight packing, workspaces,
// streams, launch sequence, timing.  gfx950 only; no CPU fallback: every entry point fails
// loudly when no HIP device is present.
#include "../../include/clair_amd.h"

#include <hip/hip_runtime.h>
#include <ctype.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.hip.h"
#include "dense.hip.h"
#include "decode.hip.h"
#include "gemm_split.hip.h"
#include "lstm32.hip.h"
#include "lstm32_pair.hip.h"
#include "lstm2_fused.hip.h"

using namespace clair;

namespace {

thread_local std::string g_create_error;
thread_local std::string *g_error_sink = nullptr;   // the staging worker: failures go to the slot they belong to, clair_wait reports them

const int64_t TENSOR_COUNT[CLAIR_T_COUNT] = {
    160 * 512, 512, 160 * 512, 512, 384 * 512, 512, 384 * 512, 512, 256 * 33 * 30, 256 * 30,
    7680 * 192, 192, 4 * 192 * 96, 4 * 96, 96 * 21, 21, 96 * 3, 3, 96 * 33, 33, 96 * 33, 33};

struct TimedLaunch {
    int kernel;
    hipEvent_t start, stop;
};

// A LANE is what one forward pass in flight needs: a HIP stream for its kernels and the inter-kernel workspaces.  Three or four lanes fill
// the chip (clair_engine_create: as many as the process has hardware queues to spare).
struct Lane {
    hipStream_t stream = nullptr;
    float *zx = nullptr;      // fragment-major x-projection, reused by both layers
    unsigned short *a1 = nullptr;   // [2][33][max_pad][256] fp16: LSTM1 output as its 2-way split
    float *a2 = nullptr;      // LSTM2 output, channel-group-major: [32 groups of 8 features][33][n_pad][8]
    float *l4part = nullptr;  // [8 splits][max_pad rounded to 64][192] split-K partials in the accumulator layout (dense.hip.h)
    unsigned *fuse_flags = nullptr;   // lstm2_fused.hip.h: [2][max_pad/32][33][8] ticket words, one error word, one claim word per workgroup
    unsigned fuse_ticket = 0;         // ticket of the last fused forward pass on this lane
    int last_n_pad = 0;
    std::mutex order;         // one forward pass is enqueued at a time (the submitting thread and the staging worker both enqueue)
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> free_events;
    // forward passes enqueued with the fused layer-2 launch since the lane's error word was last read: what recover_fused re-runs
    struct FusedRun { const float *x; float *out; int n; int slot; };   // slot: the submit it belongs to, -1 for clair_run_resident
    std::vector<FusedRun> fused_runs;
};

// A SLOT is one submit in flight at the host boundary: its input and output buffers on both sides of the link and a stream of its own for
// the transfers, so that the copy engines move batch i+1 in and batch i-1 out while the lanes compute batch i.  Slot s computes on lane
// s % lanes; the kernels of the slots that share a lane run back to back on the lane's stream (with as many slots as
// lanes every lane idles while its only slot is on the link -- 3.9 M candidates/s of 7.5).
struct Slot {
    int lane = 0;
    hipStream_t cin = nullptr, cout = nullptr;   // the streams the batch comes in on and the results go out on (copy_mode: whose they are)
    hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;   // input on the device / forward pass (and decode) finished / results on the host
    float *d_x = nullptr;     // [max_pad][1056]
    float *d_out = nullptr;   // [max_pad][90]
    float *h_out = nullptr;   // pinned [max_batch][90], then (h_word_offset) the fused launch's error word
    float *h_x = nullptr;     // pinned [max_batch][1056], allocated on first use
    short *d_counts = nullptr;   // [max_pad][1056] raw counts, allocated on first use
    char *d_records = nullptr;   // candidates copied with the caller's stride (binary tensor records as they lie), clair_submit_ex
    size_t d_records_bytes = 0;
    short *h_counts = nullptr;   // pinned [max_batch][1056]: staging of a caller's pageable count buffer (as h_x is for float input)
    // device decode (clair_submit_ex): the candidates' centre bytes in, call records out
    unsigned char *d_centre = nullptr, *h_centre = nullptr;   // [max_pad][2], pinned twin
    clair_call_t *d_calls = nullptr, *h_calls = nullptr;      // [max_pad], pinned twin
    clair_call_t *o_calls = nullptr;                          // caller's array of the pending submit (NULL: no decode requested)
    // pending host outputs of a submit
    float *o_gt21 = nullptr, *o_gt = nullptr, *o_l1 = nullptr, *o_l2 = nullptr;
    int pending_n = 0;
    bool refetch = false;     // a fused-launch recovery re-ran this slot's pass after its outputs had been fetched: fetch them again
    // hand-over to the staging worker: 0 = nothing queued, 1 = queued, 2 = enqueued on the device, 3 = failed (message in worker_error)
    int staged = 0;
    std::string worker_error;
};

}  // namespace

struct clair_engine {
    int device = 0;
    int max_batch = 0;
    int max_pad = 0;
    bool weights_ready = false;
    unsigned timing_mask = 0;   // bit k: kernel id k is bracketed by HIP events (clair_timing_enable)
    int lstm2_pair = -1;       // LSTM2 as two tiles per workgroup (lstm32_pair.hip.h): -1 = from 64 tiles (2048 candidates) on, where it wins
                               // 1-2 % (profiles/r02_lstm2_pair_by_batch.txt; at 1024 the kernel's own latency, 128 vs 81 us, costs 5 %); CLAIR_AMD_LSTM2_PAIR=0/1 forces
    std::atomic<int> lstm2_fused{0};       // layer 2 as ONE launch, projection and recurrence side by side (lstm2_fused.hip.h).  OPT-IN since round 5
                               // (CLAIR_AMD_LSTM2_FUSED=1): its zx hand-off is ordered by the L2 of ONE XCD -- the producer's stores retired into it, then
                               // its ticket -- which is how the hardware works and what every launch checks its placement for, but it is not a release
                               // the HIP memory model has a name for, and the release it does have (buffer_wbl2 per publication) costs more than the
                               // launch saves (0.5 us per write-back and XCD; even one per (direction, t) group leaves the launch
                               // slower than the two it replaces).  The default path of every handle is therefore the two launches, ordered by the
                               // stream.  What the fused launch bought on one- and two-slot handles: 104 us instead of 47 + 78 at batch 1024.
    int fused_groups = 4;      // projection workgroup groups per XCD inside the fused launch (CLAIR_AMD_FUSED_GROUPS)
    int64_t fused_launches = 0;   // fused launches of this handle so far
    int64_t fused_fault_at = 0;   // test hook CLAIR_AMD_FUSED_FAULT=k: the k-th fused launch finds logical id 0 already claimed (the kernel
                                  // raises its error word itself) and its a2 is poisoned afterwards, so only a real re-run gives right outputs
    int fused_recoveries = 0;     // passes re-run on the two-launch path after a fused launch raised its error word
    int proj2_groups = 8;      // persistent workgroup groups per XCD of the projection GEMM: 8 XCDs x 4 gate tiles x groups workgroups (see clair_engine_create)
    int w4_shift = 0;          // the W4 image is W4 * 2^w4_shift (clair_finalize_weights)
    int w3_shift = 0;          // the (W3 | b3) image is the tensor * 2^w3_shift
    bool l34_stamps = false;   // CLAIR_AMD_L34_STAMPS=1: l3l4_kernel writes its per-wave phase clocks into the (dead) zx workspace for clair_debug_read(5)
    bool tap_l3 = false;   // CLAIR_AMD_TAP_L3=1: l3l4_kernel also writes l3 into the (dead) zx workspace for clair_debug_read(4)
    std::string error;
    std::vector<std::pair<char *, size_t>> pinned;   // page-locked host buffers handed to the caller (clair_pinned_alloc)
    mutable std::mutex pinned_mu;                    // the staging workers look buffers up while the caller may allocate another
    std::vector<Slot> slots;
    std::vector<std::unique_ptr<Lane>> lanes;
    std::vector<hipStream_t> copy_streams;   // owned here; the slots point into it
    int copy_mode = 3;                  // CLAIR_AMD_COPY_STREAMS=slot|two|lane|in (0, 1, 2, 3): see clair_engine_create
    bool convert_on_lane = true;        // CLAIR_AMD_CONVERT=copy: the int16 -> float32 conversion behind the copy on the incoming stream instead of on the lane
    bool d2h_kernel = true;             // CLAIR_AMD_D2H=sdma: results fetched by the copy engine instead of written to page-locked host memory by a kernel
    // the staging worker (clair_submit* on pageable memory): the copy of the caller's batch into page-locked memory and the enqueue of its
    // transfers and kernels run on this thread, so that the submitting thread is free after a few microseconds, as the reference's
    // predict thread leaves load and output to two others (clair/call_var.py:1331-1352)
    struct Request {
        const void *input; bool counts; int64_t stride; int n;
        const uint8_t *centre; clair_call_t *calls; float *gt21, *gt, *l1, *l2;
    };
    int staging_threads = 2;            // CLAIR_AMD_STAGING_THREADS (0: everything on the submitting thread): a 4.3 MB batch takes one core ~100 us
                                        // to copy, 75 % of the 135 us the GPU needs for it
    std::vector<std::thread> workers;
    std::mutex wmu;
    std::condition_variable wcv, wdone;
    std::deque<std::pair<int, Request>> wqueue;
    bool wstop = false;
    std::vector<float> host_tensors[CLAIR_T_COUNT];
    // device weights
    float *bx1 = nullptr, *bx2 = nullptr;   // gate-scaled biases [2][512] of the two layers
    unsigned short *wh1s = nullptr, *wh2s = nullptr, *wx1s = nullptr, *w4s = nullptr, *w3s = nullptr;   // fp16 split MFMA fragment images (lstm32.hip.h, dense.hip.h)
    unsigned short *wx2s = nullptr;   // [8][2][1024][32] fp16 planes of the gate-scaled Wx2
    float *b4 = nullptr;
    unsigned short *w5s = nullptr, *whs = nullptr;   // fp16 split A fragments of the L5 branches and the heads (dense.hip.h: tail_kernel)
    float *b5 = nullptr, *bh = nullptr;
    int w5_shift[4] = {0, 0, 0, 0}, wh_shift[4] = {0, 0, 0, 0};   // per-branch power-of-two image shifts of those tensors
    double ms_sum[CLAIR_K_COUNT] = {0};
    int64_t launches[CLAIR_K_COUNT] = {0};
};

namespace {

int fail(clair_engine *e, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g_error_sink) *g_error_sink = buf; else if (e) e->error = buf; else g_create_error = buf;
    return 1;
}

#define HIP_TRY(e, call)                                                                        \
    do {                                                                                        \
        hipError_t err__ = (call);                                                              \
        if (err__ != hipSuccess)                                                                \
            return fail((e), "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
    } while (0)

int upload(clair_engine *e, float **dst, const std::vector<float> &src) {
    HIP_TRY(e, hipMalloc((void **)dst, src.size() * sizeof(float)));
    HIP_TRY(e, hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// host-side 2-way fp16 split (round to nearest even; _Float16 conversions are IEEE on the host compiler too)
inline unsigned short f16_bits(_Float16 h) { unsigned short u; memcpy(&u, &h, 2); return u; }
inline float f16_value(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
inline void split2_host(float x, unsigned short &hi, unsigned short &lo) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    hi = f16_bits(h);
    lo = f16_bits(l);
}

// Factor folded into every LSTM gate column (and bias) so the MFMA result is the exp2 argument of the
// gate's activation (lstm32.hip.h): columns are i | c~ | f | o, 128 each.
inline float gate_scale(int col512) {
    const float L2E = 1.44269504088896340736f;
    return ((col512 >> 7) == 1) ? 2.0f * L2E : -L2E;
}

// Gate-row order of the recurrent kernels (lstm32.hip.h): row rho = 8a + 4h' + c of block b of wave w is
// gate c (i | c~ | f | o) of hidden unit 32w + 8b + 4h' + a, i.e. column c*128 + unit of the reference's [K][512] kernel.
inline int gate_col(int w, int b, int rho) {
    const int a = rho >> 3, hq = (rho >> 2) & 1, c = rho & 3;
    return c * 128 + 32 * w + 8 * b + 4 * hq + a;
}

// fp16 2-way split A fragments of W^T for v_mfma_f32_32x32x16_f16: [dir][wave][b][kk][plane][lane][8]:
// W[k0 + 16*kk + 8*(lane/32) + j][gate_col(w, b, lane%32)] * gate_scale, kk < nkk
// `pow2` is an extra power-of-two factor on the image (exact): see L32_X_SHIFT in lstm32.hip.h
std::vector<unsigned short> pack_wt32(const std::vector<float> &fw, const std::vector<float> &bw, int k0, int nkk, float pow2 = 1.0f) {
    std::vector<unsigned short> out((size_t)2 * 4 * 4 * nkk * 2 * 64 * 8);
    for (int d = 0; d < 2; ++d) {
        const std::vector<float> &src = d ? bw : fw;
        for (int w = 0; w < 4; ++w)
            for (int b = 0; b < 4; ++b)
                for (int kk = 0; kk < nkk; ++kk)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int col = gate_col(w, b, lane & 31), k = k0 + 16 * kk + 8 * (lane >> 5) + j;
                            unsigned short hi, lo;
                            split2_host(src[(size_t)k * 512 + col] * gate_scale(col) * pow2, hi, lo);
                            const size_t base = (((((size_t)(d * 4 + w) * 4 + b) * nkk + kk) * 2) * 64 + lane) * 8 + j;
                            out[base] = hi;
                            out[base + 64 * 8] = lo;
                        }
    }
    return out;
}
// gate-scaled bias of both directions in gate-row order: [dir][wave][b][rho]  (= [..][a][h'][c] accumulator quads)
std::vector<float> pack_bias32(const std::vector<float> &fb, const std::vector<float> &bb) {
    std::vector<float> out(1024);
    for (int d = 0; d < 2; ++d)
        for (int w = 0; w < 4; ++w)
            for (int b = 0; b < 4; ++b)
                for (int rho = 0; rho < 32; ++rho) {
                    const int col = gate_col(w, b, rho);
                    out[((d * 4 + w) * 4 + b) * 32 + rho] = (d ? bb : fb)[col] * gate_scale(col);
                }
    return out;
}
int upload16(clair_engine *e, unsigned short **dst, const std::vector<unsigned short> &src) {
    (void)hipFree(*dst); *dst = nullptr;
    HIP_TRY(e, hipMalloc((void **)dst, src.size() * sizeof(unsigned short)));
    HIP_TRY(e, hipMemcpy(*dst, src.data(), src.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    return 0;
}

void free_lane(Lane &l) {
    if (l.stream) (void)hipStreamSynchronize(l.stream);
    for (auto &t : l.timed) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto ev : l.free_events) (void)hipEventDestroy(ev);
    (void)hipFree(l.zx); (void)hipFree(l.a1); (void)hipFree(l.a2); (void)hipFree(l.l4part); (void)hipFree(l.fuse_flags);
    if (l.stream) (void)hipStreamDestroy(l.stream);
}

void free_slot(Slot &s) {
    if (s.ev_out) (void)hipEventSynchronize(s.ev_out);
    if (s.ev_in) (void)hipEventDestroy(s.ev_in);
    if (s.ev_done) (void)hipEventDestroy(s.ev_done);
    if (s.ev_out) (void)hipEventDestroy(s.ev_out);
    (void)hipFree(s.d_x); (void)hipFree(s.d_out);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.h_x) (void)hipHostFree(s.h_x);
    if (s.d_counts) (void)hipFree(s.d_counts);
    if (s.d_records) (void)hipFree(s.d_records);
    if (s.h_counts) (void)hipHostFree(s.h_counts);
    if (s.d_centre) (void)hipFree(s.d_centre);
    if (s.h_centre) (void)hipHostFree(s.h_centre);
    if (s.d_calls) (void)hipFree(s.d_calls);
    if (s.h_calls) (void)hipHostFree(s.h_calls);
}

hipEvent_t get_event(Lane &s) {
    if (!s.free_events.empty()) {
        hipEvent_t ev = s.free_events.back();
        s.free_events.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}

struct KernelTimer {
    clair_engine *e; Lane &s; int id; hipEvent_t start = nullptr, stop = nullptr;
    KernelTimer(clair_engine *e_, Lane &s_, int id_) : e(e_), s(s_), id(id_) {
        if ((e->timing_mask >> id) & 1u) { start = get_event(s); stop = get_event(s); (void)hipEventRecord(start, s.stream); }
    }
    ~KernelTimer() {
        if (start) { (void)hipEventRecord(stop, s.stream); s.timed.push_back({id, start, stop}); }
    }
};

int quiesce(clair_engine *e);

int drain_timers(clair_engine *e) {
    if (quiesce(e)) return 1;           // nothing is being enqueued (and timed) while the lists are read
    for (auto &lp : e->lanes) {
        Lane &s = *lp;
        HIP_TRY(e, hipStreamSynchronize(s.stream));
        for (auto &t : s.timed) {
            float ms = 0.f;
            HIP_TRY(e, hipEventElapsedTime(&ms, t.start, t.stop));
            e->ms_sum[t.kernel] += ms;
            e->launches[t.kernel] += 1;
            s.free_events.push_back(t.start);
            s.free_events.push_back(t.stop);
        }
        s.timed.clear();
    }
    return 0;
}

bool fused_possible(const clair_engine *e) { return e->lstm2_fused == 1; }
bool use_lstm2_fused(const clair_engine *e, int ntiles) {   // pairs of tiles share a 64-row activation tile: even tile counts only
    if ((ntiles & 1) || !fused_possible(e)) return false;
    return e->lstm2_fused == 1 || (ntiles >= 16 && ntiles <= 64);   // 512 .. 2048 candidates: beyond, every kernel fills the chip by itself and
                                                                    // 128 projection workgroups are too few (batch 4096, one slot: 6.6 against 7.2 M/s)
}
size_t fuse_words(int max_pad) { return (size_t)2 * (max_pad / 32) * T_POS * 8; }                  // ticket words of the zx blocks
size_t fuse_claims(const clair_engine *e) { return (size_t)32 * e->fused_groups + 32 * ((e->max_pad / 64 + 7) / 8); }   // one per workgroup of the largest launch
bool use_lstm2_pair(const clair_engine *e, int ntiles) { return e->lstm2_pair < 0 ? ntiles >= 64 : e->lstm2_pair == 1; }

// Enqueue the forward pass for n candidates whose input is at x_dev ([n_pad][1056], rows >= n
// zero or any finite value) writing packed outputs to out_dev ([n][90]).
int enqueue_forward(clair_engine *e, Lane &s, const float *x_dev, float *out_dev, int n, int slot_index) {
    const int n_pad = (n + 31) & ~31;
    const int ntiles = n_pad / L32_TILE;
    const int m_rows = T_POS * n_pad;
    s.last_n_pad = n_pad;
    {   // LSTM1 with its input projection fused in (no separate GEMM, no zx round trip), fp16 split products
        KernelTimer kt(e, s, CLAIR_K_LSTM1);
        Lstm32Args a{x_dev, e->wx1s, e->bx1, nullptr, e->wh1s, s.a1, nullptr, n_pad, ntiles, -1};
        hipLaunchKernelGGL((lstm32_kernel<true>), dim3(ntiles * 2), dim3(256), 0, s.stream, a);
    }
    if (use_lstm2_fused(e, ntiles)) {   // layer 2 in one launch: projection and recurrence side by side, zx through L2 (lstm2_fused.hip.h)
        KernelTimer kt(e, s, CLAIR_K_LSTM2);
        const int groups = e->fused_groups;
        if (++s.fuse_ticket == 0) s.fuse_ticket = 1;   // (a wrapped ticket could meet a 4-billion-passes-old word; the words are zero at most once)
        Lstm2FusedArgs a{GemmSplitArgs{s.a1, e->wx2s, e->bx2, s.zx, n_pad, ntiles, m_rows, groups},
                         Lstm32Args{nullptr, nullptr, nullptr, s.zx, e->wh2s, nullptr, s.a2, n_pad, ntiles, -1},
                         FuseArgs{s.fuse_flags, s.fuse_ticket, s.fuse_flags + fuse_words(e->max_pad) + 1, s.fuse_flags + fuse_words(e->max_pad)}, 32 * groups};
        const int consumers = 32 * ((ntiles / 2 + 7) / 8);
        const bool fault = ++e->fused_launches == e->fused_fault_at;
        if (fault) HIP_TRY(e, hipMemsetD32Async((hipDeviceptr_t)a.f.claims, (int)s.fuse_ticket, 1, s.stream));
        hipLaunchKernelGGL(lstm2_fused_kernel, dim3(32 * groups + consumers), dim3(256), 0, s.stream, a);
        if (fault) HIP_TRY(e, hipMemsetAsync(s.a2, 0x7f, (size_t)T_POS * n_pad * 256 * sizeof(float), s.stream));
        s.fused_runs.push_back({x_dev, out_dev, n, slot_index});
    } else {
        {   // LSTM2 input projection on the fp16 matrix cores, fp32-grade via the 2-way split; weight-stationary persistent workgroups
            KernelTimer kt(e, s, CLAIR_K_PROJ2);
            const int x_tiles = (m_rows + GS_ROWS - 1) / GS_ROWS;
            const int groups = std::min(e->proj2_groups, (x_tiles + 7) / 8);
            GemmSplitArgs a{s.a1, e->wx2s, e->bx2, s.zx, n_pad, ntiles, m_rows, groups};
            hipLaunchKernelGGL(gemm_split_kernel, dim3(32 * groups), dim3(256), 0, s.stream, a);
        }
        {
            KernelTimer kt(e, s, CLAIR_K_LSTM2);
            if (use_lstm2_pair(e, ntiles)) {
                Lstm32PairArgs a{s.zx, e->wh2s, s.a2, n_pad, ntiles};
                hipLaunchKernelGGL(lstm32_pair_kernel, dim3(((ntiles + 1) / 2) * 2), dim3(256), 0, s.stream, a);
            } else {
                Lstm32Args a{nullptr, nullptr, nullptr, s.zx, e->wh2s, nullptr, s.a2, n_pad, ntiles, -1};
                hipLaunchKernelGGL((lstm32_kernel<false>), dim3(ntiles * 2), dim3(256), 0, s.stream, a);
            }
        }
    }
    {   // L3 (slice dense) + L4 (split-K 8: a workgroup walks the four channel groups of its split), fused
        KernelTimer kt(e, s, CLAIR_K_L4);
        L3L4Args a{s.a2, e->w3s, e->w4s, s.l4part, n_pad, std::ldexp(1.0f, -e->w3_shift), e->tap_l3 ? s.zx : nullptr,
                   e->l34_stamps ? (unsigned long long *)s.zx : nullptr};   // zx is dead by now
        hipLaunchKernelGGL(l3l4_kernel, dim3(l34_grid((n_pad + L34_CAND - 1) / L34_CAND)), dim3(L34_THREADS), 0, s.stream, a);
    }
    {
        KernelTimer kt(e, s, CLAIR_K_TAIL);
        TailArgs a{s.l4part, e->b4, e->w5s, e->b5, e->whs, e->bh, out_dev, n_pad, n, std::ldexp(1.0f, -e->w4_shift) / L34_ACT_SCALE, {}, {}};
        for (int k = 0; k < 4; ++k) {
            a.l5_scale[k] = std::ldexp(1.0f, -e->w5_shift[k]) / TAIL_ACT_SCALE;
            a.head_scale[k] = std::ldexp(1.0f, -e->wh_shift[k]) / TAIL_ACT_SCALE;
        }
        hipLaunchKernelGGL(tail_kernel, dim3(n_pad / TAIL_TILE), dim3(TAIL_THREADS), 0, s.stream, a);
    }
    HIP_TRY(e, hipGetLastError());
    return 0;
}

// A slot's page-locked input buffer exists from its first use -- by a staging worker (pageable float input) or by the caller
// (clair_slot_input), whichever comes first: one allocation, under the lock the look-ups below take.
int ensure_slot_input(clair_engine *e, Slot &s) {
    std::lock_guard<std::mutex> g(e->pinned_mu);
    if (!s.h_x) HIP_TRY(e, hipHostMalloc((void **)&s.h_x, (size_t)e->max_batch * CLAIR_INPUT_FLOATS * sizeof(float), hipHostMallocDefault));
    return 0;
}

// does [p, p + len) lie inside a buffer of clair_pinned_alloc?  (then the DMA engine can read it directly)
bool in_pinned(const clair_engine *e, const void *p, size_t len) {
    std::lock_guard<std::mutex> g(e->pinned_mu);
    for (const auto &b : e->pinned)
        if ((const char *)p >= b.first && (const char *)p + len <= b.first + b.second) return true;
    return false;
}

// n candidates of `row` bytes each, `stride` bytes apart in the caller's buffer (0: dense), into a dense staging buffer
// memory of this process's HIP devices (hipMalloc), as opposed to anything the host allocated
bool is_device_pointer(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice;
}

void gather_rows(void *dst, const void *src, int n, size_t row, int64_t stride) {
    if (stride == 0 || (size_t)stride == row) { memcpy(dst, src, (size_t)n * row); return; }
    for (int i = 0; i < n; ++i) memcpy((char *)dst + (size_t)i * row, (const char *)src + (size_t)i * (size_t)stride, row);
}

// The decode of the slot's batch on the device (decode.hip.h): probabilities in d_out + window in d_x + centre bytes -> call records.
int enqueue_decode(clair_engine *e, Lane &l, Slot &s, int n) {
    KernelTimer kt(e, l, CLAIR_K_DECODE);
    DecodeArgs a{s.d_x, s.d_out, s.d_centre, s.d_calls, n};
    hipLaunchKernelGGL(decode_kernel, dim3((n + 3) / 4), dim3(256), 0, l.stream, a);
    HIP_TRY(e, hipGetLastError());
    return 0;
}

// lstm2_fused.hip.h hands zx over through the L2 of the XCD its workgroups find themselves on; if the placement rule it relies on
// did not hold (a logical id claimed twice, a wait that ran out) a workgroup raised the word behind the slot's tickets and the
// results of the slot's fused passes cannot be trusted.  The reference never drops a batch (clair/call_var.py:1331-1352), so
// neither does this: the fused launch is switched off for the rest of the handle's life, the affected passes are enqueued again
// on the two-launch path (same arithmetic, bit-identical outputs) and the caller sees success; stderr gets one line.
// Called with the lane's order lock held; waits for the lane first.  Slots other than `current` whose pass was re-run fetch their
// outputs again when they are waited for (Slot::refetch).
int recover_fused(clair_engine *e, Lane &l, int current) {
    HIP_TRY(e, hipStreamSynchronize(l.stream));
    HIP_TRY(e, hipMemsetAsync(l.fuse_flags + fuse_words(e->max_pad), 0, sizeof(unsigned), l.stream));
    if (e->lstm2_fused != 0)
        fprintf(stderr, "clair_amd: the fused layer-2 launch found its blocks placed differently from what it assumes (a logical id claimed twice, or "
                        "a bounded wait that ran out); re-running %d pass(es) on the two-launch path and keeping it for this handle\n", (int)l.fused_runs.size());
    e->lstm2_fused = 0;
    std::vector<Lane::FusedRun> runs;
    runs.swap(l.fused_runs);
    for (const auto &r : runs) {
        if (enqueue_forward(e, l, r.x, r.out, r.n, r.slot)) return 1;
        if (r.slot >= 0 && r.slot != current) e->slots[r.slot].refetch = true;
        ++e->fused_recoveries;
    }
    HIP_TRY(e, hipStreamSynchronize(l.stream));
    return 0;
}

int check_fused_placement(clair_engine *e) {
    for (auto &lp : e->lanes) {
        Lane &l = *lp;
        if (!l.fuse_flags) continue;
        std::lock_guard<std::mutex> g(l.order);
        if (!l.fused_runs.empty()) {
            unsigned bad = 0;
            HIP_TRY(e, hipMemcpy(&bad, l.fuse_flags + fuse_words(e->max_pad), sizeof bad, hipMemcpyDeviceToHost));
            if (bad && recover_fused(e, l, -1)) return 1;
        }
        l.fused_runs.clear();
    }
    return 0;
}

// raw counts -> network input (clair/utils.py:96-98): one (position, row) quad of four channels per thread
__global__ __launch_bounds__(256) void counts_to_input_kernel(const short4 *counts, f32x4 *x, int n_quads) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_quads) return;
    const short4 c = counts[i];
    const float c0 = (float)c.x;
    x[i] = (f32x4){c0, (float)c.y - c0, (float)c.z - c0, (float)c.w - c0};
}

// the same for candidates `stride` bytes apart (binary tensor records copied as they lie): quad i = candidate i / 264, quad i % 264 of it
__global__ __launch_bounds__(256) void counts_to_input_strided_kernel(const char *base, size_t stride, f32x4 *x, int n_quads) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_quads) return;
    const int cand = i / (CLAIR_INPUT_FLOATS / 4), q = i - cand * (CLAIR_INPUT_FLOATS / 4);
    const short4 c = *(const short4 *)(base + (size_t)cand * stride + (size_t)q * 8);
    const float c0 = (float)c.x;
    x[i] = (f32x4){c0, (float)c.y - c0, (float)c.z - c0, (float)c.w - c0};
}


// Where the fused launch's error word lives in a slot's page-locked output buffer, in floats: the first 16-byte boundary past the
// last uint4 the result kernel below may write for a full batch (it copies the probabilities as WHOLE vectors: for an odd n the last
// one reaches 8 bytes past n * 360), so that neither that vector nor the end of the allocation can touch the word.
static inline size_t h_word_offset(int max_batch) { return ((((size_t)max_batch * OUT_FLOATS * sizeof(float) + 15) / 16) * 16) / sizeof(float); }
static inline size_t h_out_floats(int max_batch) { return h_word_offset(max_batch) + 4; }

// results of a forward pass (and decode) to the slot's page-locked buffers, written by the GPU itself: no copy engine, no queue switch
__global__ __launch_bounds__(256) void results_to_host_kernel(const uint4 *out, uint4 *h_out, int out_vec, const uint4 *calls, uint4 *h_calls, int call_vec,
                                                             const unsigned *word, unsigned *h_word) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < out_vec) h_out[i] = out[i];
    else if (i - out_vec < call_vec) h_calls[i - out_vec] = calls[i - out_vec];
    if (i == 0 && word) *h_word = *word;
}

// The way back: call records, probabilities and the fused launch's error word into the slot's page-locked buffers, on the slot's
// outgoing stream (which may be the lane's own), then the event clair_wait sleeps on.
int enqueue_results(clair_engine *e, Lane &l, Slot &s, int n, bool calls, bool probs) {
    unsigned *word = l.fuse_flags ? l.fuse_flags + fuse_words(e->max_pad) : nullptr;
    unsigned *h_word = (unsigned *)(s.h_out + h_word_offset(e->max_batch));
    if (e->d2h_kernel) {
        const int out_vec = probs ? (n * OUT_FLOATS * (int)sizeof(float) + 15) / 16 : 0, call_vec = calls ? n * (int)sizeof(clair_call_t) / 16 : 0;
        hipLaunchKernelGGL(results_to_host_kernel, dim3((std::max(out_vec + call_vec, 1) + 255) / 256), dim3(256), 0, s.cout, (const uint4 *)s.d_out, (uint4 *)s.h_out, out_vec,
                           (const uint4 *)s.d_calls, (uint4 *)s.h_calls, call_vec, word, h_word);
    } else {
        if (calls) HIP_TRY(e, hipMemcpyAsync(s.h_calls, s.d_calls, (size_t)n * sizeof(clair_call_t), hipMemcpyDeviceToHost, s.cout));
        if (probs) HIP_TRY(e, hipMemcpyAsync(s.h_out, s.d_out, (size_t)n * OUT_FLOATS * sizeof(float), hipMemcpyDeviceToHost, s.cout));
        if (word) HIP_TRY(e, hipMemcpyAsync(h_word, word, sizeof(unsigned), hipMemcpyDeviceToHost, s.cout));
    }
    HIP_TRY(e, hipEventRecord(s.ev_out, s.cout));
    return 0;
}

// Everything one submit puts on the device, in order: the batch over the link on the slot's copy stream (through page-locked staging
// when the caller's memory is pageable), the forward pass (and the decode) on the slot's lane once the batch has arrived, the results
// back over the link once the lane is done.  Runs on the submitting thread or on the staging worker.
int enqueue_request(clair_engine *e, int slot_index, const clair_engine::Request &q) {
    Slot &s = e->slots[slot_index];
    Lane &l = *e->lanes[s.lane];
    const int n = q.n, n_pad = (n + 31) & ~31;
    const bool want_probs = q.gt21 != nullptr;
    const bool same_in = s.cin == l.stream, same_out = s.cout == l.stream;     // copy_mode "lane": copies in line with the kernels, no events
    std::unique_lock<std::mutex> whole(l.order, std::defer_lock);
    if (same_in) whole.lock();
    // A caller's buffer from clair_pinned_alloc (or the slot's own input buffer) is read by the DMA engine where it lies: no pass over
    // the batch on any host thread.  Anything else goes through the slot's page-locked staging buffer.
    const size_t row_bytes = CLAIR_INPUT_FLOATS * (q.counts ? sizeof(short) : sizeof(float));
    const size_t stride = q.stride ? (size_t)q.stride : row_bytes;
    const size_t span = (size_t)(n - 1) * stride + row_bytes;
    const float *own_input;
    { std::lock_guard<std::mutex> g(e->pinned_mu); own_input = s.h_x; }
    const bool direct = (q.input == (const void *)own_input && own_input && stride == row_bytes && !q.counts) || in_pinned(e, q.input, span);
    const bool on_device = !direct && is_device_pointer(q.input);   // e.g. the windows of clair_frontend_build_windows: no copy at all
    if (on_device && !q.counts) return fail(e, "a device pointer is taken for int16 counts only");
    enum { NONE, DENSE, STRIDED } convert = NONE;                    // the int16 -> float32 kernel the lane runs first
    const char *convert_from = nullptr;
    if (q.counts) {
        if (!s.d_counts && !on_device) HIP_TRY(e, hipMalloc((void **)&s.d_counts, (size_t)e->max_pad * CLAIR_INPUT_FLOATS * sizeof(short)));
        if (on_device) {
            convert = STRIDED; convert_from = (const char *)q.input;
        } else if (direct && stride != row_bytes) {   // records as they lie: ONE contiguous copy of the span, the conversion kernel skips what lies between the counts
            if (s.d_records_bytes < span) {
                (void)hipFree(s.d_records); s.d_records = nullptr; s.d_records_bytes = 0;
                const size_t want = std::max(span, (size_t)e->max_batch * stride);
                HIP_TRY(e, hipMalloc((void **)&s.d_records, want));
                s.d_records_bytes = want;
            }
            HIP_TRY(e, hipMemcpyAsync(s.d_records, q.input, span, hipMemcpyHostToDevice, s.cin));
            convert = STRIDED; convert_from = s.d_records;
        } else {
            const void *src = q.input;
            if (!direct) {
                if (!s.h_counts) HIP_TRY(e, hipHostMalloc((void **)&s.h_counts, (size_t)e->max_batch * CLAIR_INPUT_FLOATS * sizeof(short), hipHostMallocDefault));
                gather_rows(s.h_counts, q.input, n, row_bytes, q.stride);
                src = s.h_counts;
            }
            HIP_TRY(e, hipMemcpyAsync(s.d_counts, src, (size_t)n * row_bytes, hipMemcpyHostToDevice, s.cin));
            convert = DENSE; convert_from = (const char *)s.d_counts;
        }
    } else if (direct && stride != row_bytes) {
        HIP_TRY(e, hipMemcpy2DAsync(s.d_x, row_bytes, q.input, stride, row_bytes, (size_t)n, hipMemcpyHostToDevice, s.cin));
    } else {
        const void *src = q.input;
        if (!direct) {
            if (ensure_slot_input(e, s)) return 1;
            gather_rows(s.h_x, q.input, n, row_bytes, q.stride);
            src = s.h_x;
        }
        HIP_TRY(e, hipMemcpyAsync(s.d_x, src, (size_t)n * row_bytes, hipMemcpyHostToDevice, s.cin));
    }
    if (n_pad > n)
        HIP_TRY(e, hipMemsetAsync(s.d_x + (size_t)n * CLAIR_INPUT_FLOATS, 0, (size_t)(n_pad - n) * CLAIR_INPUT_FLOATS * sizeof(float), s.cin));
    if (q.calls) {
        if (!s.d_centre) {
            HIP_TRY(e, hipMalloc((void **)&s.d_centre, (size_t)e->max_pad * 2));
            HIP_TRY(e, hipHostMalloc((void **)&s.h_centre, (size_t)e->max_batch * 2, hipHostMallocDefault));
            HIP_TRY(e, hipMalloc((void **)&s.d_calls, (size_t)e->max_pad * sizeof(clair_call_t)));
            HIP_TRY(e, hipHostMalloc((void **)&s.h_calls, (size_t)e->max_batch * sizeof(clair_call_t), hipHostMallocDefault));
        }
        memcpy(s.h_centre, q.centre, (size_t)n * 2);
        HIP_TRY(e, hipMemcpyAsync(s.d_centre, s.h_centre, (size_t)n * 2, hipMemcpyHostToDevice, s.cin));
    }
    // int16 counts -> the float32 tensor.  On the LANE, in front of LSTM1 (default): a kernel of 1 056 small workgroups on the incoming stream has to
    // find CUs of its own among lanes whose recurrent workgroups hold whole CUs (one wave per SIMD, every register), and now and then it waits for
    // them long enough to leave a lane without input -- 5.3-5.5 instead of 7.2-7.6 M candidates/s in one run out of four (profiles/r04_convert_stream.txt);
    // on the lane it costs its own ~4 us per pass and nothing else.  CLAIR_AMD_CONVERT=copy puts it back behind the copy.
    auto launch_convert = [&](hipStream_t st) {
        const int n_quads = n * (CLAIR_INPUT_FLOATS / 4);
        if (convert == DENSE)
            hipLaunchKernelGGL(counts_to_input_kernel, dim3((n_quads + 255) / 256), dim3(256), 0, st, (const short4 *)convert_from, (f32x4 *)s.d_x, n_quads);
        else if (convert == STRIDED)
            hipLaunchKernelGGL(counts_to_input_strided_kernel, dim3((n_quads + 255) / 256), dim3(256), 0, st, convert_from, stride, (f32x4 *)s.d_x, n_quads);
    };
    const bool convert_on_lane = e->convert_on_lane && !same_in;
    if (!convert_on_lane) launch_convert(s.cin);
    if (!same_in) HIP_TRY(e, hipEventRecord(s.ev_in, s.cin));
    {
        std::unique_lock<std::mutex> g(l.order, std::defer_lock);
        if (!same_in) g.lock();
        if (!same_in) HIP_TRY(e, hipStreamWaitEvent(l.stream, s.ev_in, 0));
        if (convert_on_lane) launch_convert(l.stream);
        if (enqueue_forward(e, l, s.d_x, s.d_out, n, slot_index)) return 1;
        if (q.calls && enqueue_decode(e, l, s, n)) return 1;
        if (same_out) return enqueue_results(e, l, s, n, q.calls != nullptr, want_probs);      // in line with the kernels, under the lane's lock
        HIP_TRY(e, hipEventRecord(s.ev_done, l.stream));
    }
    HIP_TRY(e, hipStreamWaitEvent(s.cout, s.ev_done, 0));
    return enqueue_results(e, l, s, n, q.calls != nullptr, want_probs);
}

void staging_worker(clair_engine *e) {
    (void)hipSetDevice(e->device);
    for (;;) {
        std::pair<int, clair_engine::Request> job;
        {
            std::unique_lock<std::mutex> g(e->wmu);
            e->wcv.wait(g, [&] { return e->wstop || !e->wqueue.empty(); });
            if (e->wqueue.empty()) return;            // stop asked for and nothing left to do
            job = e->wqueue.front();
            e->wqueue.pop_front();
        }
        g_error_sink = &e->slots[job.first].worker_error;   // `error` belongs to the caller's threads
        const int rc = enqueue_request(e, job.first, job.second);
        g_error_sink = nullptr;
        {
            std::lock_guard<std::mutex> g(e->wmu);
            e->slots[job.first].staged = rc ? 3 : 2;
        }
        e->wdone.notify_all();
    }
}

// Validates a request on the caller's thread, then hands it to the staging worker when the caller's memory has to be copied first
// (the worker does the copy AND the enqueue) or enqueues it right here.
int submit_request(clair_engine *e, int slot, const clair_engine::Request &q) {
    HIP_TRY(e, hipSetDevice(e->device));
    Slot &s = e->slots[slot];
    if (s.pending_n) return fail(e, "slot %d still has a pending submit; call clair_wait first", slot);
    s.o_gt21 = q.gt21; s.o_gt = q.gt; s.o_l1 = q.l1; s.o_l2 = q.l2;
    s.o_calls = q.calls;
    s.refetch = false;
    if (!e->workers.empty()) {
        {
            std::lock_guard<std::mutex> g(e->wmu);
            s.staged = 1;
            e->wqueue.emplace_back(slot, q);
        }
        e->wcv.notify_one();
    } else {
        s.staged = 0;
        if (enqueue_request(e, slot, q)) return 1;
    }
    s.pending_n = q.n;
    return 0;
}

// the staging worker has nothing queued and every stream of the handle is idle
int quiesce(clair_engine *e) {
    {
        std::unique_lock<std::mutex> g(e->wmu);
        e->wdone.wait(g, [&] { for (auto &s : e->slots) if (s.staged == 1) return false; return true; });
    }
    for (auto st : e->copy_streams) HIP_TRY(e, hipStreamSynchronize(st));
    for (auto &lp : e->lanes) HIP_TRY(e, hipStreamSynchronize(lp->stream));
    return 0;
}

int check_slot(clair_engine *e, int slot) {
    if (!e) return fail(nullptr, "engine is NULL");
    if (slot < 0 || slot >= (int)e->slots.size()) return fail(e, "slot %d out of range [0,%d)", slot, (int)e->slots.size());
    if (!e->weights_ready) return fail(e, "weights not loaded: call clair_set_tensor for all tensors, then clair_finalize_weights");
    return 0;
}

}  // namespace

extern "C" {

int clair_abi_version(void) { return CLAIR_ABI_VERSION; }

int clair_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int clair_device_pci_bus_id(int device, char *buf, int len) {
    if (!buf || len < 16) return fail(nullptr, "clair_device_pci_bus_id: buffer of at least 16 bytes needed");
    buf[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(nullptr, "clair_device_pci_bus_id: no HIP device %d (%d visible)", device, n);
    if (hipDeviceGetPCIBusId(buf, len, device) != hipSuccess) { buf[0] = 0; return fail(nullptr, "hipDeviceGetPCIBusId(%d) failed", device); }
    for (char *c = buf; *c; ++c) *c = (char)tolower((unsigned char)*c);      // sysfs spells the hexadecimal digits in lower case
    return 0;
}

const char *clair_last_error(const clair_engine_t *e) { return e ? e->error.c_str() : g_create_error.c_str(); }

int clair_engine_create(int device, int max_batch, int n_slots, clair_engine_t **out) {
    if (!out) return fail(nullptr, "out is NULL");
    *out = nullptr;
    if (max_batch < 1 || max_batch > (1 << 20)) return fail(nullptr, "max_batch %d out of range [1, 2^20]", max_batch);
    if (n_slots < 1 || n_slots > 64) return fail(nullptr, "n_slots %d out of range [1,64]", n_slots);
    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev <= 0)
        return fail(nullptr, "no HIP device available (hipGetDeviceCount: %s); the MI355X engine has no CPU fallback",
                    hipGetErrorString(err));
    if (device < 0 || device >= ndev) return fail(nullptr, "device %d out of range [0,%d)", device, ndev);
    HIP_TRY(nullptr, hipSetDevice(device));
    clair_engine *e = new clair_engine();
    e->device = device;
    e->max_batch = max_batch;
    e->max_pad = (max_batch + 31) & ~31;
    { const char *t = getenv("CLAIR_AMD_TAP_L3"); e->tap_l3 = t && t[0] == '1'; }
    { const char *t = getenv("CLAIR_AMD_L34_STAMPS"); e->l34_stamps = !e->tap_l3 && t && t[0] == '1'; }
    // A handle with one slot runs its kernels alone: the projection GEMM takes every CU (8 XCDs x 4 gate tiles x 8 groups).  With
    // batches in flight on several slots the 64-workgroup recurrent kernels of the other slots hold whole CUs for ~80 us; a
    // 256-workgroup persistent GEMM then runs its last 64 workgroups as a second round on a quarter of the chip.  Four groups (128
    // workgroups) pack beside two recurrent kernels: +4 % whole-pipeline throughput at 3 slots (profiles/r01_microbench.txt).
    // Lanes.  The runtime gives a process four hardware queues: FOUR forward passes in flight when nothing else needs one (a handle of up to four
    // slots: clair_run_resident, clair_predict -- 8.05 against 7.89 M candidates/s with three, profiles/r04_lanes_sweep.txt; a fifth stream shares a
    // queue with one of the four and sets the whole pipeline back, 6.5 M/s); THREE plus the incoming copy stream for a handle with more slots than
    // that, i.e. one that is fed from the host with batches in flight on the link (four lanes there: 6.2 against 7.3 M/s).
    int n_lanes = n_slots <= 4 ? n_slots : 3;
    { const char *t = getenv("CLAIR_AMD_LANES"); if (t && atoi(t) > 0) n_lanes = std::min(atoi(t), n_slots); }
    e->proj2_groups = n_lanes >= 4 ? 3 : (n_lanes > 1 ? 4 : 8);       // four lanes: 96 workgroups (8.00 against 7.91 M/s with 128, profiles/r04_lanes_sweep.txt)
    { const char *t = getenv("CLAIR_AMD_PROJ2_GROUPS"); if (t && atoi(t) > 0) e->proj2_groups = atoi(t); }
    { const char *t = getenv("CLAIR_AMD_LSTM2_FUSED"); if (t && (t[0] == '0' || t[0] == '1')) e->lstm2_fused = t[0] - '0'; }
    { const char *t = getenv("CLAIR_AMD_FUSED_GROUPS"); if (t && atoi(t) > 0 && atoi(t) <= 8) e->fused_groups = atoi(t); }
    { const char *t = getenv("CLAIR_AMD_LSTM2_PAIR"); if (t && (t[0] == '0' || t[0] == '1')) e->lstm2_pair = t[0] - '0'; }
    { const char *t = getenv("CLAIR_AMD_FUSED_FAULT"); if (t && atoll(t) > 0) e->fused_fault_at = atoll(t); }
    { const char *t = getenv("CLAIR_AMD_STAGING_THREADS"); if (t && atoi(t) >= 0 && atoi(t) <= 16) e->staging_threads = atoi(t); }
    { const char *t = getenv("CLAIR_AMD_ASYNC_STAGING"); if (t && t[0] == '0') e->staging_threads = 0; }
    { const char *t = getenv("CLAIR_AMD_COPY_STREAMS"); if (t) e->copy_mode = !strcmp(t, "slot") ? 0 : !strcmp(t, "two") ? 1 : !strcmp(t, "lane") ? 2 : 3; }
    { const char *t = getenv("CLAIR_AMD_D2H"); if (t) e->d2h_kernel = !strcmp(t, "kernel"); }
    { const char *t = getenv("CLAIR_AMD_CONVERT"); if (t) e->convert_on_lane = strcmp(t, "copy") != 0; }
    for (int i = 0; i < n_lanes; ++i) e->lanes.emplace_back(new Lane());
    e->slots.resize(n_slots);
    const size_t mp = e->max_pad;
    hipError_t r = hipSuccess;
    for (auto &lp : e->lanes) {
        Lane &l = *lp;
        if (r == hipSuccess) r = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
        if (r == hipSuccess) r = hipMalloc((void **)&l.zx, (size_t)T_POS * mp * 1024 * sizeof(float));
        if (r == hipSuccess) r = hipMalloc((void **)&l.a1, ((size_t)2 * T_POS * mp * 256 + 128 * 256) * sizeof(unsigned short));   // + slack rows read (never used) by gemm_split's ragged last tile
        if (r == hipSuccess) r = hipMalloc((void **)&l.a2, (size_t)T_POS * mp * 256 * sizeof(float));
        if (r == hipSuccess) r = hipMalloc((void **)&l.l4part, (size_t)L4_SPLITS * ((mp + L34_CAND - 1) / L34_CAND * L34_CAND) * L4_UNITS * sizeof(float));
        if (r == hipSuccess && fused_possible(e)) {
            const size_t words = fuse_words(e->max_pad) + 1 + fuse_claims(e);   // tickets | error word | claims
            r = hipMalloc((void **)&l.fuse_flags, words * sizeof(unsigned));
            if (r == hipSuccess) r = hipMemset(l.fuse_flags, 0, words * sizeof(unsigned));
        }
    }
    for (size_t i = 0; i < e->slots.size(); ++i) {
        Slot &s = e->slots[i];
        s.lane = (int)(i % e->lanes.size());
        if (e->copy_mode == 0) {            // a stream per slot for both directions
            hipStream_t st = nullptr;
            if (r == hipSuccess) r = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (r == hipSuccess) e->copy_streams.push_back(st);
            s.cin = s.cout = st;
        } else if (e->copy_mode == 1) {     // one stream for everything that comes in, one for everything that goes out
            while (r == hipSuccess && e->copy_streams.size() < 2) {
                hipStream_t st = nullptr;
                r = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                if (r == hipSuccess) e->copy_streams.push_back(st);
            }
            if (r == hipSuccess) { s.cin = e->copy_streams[0]; s.cout = e->copy_streams[1]; }
        } else if (e->copy_mode == 2) {     // on the lane's own stream, in line with its kernels
            s.cin = s.cout = e->lanes[s.lane]->stream;
        } else {                            // ONE stream for everything that comes in; the results leave on the lane's own stream
            if (r == hipSuccess && e->copy_streams.empty()) {
                hipStream_t st = nullptr;
                r = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                if (r == hipSuccess) e->copy_streams.push_back(st);
            }
            if (r == hipSuccess) { s.cin = e->copy_streams[0]; s.cout = e->lanes[s.lane]->stream; }
        }
        if (r == hipSuccess) r = hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming);
        if (r == hipSuccess) r = hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming);
        if (r == hipSuccess) r = hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming);
        if (r == hipSuccess) r = hipMalloc((void **)&s.d_x, mp * CLAIR_INPUT_FLOATS * sizeof(float));
        if (r == hipSuccess) r = hipMemset(s.d_x, 0, mp * CLAIR_INPUT_FLOATS * sizeof(float));
        if (r == hipSuccess) r = hipMalloc((void **)&s.d_out, mp * OUT_FLOATS * sizeof(float));
        if (r == hipSuccess) r = hipHostMalloc((void **)&s.h_out, h_out_floats(max_batch) * sizeof(float), hipHostMallocDefault);   // + the fused launch's error word
    }
    if (r != hipSuccess) {
        fail(nullptr, "allocating workspaces for max_batch=%d, %d slot(s) failed: %s", max_batch, n_slots, hipGetErrorString(r));
        clair_engine_destroy(e);
        return 1;
    }
    for (int i = 0; i < e->staging_threads; ++i) e->workers.emplace_back(staging_worker, e);
    *out = e;
    return 0;
}

void clair_engine_destroy(clair_engine_t *e) {
    if (!e) return;
    if (!e->workers.empty()) {
        { std::lock_guard<std::mutex> g(e->wmu); e->wstop = true; }
        e->wcv.notify_all();
        for (auto &t : e->workers) t.join();
    }
    (void)hipSetDevice(e->device);
    for (auto &lp : e->lanes) if (lp->stream) (void)hipStreamSynchronize(lp->stream);
    for (auto st : e->copy_streams) (void)hipStreamSynchronize(st);
    for (auto &s : e->slots) free_slot(s);
    for (auto st : e->copy_streams) (void)hipStreamDestroy(st);
    for (auto &lp : e->lanes) free_lane(*lp);
    for (auto &b : e->pinned) (void)hipHostFree(b.first);
    float *w[] = {e->bx1, e->bx2, e->b4, e->b5, e->bh};
    for (float *p : w) (void)hipFree(p);
    (void)hipFree(e->w5s); (void)hipFree(e->whs);
    (void)hipFree(e->wx2s); (void)hipFree(e->wh1s); (void)hipFree(e->wh2s); (void)hipFree(e->wx1s); (void)hipFree(e->w4s); (void)hipFree(e->w3s);
    delete e;
}

int clair_set_tensor(clair_engine_t *e, int id, const float *host, int64_t count) {
    if (!e) return fail(nullptr, "engine is NULL");
    if (id < 0 || id >= CLAIR_T_COUNT) return fail(e, "tensor id %d out of range", id);
    if (!host) return fail(e, "tensor %d: host pointer is NULL", id);
    if (count != TENSOR_COUNT[id]) return fail(e, "tensor %d: got %lld floats, expected %lld", id, (long long)count, (long long)TENSOR_COUNT[id]);
    e->host_tensors[id].assign(host, host + count);
    e->weights_ready = false;
    return 0;
}

int clair_finalize_weights(clair_engine_t *e) {
    if (!e) return fail(nullptr, "engine is NULL");
    for (int i = 0; i < CLAIR_T_COUNT; ++i)
        if ((int64_t)e->host_tensors[i].size() != TENSOR_COUNT[i]) return fail(e, "tensor %d has not been set", i);
    HIP_TRY(e, hipSetDevice(e->device));
    if (quiesce(e)) return 1;
    float **dev[] = {&e->bx1, &e->bx2, &e->b4, &e->b5, &e->bh};
    for (float **p : dev) { (void)hipFree(*p); *p = nullptr; }
    auto &T = e->host_tensors;
    if (upload(e, &e->bx1, pack_bias32(T[1], T[3])) || upload(e, &e->bx2, pack_bias32(T[5], T[7]))) return 1;
    {   // Wx2^T (gate-scaled, gate-row order) as A fragments of the weight-stationary projection GEMM (gemm_split.hip.h):
        // [gate tile][wm][mi][kk][plane][lane][8]
        std::vector<unsigned short> w2s((size_t)8 * 2 * 2 * 16 * 2 * 64 * 8);
        for (int gt = 0; gt < 8; ++gt)
            for (int wm = 0; wm < 2; ++wm)
                for (int mi = 0; mi < 2; ++mi)
                    for (int kk = 0; kk < 16; ++kk)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int R = gt * 128 + wm * 64 + mi * 32 + (lane & 31), k = 16 * kk + 8 * (lane >> 5) + j;
                                const int d = R >> 9, col = gate_col((R >> 7) & 3, (R >> 5) & 3, R & 31);
                                unsigned short hi, lo;
                                split2_host((d ? T[6] : T[4])[(size_t)k * 512 + col] * gate_scale(col), hi, lo);
                                const size_t base = ((((((size_t)gt * 2 + wm) * 2 + mi) * 16 + kk) * 2) * 64 + lane) * 8 + j;
                                w2s[base] = hi;
                                w2s[base + 64 * 8] = lo;
                            }
        if (upload16(e, &e->wx2s, w2s)) return 1;
    }
    if (upload16(e, &e->wh1s, pack_wt32(T[0], T[2], F_IN, 8)) || upload16(e, &e->wh2s, pack_wt32(T[4], T[6], 2 * HID, 8)) ||
        upload16(e, &e->wx1s, pack_wt32(T[0], T[2], 0, 2, (float)(1 << L32_X_SHIFT)))) return 1;
    // Power-of-two image shift of a tensor: puts its largest magnitude into [2^13, 2^14).  A freshly initialised W4 has sigma = 0.011 and a
    // trained one may be smaller still, i.e. residuals below the fp16 normal range -- the low plane would keep them to 3e-8 ABSOLUTE
    // only (common.hip.h); the kernel that consumes the product multiplies by 2^-shift (exact).
    auto image_shift = [](float vmax) {
        if (!(vmax > 0.0f) || !std::isfinite(vmax)) return 0;
        int ex = 0;
        (void)std::frexp(vmax, &ex);              // vmax = m * 2^ex, m in [0.5, 1)
        return std::max(-20, std::min(40, 14 - ex));
    };
    {   // L3 A fragments (dense.hip.h: l3l4_kernel): (W3[c]^T | b3[c]) * 2^w3_shift as fp16 split, [c][kk][plane][lane][8]: row u = lane%32,
        // k = 16kk + 8(lane/32) + j: t for k < 33, the bias at k = 33 (the activation operand carries 1.0 there), zero beyond and for u >= 30
        float vmax = 0.0f;
        for (float v : T[8]) vmax = std::max(vmax, std::fabs(v));
        for (float v : T[9]) vmax = std::max(vmax, std::fabs(v));
        e->w3_shift = image_shift(vmax);
        const float pow2 = std::ldexp(1.0f, e->w3_shift);
        std::vector<unsigned short> w3s((size_t)256 * 3 * 2 * 64 * 8, 0);
        for (int c = 0; c < 256; ++c)
            for (int kk = 0; kk < 3; ++kk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int u = lane & 31, k = 16 * kk + 8 * (lane >> 5) + j;
                        float v = 0.0f;
                        if (u < L3_UNITS && k < T_POS) v = T[8][((size_t)c * T_POS + k) * L3_UNITS + u];
                        else if (u < L3_UNITS && k == T_POS) v = T[9][(size_t)c * L3_UNITS + u];
                        unsigned short hi, lo;
                        split2_host(v * pow2, hi, lo);
                        const size_t base = ((((size_t)c * 3 + kk) * 2) * 64 + lane) * 8 + j;
                        w3s[base] = hi;
                        w3s[base + 64 * 8] = lo;
                    }
        if (upload16(e, &e->w3s, w3s)) return 1;
    }
    {   // W4 as fp16 split B fragments of the fused L3/L4 kernel (dense.hip.h): [cg][ks][nb][plane][lane][8]: row (2ks + lane/32)*256 + cg*8 + j
        // of W4 * 2^w4_shift, column nb*32 + lane%32; the kernel that reduces the split-K partials multiplies by 2^-w4_shift.
        float w4max = 0.0f;
        for (float v : T[10]) w4max = std::max(w4max, std::fabs(v));
        e->w4_shift = image_shift(w4max);
        const float w4_pow2 = std::ldexp(1.0f, e->w4_shift);
        std::vector<unsigned short> w4s((size_t)L34_GROUPS * L34_KS * 6 * 2 * 64 * 8);
        for (int cg = 0; cg < L34_GROUPS; ++cg)
            for (int ks = 0; ks < L34_KS; ++ks)
                for (int nb = 0; nb < 6; ++nb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int u = 2 * ks + (lane >> 5), col = nb * 32 + (lane & 31);
                            unsigned short hi, lo;
                            split2_host(T[10][((size_t)u * 256 + cg * L34_CH + j) * L4_UNITS + col] * w4_pow2, hi, lo);
                            const size_t base = (((((size_t)cg * L34_KS + ks) * 6 + nb) * 2) * 64 + lane) * 8 + j;
                            w4s[base] = hi;
                            w4s[base + 64 * 8] = lo;
                        }
        if (upload16(e, &e->w4s, w4s) || upload(e, &e->b4, T[11])) return 1;
    }
    {   // tail A fragments (dense.hip.h: tail_kernel): W5_k^T and Wh_k^T as fp16 split, each tensor shifted by its own power of two
        const int sizes[4] = {21, 3, 33, 33};
        std::vector<unsigned short> w5s((size_t)4 * 12 * 3 * 2 * 64 * 8), whs((size_t)4 * 6 * 2 * 2 * 64 * 8, 0);
        std::vector<float> bh(4 * 64, 0.0f);
        for (int k5 = 0; k5 < 4; ++k5) {
            const float *W5 = T[12].data() + (size_t)k5 * L4_UNITS * L5_UNITS;
            const std::vector<float> &Wh = T[14 + 2 * k5];
            float m5 = 0.0f, mh = 0.0f;
            for (int i = 0; i < L4_UNITS * L5_UNITS; ++i) m5 = std::max(m5, std::fabs(W5[i]));
            for (float v : Wh) mh = std::max(mh, std::fabs(v));
            e->w5_shift[k5] = image_shift(m5);
            e->wh_shift[k5] = image_shift(mh);
            const float p5 = std::ldexp(1.0f, e->w5_shift[k5]), ph = std::ldexp(1.0f, e->wh_shift[k5]);
            for (int ks = 0; ks < 12; ++ks)
                for (int nb = 0; nb < 3; ++nb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int k = 16 * ks + 8 * (lane >> 5) + j, n = nb * 32 + (lane & 31);
                            unsigned short hi, lo;
                            split2_host(W5[(size_t)k * L5_UNITS + n] * p5, hi, lo);
                            const size_t base = ((((((size_t)k5 * 12 + ks) * 3 + nb) * 2) * 64) + lane) * 8 + j;
                            w5s[base] = hi;
                            w5s[base + 64 * 8] = lo;
                        }
            for (int ks = 0; ks < 6; ++ks)
                for (int nb = 0; nb < 2; ++nb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int k = 16 * ks + 8 * (lane >> 5) + j, c = nb * 32 + (lane & 31);
                            unsigned short hi = 0, lo = 0;
                            if (c < sizes[k5]) split2_host(Wh[(size_t)k * sizes[k5] + c] * ph, hi, lo);
                            const size_t base = ((((((size_t)k5 * 6 + ks) * 2 + nb) * 2) * 64) + lane) * 8 + j;
                            whs[base] = hi;
                            whs[base + 64 * 8] = lo;
                        }
            for (int j = 0; j < sizes[k5]; ++j) bh[k5 * 64 + j] = T[15 + 2 * k5][j];
        }
        if (upload16(e, &e->w5s, w5s) || upload16(e, &e->whs, whs) || upload(e, &e->b5, T[13]) || upload(e, &e->bh, bh)) return 1;
    }
    e->weights_ready = true;
    return 0;
}

int clair_submit(clair_engine_t *e, int slot, const float *x, int n, float *gt21, float *genotype, float *l1, float *l2) {
    if (check_slot(e, slot)) return 1;
    if (n < 1 || n > e->max_batch) return fail(e, "n=%d out of range [1,%d]", n, e->max_batch);
    if (!x || !gt21 || !genotype || !l1 || !l2) return fail(e, "NULL input/output pointer");
    return submit_request(e, slot, clair_engine::Request{x, false, 0, n, nullptr, nullptr, gt21, genotype, l1, l2});
}

int clair_submit_counts(clair_engine_t *e, int slot, const int16_t *counts, int n, float *gt21, float *genotype, float *l1, float *l2) {
    if (check_slot(e, slot)) return 1;
    if (n < 1 || n > e->max_batch) return fail(e, "n=%d out of range [1,%d]", n, e->max_batch);
    if (!counts || !gt21 || !genotype || !l1 || !l2) return fail(e, "NULL input/output pointer");
    return submit_request(e, slot, clair_engine::Request{counts, true, 0, n, nullptr, nullptr, gt21, genotype, l1, l2});
}

int clair_slot_input(clair_engine_t *e, int slot, float **x_pinned) {
    if (check_slot(e, slot)) return 1;
    if (!x_pinned) return fail(e, "x_pinned is NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    Slot &s = e->slots[slot];
    if (ensure_slot_input(e, s)) return 1;
    *x_pinned = s.h_x;
    return 0;
}

int clair_wait(clair_engine_t *e, int slot) {
    if (check_slot(e, slot)) return 1;
    HIP_TRY(e, hipSetDevice(e->device));
    Slot &s = e->slots[slot];
    Lane &l = *e->lanes[s.lane];
    const int n = s.pending_n;
    if (s.staged) {     // handed to the staging worker: it has enqueued the batch by the time this returns
        std::unique_lock<std::mutex> g(e->wmu);
        e->wdone.wait(g, [&] { return s.staged != 1; });
        const bool failed = s.staged == 3;
        s.staged = 0;
        if (failed) {
            g.unlock();
            (void)hipStreamSynchronize(s.cout);
            s.pending_n = 0; s.o_calls = nullptr;
            e->error = s.worker_error;
            return 1;
        }
    }
    if (n) HIP_TRY(e, hipEventSynchronize(s.ev_out));
    bool again = s.refetch;
    s.refetch = false;
    if (l.fuse_flags) {
        std::lock_guard<std::mutex> g(l.order);
        unsigned bad;
        memcpy(&bad, s.h_out + h_word_offset(e->max_batch), sizeof bad);
        bool mine = false;
        for (const auto &r : l.fused_runs) mine = mine || r.slot == slot;
        if (n && bad && mine) {   // re-run on the two-launch path (d_x still holds the input), fetch the outputs again
            if (recover_fused(e, l, slot)) { s.pending_n = 0; return 1; }
            again = true;
        }
        l.fused_runs.erase(std::remove_if(l.fused_runs.begin(), l.fused_runs.end(), [slot](const Lane::FusedRun &r) { return r.slot == slot; }), l.fused_runs.end());
    }
    if (n && again) {
        std::lock_guard<std::mutex> g(l.order);
        if (s.o_calls) {
            if (enqueue_decode(e, l, s, n)) { s.pending_n = 0; return 1; }
            HIP_TRY(e, hipMemcpyAsync(s.h_calls, s.d_calls, (size_t)n * sizeof(clair_call_t), hipMemcpyDeviceToHost, l.stream));
        }
        if (s.o_gt21) HIP_TRY(e, hipMemcpyAsync(s.h_out, s.d_out, (size_t)n * OUT_FLOATS * sizeof(float), hipMemcpyDeviceToHost, l.stream));
        HIP_TRY(e, hipStreamSynchronize(l.stream));
    }
    if (s.o_gt21)
        for (int i = 0; i < n; ++i) {
            const float *row = s.h_out + (size_t)i * OUT_FLOATS;
            memcpy(s.o_gt21 + (size_t)i * 21, row, 21 * sizeof(float));
            memcpy(s.o_gt + (size_t)i * 3, row + 21, 3 * sizeof(float));
            memcpy(s.o_l1 + (size_t)i * 33, row + 24, 33 * sizeof(float));
            memcpy(s.o_l2 + (size_t)i * 33, row + 57, 33 * sizeof(float));
        }
    if (s.o_calls) memcpy(s.o_calls, s.h_calls, (size_t)n * sizeof(clair_call_t));
    s.o_calls = nullptr;
    s.pending_n = 0;
    return 0;
}

// The pipelined call with the decode on the device: input as float32 tensor (input_is_counts == 0) or raw int16 counts; calls != NULL
// asks for the call records (centre: [n][2] bytes, required then); the four probability arrays are optional then (all or none).
int clair_submit_ex(clair_engine_t *e, int slot, const void *input, int input_is_counts, int64_t input_stride_bytes, int n, const uint8_t *centre,
                    clair_call_t *calls, float *gt21, float *genotype, float *l1, float *l2) {
    if (check_slot(e, slot)) return 1;
    if (n < 1 || n > e->max_batch) return fail(e, "n=%d out of range [1,%d]", n, e->max_batch);
    const bool want_probs = gt21 || genotype || l1 || l2;
    if (!input) return fail(e, "NULL input pointer");
    if (want_probs && !(gt21 && genotype && l1 && l2)) return fail(e, "the four probability arrays come together or not at all");
    if (!want_probs && !calls) return fail(e, "nothing asked for: neither call records nor probabilities");
    if (calls && !centre) return fail(e, "call records need the candidates' centre bytes");
    if (input_stride_bytes != 0 && input_stride_bytes < (int64_t)(CLAIR_INPUT_FLOATS * (input_is_counts ? sizeof(short) : sizeof(float))))
        return fail(e, "input stride of %lld bytes is shorter than one candidate", (long long)input_stride_bytes);
    return submit_request(e, slot, clair_engine::Request{input, input_is_counts != 0, input_stride_bytes, n, centre, calls, gt21, genotype, l1, l2});
}

// The decode alone, on probabilities the caller already has (call_var's --input_probabilities path, clair/call_var.py:1276-1309): synchronous.
int clair_decode(clair_engine_t *e, int slot, const float *x, const float *gt21, const float *genotype, const float *l1, const float *l2, int n,
                 const uint8_t *centre, clair_call_t *calls) {
    if (!e) return fail(nullptr, "engine is NULL");
    if (slot < 0 || slot >= (int)e->slots.size()) return fail(e, "slot %d out of range [0,%d)", slot, (int)e->slots.size());
    if (n < 1 || n > e->max_batch) return fail(e, "n=%d out of range [1,%d]", n, e->max_batch);
    if (!x || !gt21 || !genotype || !l1 || !l2 || !centre || !calls) return fail(e, "NULL input/output pointer");
    HIP_TRY(e, hipSetDevice(e->device));
    Slot &s = e->slots[slot];
    if (s.pending_n) return fail(e, "slot %d still has a pending submit; call clair_wait first", slot);
    if (!s.d_centre) {
        HIP_TRY(e, hipMalloc((void **)&s.d_centre, (size_t)e->max_pad * 2));
        HIP_TRY(e, hipHostMalloc((void **)&s.h_centre, (size_t)e->max_batch * 2, hipHostMallocDefault));
        HIP_TRY(e, hipMalloc((void **)&s.d_calls, (size_t)e->max_pad * sizeof(clair_call_t)));
        HIP_TRY(e, hipHostMalloc((void **)&s.h_calls, (size_t)e->max_batch * sizeof(clair_call_t), hipHostMallocDefault));
    }
    for (int i = 0; i < n; ++i) {   // the packed [n][90] rows the kernels exchange
        float *row = s.h_out + (size_t)i * OUT_FLOATS;
        memcpy(row, gt21 + (size_t)i * 21, 21 * sizeof(float));
        memcpy(row + 21, genotype + (size_t)i * 3, 3 * sizeof(float));
        memcpy(row + 24, l1 + (size_t)i * 33, 33 * sizeof(float));
        memcpy(row + 57, l2 + (size_t)i * 33, 33 * sizeof(float));
    }
    memcpy(s.h_centre, centre, (size_t)n * 2);
    Lane &l = *e->lanes[s.lane];
    std::lock_guard<std::mutex> g(l.order);
    HIP_TRY(e, hipMemcpyAsync(s.d_x, x, (size_t)n * CLAIR_INPUT_FLOATS * sizeof(float), hipMemcpyHostToDevice, l.stream));
    HIP_TRY(e, hipMemcpyAsync(s.d_out, s.h_out, (size_t)n * OUT_FLOATS * sizeof(float), hipMemcpyHostToDevice, l.stream));
    HIP_TRY(e, hipMemcpyAsync(s.d_centre, s.h_centre, (size_t)n * 2, hipMemcpyHostToDevice, l.stream));
    if (enqueue_decode(e, l, s, n)) return 1;
    HIP_TRY(e, hipMemcpyAsync(s.h_calls, s.d_calls, (size_t)n * sizeof(clair_call_t), hipMemcpyDeviceToHost, l.stream));
    HIP_TRY(e, hipStreamSynchronize(l.stream));
    memcpy(calls, s.h_calls, (size_t)n * sizeof(clair_call_t));
    return 0;
}

int clair_predict(clair_engine_t *e, const float *x, int n, float *gt21, float *genotype, float *l1, float *l2) {
    if (clair_submit(e, 0, x, n, gt21, genotype, l1, l2)) return 1;
    return clair_wait(e, 0);
}

int clair_dataset_alloc(clair_engine_t *e, int64_t N, void **x_dev, void **out_dev) {
    if (!e) return fail(nullptr, "engine is NULL");
    if (N < 1 || !x_dev || !out_dev) return fail(e, "bad arguments to clair_dataset_alloc");
    HIP_TRY(e, hipSetDevice(e->device));
    // +max_pad rows of zeroed slack so the last (padded) tile of a run never reads past the end
    const size_t rows = (size_t)N + e->max_pad;
    HIP_TRY(e, hipMalloc(x_dev, rows * CLAIR_INPUT_FLOATS * sizeof(float)));
    HIP_TRY(e, hipMemset(*x_dev, 0, rows * CLAIR_INPUT_FLOATS * sizeof(float)));
    HIP_TRY(e, hipMalloc(out_dev, rows * OUT_FLOATS * sizeof(float)));
    HIP_TRY(e, hipMemset(*out_dev, 0, rows * OUT_FLOATS * sizeof(float)));
    return 0;
}

int clair_dataset_free(clair_engine_t *e, void *x_dev, void *out_dev) {
    if (!e) return fail(nullptr, "engine is NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipFree(x_dev));
    HIP_TRY(e, hipFree(out_dev));
    return 0;
}

int clair_dataset_upload(clair_engine_t *e, void *x_dev, int64_t first, const float *x_host, int64_t n) {
    if (!e) return fail(nullptr, "engine is NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipMemcpy((float *)x_dev + (size_t)first * CLAIR_INPUT_FLOATS, x_host, (size_t)n * CLAIR_INPUT_FLOATS * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int clair_dataset_download(clair_engine_t *e, const void *out_dev, int64_t first, float *out_host, int64_t n) {
    if (!e) return fail(nullptr, "engine is NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipMemcpy(out_host, (const float *)out_dev + (size_t)first * OUT_FLOATS, (size_t)n * OUT_FLOATS * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int clair_run_resident(clair_engine_t *e, int slot, const void *x_dev, void *out_dev, int64_t first, int n) {
    if (check_slot(e, slot)) return 1;
    if (n < 1 || n > e->max_batch) return fail(e, "n=%d out of range [1,%d]", n, e->max_batch);
    HIP_TRY(e, hipSetDevice(e->device));
    Lane &l = *e->lanes[e->slots[slot].lane];
    std::lock_guard<std::mutex> g(l.order);
    return enqueue_forward(e, l, (const float *)x_dev + (size_t)first * CLAIR_INPUT_FLOATS, (float *)out_dev + (size_t)first * OUT_FLOATS, n, -1);
}

int clair_sync(clair_engine_t *e) {
    if (!e) return fail(nullptr, "engine is NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    if (quiesce(e)) return 1;
    return check_fused_placement(e);
}

int clair_timing_enable(clair_engine_t *e, int on) {
    if (!e) return fail(nullptr, "engine is NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    if (drain_timers(e)) return 1;
    e->timing_mask = on == 0 ? 0u : (on == 1 ? ~0u : (unsigned)on);
    return 0;
}

int clair_kernel_times(clair_engine_t *e, double *ms_sum, int64_t *launches) {
    if (!e) return fail(nullptr, "engine is NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    if (drain_timers(e)) return 1;
    for (int k = 0; k < CLAIR_K_COUNT; ++k) {
        if (ms_sum) ms_sum[k] = e->ms_sum[k];
        if (launches) launches[k] = e->launches[k];
    }
    return 0;
}

int clair_timing_reset(clair_engine_t *e) {
    if (!e) return fail(nullptr, "engine is NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    if (drain_timers(e)) return 1;
    for (int k = 0; k < CLAIR_K_COUNT; ++k) { e->ms_sum[k] = 0; e->launches[k] = 0; }
    return 0;
}

int clair_kernel_workgroups(clair_engine_t *e, int n, int *workgroups) {
    if (!e) return fail(nullptr, "engine is NULL");
    if (!workgroups) return fail(e, "workgroups is NULL");
    if (n < 1 || n > e->max_batch) return fail(e, "n %d out of range [1,%d]", n, e->max_batch);
    const int n_pad = (n + 31) & ~31, ntiles = n_pad / 32;
    const int x_tiles = (T_POS * n_pad + GS_ROWS - 1) / GS_ROWS;
    for (int k = 0; k < CLAIR_K_COUNT; ++k) workgroups[k] = 0;
    workgroups[CLAIR_K_LSTM1] = ntiles * 2;
    workgroups[CLAIR_K_LSTM2] = use_lstm2_pair(e, ntiles) ? ((ntiles + 1) / 2) * 2 : ntiles * 2;
    workgroups[CLAIR_K_PROJ2] = 32 * std::min(e->proj2_groups, (x_tiles + 7) / 8);
    if (use_lstm2_fused(e, ntiles)) {   // one launch: the projection workgroups first, then the recurrent ones
        workgroups[CLAIR_K_LSTM2] = 32 * e->fused_groups + 32 * ((ntiles / 2 + 7) / 8);
        workgroups[CLAIR_K_PROJ2] = 0;
    }
    workgroups[CLAIR_K_L4] = l34_grid((n_pad + L34_CAND - 1) / L34_CAND);
    workgroups[CLAIR_K_TAIL] = n_pad / TAIL_TILE;
    workgroups[CLAIR_K_DECODE] = (n + 3) / 4;       // launched only by clair_submit_ex with call records asked for
    return 0;
}

int clair_pinned_alloc(clair_engine_t *e, int64_t bytes, void **ptr) {
    if (!e) return fail(nullptr, "engine is NULL");
    if (!ptr || bytes < 1) return fail(e, "bad arguments to clair_pinned_alloc");
    HIP_TRY(e, hipSetDevice(e->device));
    void *p = nullptr;
    HIP_TRY(e, hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault));
    {
        std::lock_guard<std::mutex> g(e->pinned_mu);
        e->pinned.emplace_back((char *)p, (size_t)bytes);
    }
    *ptr = p;
    return 0;
}

int clair_pinned_free(clair_engine_t *e, void *ptr) {
    if (!e) return fail(nullptr, "engine is NULL");
    bool known = false;
    {
        std::lock_guard<std::mutex> g(e->pinned_mu);
        for (const auto &b : e->pinned) known |= b.first == (char *)ptr;
    }
    if (!known) return fail(e, "clair_pinned_free: not a buffer of clair_pinned_alloc");
    HIP_TRY(e, hipSetDevice(e->device));
    if (quiesce(e)) return 1;   // no copy may still be reading it (and no staging worker is looking it up)
    HIP_TRY(e, hipHostFree(ptr));
    std::lock_guard<std::mutex> g(e->pinned_mu);
    for (size_t i = 0; i < e->pinned.size(); ++i)
        if (e->pinned[i].first == (char *)ptr) { e->pinned.erase(e->pinned.begin() + (long)i); break; }
    return 0;
}

int clair_engine_counter(clair_engine_t *e, int which, int64_t *value) {
    if (!e) return fail(nullptr, "engine is NULL");
    if (!value) return fail(e, "value is NULL");
    switch (which) {
        case 0: *value = e->fused_launches; return 0;
        case 1: *value = e->fused_recoveries; return 0;
        default: return fail(e, "clair_engine_counter: unknown counter %d", which);
    }
}

int clair_debug_read(clair_engine_t *e, int slot, int which, float *host, int64_t count) {
    if (check_slot(e, slot)) return 1;
    HIP_TRY(e, hipSetDevice(e->device));
    if (quiesce(e)) return 1;
    Lane &s = *e->lanes[e->slots[slot].lane];
    const float *src = nullptr;
    int64_t avail = 0;
    const int64_t np = s.last_n_pad;
    if (which == 1) {   // LSTM1 output lives as two fp16 planes: hand back their fp32 sum
        avail = (int64_t)T_POS * np * 256;
        if (count > avail) return fail(e, "clair_debug_read: asked %lld floats, tap 1 holds %lld", (long long)count, (long long)avail);
        std::vector<unsigned short> planes((size_t)2 * avail);
        HIP_TRY(e, hipMemcpy(planes.data(), s.a1, planes.size() * sizeof(unsigned short), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < count; ++i) host[i] = f16_value(planes[i]) + f16_value(planes[avail + i]);
        return 0;
    }
    if (which == 2) {   // LSTM2 output lives channel-group-major [32][33][n_pad][8] (lstm32.hip.h): hand it back as [33][n_pad][256]
        avail = (int64_t)T_POS * np * 256;
        if (count > avail) return fail(e, "clair_debug_read: asked %lld floats, tap 2 holds %lld", (long long)count, (long long)avail);
        std::vector<float> raw((size_t)avail);
        HIP_TRY(e, hipMemcpy(raw.data(), s.a2, raw.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < count; ++i) {
            const int64_t t = i / (np * 256), n = (i / 256) % np, c = i % 256;
            host[i] = raw[(size_t)((((c / 8) * T_POS + t) * np + n) * 8 + c % 8)];
        }
        return 0;
    }
    switch (which) {
        case 4: if (!e->tap_l3) return fail(e, "clair_debug_read: tap 4 needs CLAIR_AMD_TAP_L3=1 at engine creation");
                src = s.zx; avail = np * L3_OUT; break;
        case 5: if (!e->l34_stamps) return fail(e, "clair_debug_read: tap 5 needs CLAIR_AMD_L34_STAMPS=1 at engine creation");
                src = s.zx; avail = ((np + L34_CAND - 1) / L34_CAND) * L4_SPLITS * 8 * 16 * 2; break;   // uint64 pairs of floats
        case 3: {   // split-K partials live in the accumulator layout (dense.hip.h): hand them back as [8 splits][n_pad][192] in W4's own units
            avail = (int64_t)L4_SPLITS * np * L4_UNITS;
            if (count > avail) return fail(e, "clair_debug_read: asked %lld floats, tap 3 holds %lld", (long long)count, (long long)avail);
            const int64_t nblk = (np + L34_CAND - 1) / L34_CAND;
            std::vector<float> raw((size_t)L4_SPLITS * nblk * L34_CAND * L4_UNITS);
            HIP_TRY(e, hipMemcpy(raw.data(), s.l4part, raw.size() * sizeof(float), hipMemcpyDeviceToHost));
            const float inv = std::ldexp(1.0f, -e->w4_shift) / L34_ACT_SCALE;
            for (int64_t i = 0; i < count; ++i) {
                const int64_t cg = i / (np * L4_UNITS), n = (i / L4_UNITS) % np, col = i % L4_UNITS;
                const int64_t blk = n / 64, row = n % 64, mb = row / 32, a = (row % 32) / 8, hq = (row % 8) / 4, r = row % 4;
                const int64_t nh = col / 96, nb = (col % 96) / 32, l32 = col % 32;
                host[i] = raw[(size_t)((((((cg * nblk + blk) * 2 + nh) * 6 + mb * 3 + nb) * 4 + a) * 64 + hq * 32 + l32) * 4 + r)] * inv;
            }
            return 0;
        }
        default: return fail(e, "clair_debug_read: unknown tap %d", which);
    }
    if (count > avail) return fail(e, "clair_debug_read: asked %lld floats, tap %d holds %lld", (long long)count, which, (long long)avail);
    HIP_TRY(e, hipMemcpy(host, src, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
