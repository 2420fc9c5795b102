// Host-side helpers of the Clair hot path (include/clair_host.h): native ingest.  Plain C++17, no HIP.
#include "../../include/clair_host.h"

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

namespace {

thread_local std::string g_error;

int fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return 1;
}

// str.split() whitespace: space, \t, \n, \r, \v, \f (and the ASCII separators 0x1c-0x1f, which text records never hold)
inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

// centre-base filter: keys of IUPAC_base_to_num_dict (shared/utils.py:24-27)
inline bool is_iupac(unsigned char c) {
    switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'U': case 'R': case 'Y': case 'S':
        case 'W': case 'K': case 'M': case 'B': case 'D': case 'H': case 'V': case 'N': return true;
        default: return false;
    }
}

// One value token -> float32, as np.array([...], dtype=np.float32) parses strings: decimal integers (what CreateTensor
// writes, %d) on a fast path, anything else through strtof.
bool parse_value(const char *p, const char *end, float *out) {
    const char *q = p;
    bool neg = false;
    if (q < end && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
    if (q < end && end - q <= 18) {
        long long v = 0;
        const char *r = q;
        while (r < end && *r >= '0' && *r <= '9') v = v * 10 + (*r++ - '0');
        if (r == end && r > q) { *out = (float)(neg ? -v : v); return true; }
    }
    char tmp[64];
    const size_t n = (size_t)(end - p);
    if (n == 0 || n >= sizeof tmp) return false;
    memcpy(tmp, p, n);
    tmp[n] = 0;
    char *stop = nullptr;
    errno = 0;
    const float f = strtof(tmp, &stop);
    if (stop != tmp + n) return false;
    *out = f;
    return true;
}

}  // namespace

// shared with host_decode.cpp
int clair_host_fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return 1;
}

extern "C" {

int clair_host_abi_version(void) { return CLAIR_HOST_ABI_VERSION; }
const char *clair_host_last_error(void) { return g_error.c_str(); }

int clair_host_threads(int work_items) {
    // worker threads for one call: CLAIR_HOST_THREADS, else up to 16 hardware threads; small calls stay single-threaded
    static const int configured = [] {
        const char *e = getenv("CLAIR_HOST_THREADS");
        int n = e ? atoi(e) : 0;
        if (n <= 0) { n = (int)std::thread::hardware_concurrency(); if (n > 16) n = 16; }
        return n < 1 ? 1 : n;
    }();
    const int by_work = work_items / 128;
    return by_work < 1 ? 1 : (by_work < configured ? by_work : configured);
}

}  // extern "C" (closed for the helper below; re-opened after it)

// work(t) for t in [0, nthreads) on a persistent pool (the calling thread takes part).  A call per 1024-candidate batch used to create
// and join up to 16 threads -- 0.2-0.3 ms, as long as the work itself once the decode's arithmetic moved to the GPU.  One job at a
// time: a second caller that finds the pool busy (or a child process that inherited a pool without threads) runs its own threads.
void clair_host_parallel(int nthreads, const std::function<void(int)> &work) {
    if (nthreads <= 1) { work(0); return; }
    struct Pool {
        std::mutex m, run;
        std::condition_variable wake, done;
        std::vector<std::thread> workers;
        const std::function<void(int)> *job = nullptr;
        int n_jobs = 0, next = 0, pending = 0;
        uint64_t epoch = 0;
        pid_t owner = 0;
        void loop() {
            uint64_t seen = 0;
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                wake.wait(lk, [&] { return epoch != seen; });
                seen = epoch;
                while (next < n_jobs) {
                    const int t = next++;
                    lk.unlock();
                    (*job)(t);
                    lk.lock();
                    if (--pending == 0) done.notify_all();
                }
            }
        }
    };
    static Pool *pool = new Pool;      // never destroyed: its threads are detached and live as long as the process
    std::unique_lock<std::mutex> one(pool->run, std::try_to_lock);
    if (!one.owns_lock() || (pool->owner != 0 && pool->owner != getpid())) {
        std::vector<std::thread> own;
        for (int t = 1; t < nthreads; ++t) own.emplace_back([&, t] { work(t); });
        work(0);
        for (auto &th : own) th.join();
        return;
    }
    std::unique_lock<std::mutex> lk(pool->m);
    if (pool->owner == 0) pool->owner = getpid();
    while ((int)pool->workers.size() < nthreads - 1) {
        pool->workers.emplace_back([] { pool->loop(); });
        pool->workers.back().detach();
    }
    pool->job = &work;
    pool->n_jobs = nthreads;
    pool->next = 0;
    pool->pending = nthreads;
    ++pool->epoch;
    pool->wake.notify_all();
    while (pool->next < pool->n_jobs) {   // the caller works too
        const int t = pool->next++;
        lk.unlock();
        work(t);
        lk.lock();
        --pool->pending;
    }
    pool->done.wait(lk, [&] { return pool->pending == 0; });
    pool->job = nullptr;
}

extern "C" {

int clair_host_parse_tensors(const char *buf, int64_t len, int final, int max_rows,
                             float *x, int32_t *tok, int *rows_taken, int *rows_kept, int64_t *bytes_consumed) {
    if (!buf || !x || !tok || !rows_taken || !rows_kept || !bytes_consumed || len < 0 || max_rows < 0)
        return fail("clair_host_parse_tensors: bad arguments");
    constexpr int NV = CLAIR_HOST_VALUES;
    // pass 1: line boundaries of the lines to take
    std::vector<int64_t> lstart, lend;
    lstart.reserve((size_t)max_rows);
    lend.reserve((size_t)max_rows);
    int64_t pos = 0;
    while ((int)lstart.size() < max_rows && pos < len) {
        const char *nl = (const char *)memchr(buf + pos, '\n', (size_t)(len - pos));
        if (!nl && !final) break;                     // incomplete last line: the caller supplies more text
        const int64_t e = nl ? nl - buf : len;
        lstart.push_back(pos);
        lend.push_back(e);
        pos = nl ? e + 1 : e;
    }
    const int taken = (int)lstart.size();
    // pass 2: every line into the row slot of its line index (independent: threads), pass 3: drop the filtered rows
    std::vector<signed char> state((size_t)taken, 0);   // 1 kept, 0 dropped, -1 malformed
    std::vector<std::string> errors((size_t)taken);
    std::vector<int32_t> ltok((size_t)taken * 6);
    auto parse_line = [&](int li) {
        int32_t starts[3], ends[3];
        float *row = x + (size_t)li * NV;
        const char *p = buf + lstart[li], *e = buf + lend[li];
        auto bad = [&](const char *fmt, int a, int b) {
            char msg[256];
            snprintf(msg, sizeof msg, fmt, li, a, b);
            errors[li] = msg;
            state[li] = -1;
        };
        // fast path: what CreateTensor writes -- three head tokens, then exactly 1056 plain decimal integers
        bool fast_ok = false;
        {
            int t = 0;
            for (; t < 3; ++t) {
                while (p < e && is_space((unsigned char)*p)) ++p;
                if (p >= e) break;
                starts[t] = (int32_t)(p - buf);
                while (p < e && !is_space((unsigned char)*p)) ++p;
                ends[t] = (int32_t)(p - buf);
            }
            if (t == 3) {
                int v = 0;
                for (; v < NV; ++v) {
                    while (p < e && (*p == ' ' || *p == '\t')) ++p;
                    if (p >= e) break;
                    bool neg = false;
                    if (*p == '-') { neg = true; ++p; }
                    const char *d0 = p;
                    int val = 0;
                    while (p < e && (unsigned)(*p - '0') <= 9u) val = val * 10 + (*p++ - '0');
                    if (p == d0 || p - d0 > 9 || (p < e && !is_space((unsigned char)*p))) break;   // not a short plain integer
                    row[v] = (float)(neg ? -val : val);
                }
                if (v == NV) {
                    while (p < e && is_space((unsigned char)*p)) ++p;
                    fast_ok = p == e;            // nothing may follow the 1056th value
                }
            }
        }
        if (!fast_ok) {   // general path: tokenise the whole line, as str.split() does
            std::vector<int32_t> ts, te;
            int64_t i = lstart[li];
            const int64_t line_end = lend[li];
            while (i < line_end) {
                while (i < line_end && is_space((unsigned char)buf[i])) ++i;
                if (i >= line_end) break;
                const int64_t s0 = i;
                while (i < line_end && !is_space((unsigned char)buf[i])) ++i;
                ts.push_back((int32_t)s0);
                te.push_back((int32_t)i);
                if ((int)ts.size() > NV + 8) break;
            }
            const int ntok = (int)ts.size();
            if (ntok < NV) return bad("line %d: %d columns, fewer than the %d tensor values", ntok, NV);
            if (ntok != NV + 3) return bad("line %d: %d columns before the %d tensor values, expected exactly 3 (ctg pos refseq)", ntok - NV, NV);
            for (int t = 0; t < 3; ++t) { starts[t] = ts[t]; ends[t] = te[t]; }
            for (int v = 0; v < NV; ++v)
                if (!parse_value(buf + ts[3 + v], buf + te[3 + v], &row[v])) return bad("line %d: value %d is not a number%.0d", v, 0);
        }
        if (ends[2] - starts[2] <= 16) return bad("line %d: reference sequence has %d characters, the centre base is index 16%.0d", ends[2] - starts[2], 0);
        for (int t = 0; t < 3; ++t) { ltok[(size_t)li * 6 + 2 * t] = starts[t]; ltok[(size_t)li * 6 + 2 * t + 1] = ends[t] - starts[t]; }
        if (!is_iupac((unsigned char)buf[starts[2] + 16])) { state[li] = 0; return; }   // dropped (utils.py:90-91)
        for (int g = 0; g < NV; g += 4) {                                               // channels 1..3 -= channel 0
            const float c0 = row[g];
            row[g + 1] -= c0; row[g + 2] -= c0; row[g + 3] -= c0;
        }
        state[li] = 1;
    };
    const int nthreads = clair_host_threads(taken);
    if (nthreads <= 1) {
        for (int li = 0; li < taken; ++li) parse_line(li);
    } else {
        clair_host_parallel(nthreads, [&](int t) { for (int li = t; li < taken; li += nthreads) parse_line(li); });      // the persistent pool: no thread is created per batch
    }
    int kept = 0;
    for (int li = 0; li < taken; ++li) {
        if (state[li] < 0) return fail("%s", errors[li].c_str());       // the first malformed line, as a sequential parse reports it
        if (state[li] == 0) continue;
        if (kept != li) memmove(x + (size_t)kept * NV, x + (size_t)li * NV, NV * sizeof(float));
        memcpy(tok + (size_t)kept * 6, &ltok[(size_t)li * 6], 6 * sizeof(int32_t));
        ++kept;
    }
    *rows_taken = taken;
    *rows_kept = kept;
    *bytes_consumed = pos;
    return 0;
}

}  // extern "C"

namespace {
template <typename T>
int counts_to_input(const T *counts, int64_t n_quads, float *x) {
    if (!counts || !x || n_quads < 0) return fail("clair_host_counts_to_input: bad arguments");
    const int nthreads = clair_host_threads((int)std::min<int64_t>(n_quads / (CLAIR_HOST_VALUES / 4), 1 << 30));
    auto work = [&](int t) {
        const int64_t lo = n_quads * t / nthreads, hi = n_quads * (t + 1) / nthreads;
        for (int64_t q = lo; q < hi; ++q) {   // float32 first, then the subtraction, as the reference does (exact for counts below 2^24 either way)
            const float c0 = (float)counts[4 * q];
            x[4 * q] = c0;
            x[4 * q + 1] = (float)counts[4 * q + 1] - c0;
            x[4 * q + 2] = (float)counts[4 * q + 2] - c0;
            x[4 * q + 3] = (float)counts[4 * q + 3] - c0;
        }
    };
    if (nthreads <= 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    return 0;
}
}  // namespace

extern "C" int clair_host_counts_to_input_i16(const int16_t *counts, int64_t n_quads, float *x) { return counts_to_input(counts, n_quads, x); }
extern "C" int clair_host_counts_to_input_i32(const int32_t *counts, int64_t n_quads, float *x) { return counts_to_input(counts, n_quads, x); }

// CRC32C (Castagnoli, reflected polynomial 0x82F63B78) of a byte range, slicing-by-8: what the TensorFlow bundle format
// protects its index blocks and tensor data with (clair_amd/tf_bundle.py masks it the LevelDB way).  Known answer:
// "123456789" -> 0xE3069283.
extern "C" uint32_t clair_host_crc32c(const uint8_t *data, int64_t n) {
    static uint32_t table[8][256];
    static bool ready = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
        return true;
    }();
    (void)ready;
    uint32_t crc = 0xFFFFFFFFu;
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint32_t lo, hi;
        memcpy(&lo, data + i, 4);
        memcpy(&hi, data + i + 4, 4);
        lo ^= crc;
        crc = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
              table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    }
    for (; i < n; ++i) crc = table[0][(crc ^ data[i]) & 0xFF] ^ (crc >> 8);
    return crc ^ 0xFFFFFFFFu;
}
