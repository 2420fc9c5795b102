// Host-side helpers of the Clair hot path (include/clair_host.h): native ingest.  Plain C++17, no HIP.
#include "../../include/clair_host.h"

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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

int clair_host_parse_tensors(const char *buf, int64_t len, int final, int max_rows,
                             float *x, int32_t *tok, int *rows_taken, int *rows_kept, int64_t *bytes_consumed) {
    if (!buf || !x || !tok || !rows_taken || !rows_kept || !bytes_consumed || len < 0 || max_rows < 0)
        return fail("clair_host_parse_tensors: bad arguments");
    constexpr int NV = CLAIR_HOST_VALUES;
    int taken = 0, kept = 0;
    int64_t pos = 0;
    // token boundaries of the current line: only the last NV + 3 matter, but a line has to have exactly NV + 3
    static thread_local int32_t starts[NV + 8], ends[NV + 8];
    while (taken < max_rows && pos < len) {
        const char *nl = (const char *)memchr(buf + pos, '\n', (size_t)(len - pos));
        int64_t line_end;
        if (nl) line_end = nl - buf;
        else if (final) line_end = len;
        else break;                                   // incomplete last line: the caller supplies more text
        float *row = x + (size_t)kept * NV;
        // fast path: what CreateTensor writes -- three head tokens, then exactly 1056 plain decimal integers, single separators
        bool fast_ok = false;
        {
            const char *p = buf + pos, *e = buf + line_end;
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
        if (!fast_ok) {
        int ntok = 0;
        bool too_many = false;
        int64_t i = pos;
        while (i < line_end) {
            while (i < line_end && is_space((unsigned char)buf[i])) ++i;
            if (i >= line_end) break;
            const int64_t s = i;
            while (i < line_end && !is_space((unsigned char)buf[i])) ++i;
            if (ntok < NV + 8) { starts[ntok] = (int32_t)s; ends[ntok] = (int32_t)i; ++ntok; }
            else too_many = true;
        }
        if (too_many || ntok > NV + 3)
            return fail("line %d: %s columns before the %d tensor values, expected exactly 3 (ctg pos refseq)", taken,
                        "more than 3", NV);
        if (ntok < NV)
            return fail("line %d: %d columns, fewer than the %d tensor values", taken, ntok, NV);
        if (ntok != NV + 3)
            return fail("line %d: %d columns before the %d tensor values, expected exactly 3 (ctg pos refseq)", taken, ntok - NV, NV);
        for (int v = 0; v < NV; ++v)
            if (!parse_value(buf + starts[3 + v], buf + ends[3 + v], &row[v]))
                return fail("line %d: value %d (\"%.*s\") is not a number", taken, v, (int)(ends[3 + v] - starts[3 + v]), buf + starts[3 + v]);
        }
        if (ends[2] - starts[2] <= 16)
            return fail("line %d: reference sequence has %d characters, the centre base is index 16", taken, ends[2] - starts[2]);
        ++taken;
        pos = nl ? line_end + 1 : line_end;
        if (!is_iupac((unsigned char)buf[starts[2] + 16])) continue;   // dropped: the row slot is reused
        for (int g = 0; g < NV; g += 4) {                               // channels 1..3 -= channel 0
            const float c0 = row[g];
            row[g + 1] -= c0; row[g + 2] -= c0; row[g + 3] -= c0;
        }
        for (int t = 0; t < 3; ++t) { tok[kept * 6 + 2 * t] = starts[t]; tok[kept * 6 + 2 * t + 1] = ends[t] - starts[t]; }
        ++kept;
    }
    *rows_taken = taken;
    *rows_kept = kept;
    *bytes_consumed = pos;
    return 0;
}

}  // extern "C"
