// lr_format.hip - host-only: the text form of the logs.
//
// The reference writes its logs through Python's csv module (LRF:334-359, DD:236-238): every number is str(float) - the
// SHORTEST decimal string that reads back to the same double, fixed notation with at least one fractional digit
// ("24.0") for 1e-4 <= |x| < 1e16 and exponent notation with a sign and at least two exponent digits otherwise
// ("1e-05", "1.5e+16").  A many-chain run samples millions of numbers per window; CPython needs ~0.3 us for each
// (12 us per logged row, more than the device needs for the 1000 iterations between two samples of 1024 chains), so
// the window's rows are formatted here: std::to_chars gives the shortest digits, the layout rules above are Python's
// (Objects/floatobject.c float_repr -> PyOS_double_to_string(x, 'r', 0, Py_DTSF_ADD_DOT_0): exponent form when the
// decimal point position decpt <= -4 or decpt > 16).  No GPU involved.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/literate_hip.h"

namespace {

// str(float(v)) into p (at least 32 bytes free); returns the end
inline char* lr_py_float(char* p, double v) {
    if (std::isnan(v)) {
        std::memcpy(p, "nan", 3);
        return p + 3;
    }
    if (std::isinf(v)) {
        if (v < 0) *p++ = '-';
        std::memcpy(p, "inf", 3);
        return p + 3;
    }
    char sci[40];
    // shortest round-trip digits: [-]d[.ddd]e[+-]XX
    const std::to_chars_result r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);
    const char* s = sci;
    const char* end = r.ptr;
    if (*s == '-') *p++ = *s++;
    char digits[24];
    int nd = 0;
    digits[nd++] = *s++;
    if (*s == '.') {
        ++s;
        while (*s != 'e') digits[nd++] = *s++;
    }
    ++s;  // 'e'
    const bool eneg = *s == '-';
    ++s;
    int ex = 0;
    while (s < end) ex = ex * 10 + (*s++ - '0');
    if (eneg) ex = -ex;
    const int decpt = ex + 1;  // value = 0.d1d2... x 10^decpt
    if (decpt <= -4 || decpt > 16) {
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            std::memcpy(p, digits + 1, (size_t)(nd - 1));
            p += nd - 1;
        }
        *p++ = 'e';
        int e = decpt - 1;
        *p++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        if (e >= 100) *p++ = (char)('0' + e / 100);
        *p++ = (char)('0' + (e / 10) % 10);
        *p++ = (char)('0' + e % 10);
        return p;
    }
    if (decpt <= 0) {  // 0.000ddd
        *p++ = '0';
        *p++ = '.';
        for (int k = 0; k < -decpt; ++k) *p++ = '0';
        std::memcpy(p, digits, (size_t)nd);
        return p + nd;
    }
    if (decpt >= nd) {  // ddd000.0
        std::memcpy(p, digits, (size_t)nd);
        p += nd;
        for (int k = nd; k < decpt; ++k) *p++ = '0';
        *p++ = '.';
        *p++ = '0';
        return p;
    }
    std::memcpy(p, digits, (size_t)decpt);  // dd.ddd
    p += decpt;
    *p++ = '.';
    std::memcpy(p, digits + decpt, (size_t)(nd - decpt));
    return p + (nd - decpt);
}

inline char* lr_py_int(char* p, double v) {
    if (!(std::fabs(v) < 9.2e18)) return lr_py_float(p, v);     // (nan, inf, beyond int64: no integer to write)
    const std::to_chars_result r = std::to_chars(p, p + 24, (long long)v);
    return r.ptr;
}

}  // namespace

extern "C" int64_t lr_format_rows(const double* vals, const int64_t* row_start, int64_t n_rows, uint64_t int_cols,
                                  int32_t flags, char* out, int64_t cap) {
    if (!vals || !row_start || !out) return LR_ERR_NULL;
    if (n_rows < 0) return LR_ERR_SIZE;
    const int64_t n_vals = n_rows > 0 ? row_start[n_rows] - row_start[0] : 0;
    const bool crlf = (flags & LR_FORMAT_CRLF) != 0;     // csv.writer's default line end (DD:236, trend_rate.py:189)
    if (n_vals < 0 || cap < 26 * n_vals + 2 * n_rows) return LR_ERR_WORKSPACE;   // 24 characters at most per number + a separator
    char* p = out;
    for (int64_t i = 0; i < n_rows; ++i) {
        const int64_t a = row_start[i], b = row_start[i + 1];
        for (int64_t j = a; j < b; ++j) {
            const int64_t c = j - a;
            if (c > 0) *p++ = '\t';
            if (c < 64 && ((int_cols >> c) & 1)) p = lr_py_int(p, vals[j]);
            else p = lr_py_float(p, vals[j]);
        }
        if (crlf) *p++ = '\r';
        *p++ = '\n';
    }
    return (int64_t)(p - out);
}
