// Host-side fast path for the feature matrices of a memory_store event file (no GPU call in this file).
//
// The reference reads an event with json.load + np.array(list) (hippomm/core/hippocampal_memory.py:369-395): one Python float
// object per stored fp32, ~20 bytes of text each -- 0.23 s for a 600-frame event, 87 MB/s.  99.9 % of that text is a handful of
// 2-D arrays of numbers.  hmm_json_find_matrices locates them (byte spans + shape) without converting anything;
// hmm_json_parse_matrix_f32 converts one span into a caller-owned fp32 buffer with std::from_chars (correctly rounded, locale-free)
// followed by the double -> float cast, i.e. exactly float32(float64(text)), what np.array(json.load(...)).astype(float32) yields.
// hippomm_amd/event_store.py cuts the spans out of the text, hands the rest (a few KB) to Python's json and puts the matrices back;
// anything this scanner does not recognise is simply not reported and stays on json.load's path.
#include "hmm_common.h"

#include <charconv>
#include <cstdint>
#include <cstring>

namespace {

struct Cursor {
    const char* p;
    const char* end;
    bool done() const { return p >= end; }
};

// json.dump(indent=2) puts every number on its own line behind eight spaces: a third of the text.  Whole words of spaces first.
inline void skip_ws(Cursor& c) {
    for (;;) {
        uint64_t w;
        while (c.end - c.p >= 8 && (std::memcpy(&w, c.p, 8), w == 0x2020202020202020ull)) c.p += 8;
        if (c.p < c.end && (*c.p == ' ' || *c.p == '\n' || *c.p == '\t' || *c.p == '\r')) ++c.p;
        else return;
    }
}

// Eight ASCII digits at once (the SWAR step of simdjson / fast_float, restated): are these 8 bytes all '0'..'9', and their value.
inline bool eight_digits(const char* p, uint64_t* w) {
    std::memcpy(w, p, 8);
    return !((((*w + 0x4646464646464646ull) | (*w - 0x3030303030303030ull)) & 0x8080808080808080ull));
}
inline uint32_t eight_digits_value(uint64_t w) {                    // little-endian: the first character is the lowest byte
    w -= 0x3030303030303030ull;
    w = w * 10 + (w >> 8);                                          // pairs
    w = (((w & 0x000000FF000000FFull) * 0x000F424000000064ull) + (((w >> 16) & 0x000000FF000000FFull) * 0x0000271000000001ull)) >> 32;
    return (uint32_t)w;
}
inline const char* skip_digits(const char* p, const char* end) {
    uint64_t w;
    while (end - p >= 8 && eight_digits(p, &w)) p += 8;
    while (p < end && *p >= '0' && *p <= '9') ++p;
    return p;
}

// One JSON number, or one of the three non-finite literals Python's json writes (NaN, Infinity, -Infinity).  Returns the end of
// the token or nullptr.  Grammar only; nothing is converted.
inline const char* scan_number(const char* p, const char* end) {
    const char* s = p;
    if (p < end && *p == '-') ++p;
    if (end - p >= 8 && std::memcmp(p, "Infinity", 8) == 0) return p + 8;
    if (s == p && end - p >= 3 && std::memcmp(p, "NaN", 3) == 0) return p + 3;
    if (p >= end || *p < '0' || *p > '9') return nullptr;
    if (*p == '0') ++p;
    else p = skip_digits(p, end);
    if (p < end && *p == '.') {
        ++p;
        if (p >= end || *p < '0' || *p > '9') return nullptr;
        p = skip_digits(p, end);
    }
    if (p < end && (*p == 'e' || *p == 'E')) {
        ++p;
        if (p < end && (*p == '+' || *p == '-')) ++p;
        if (p >= end || *p < '0' || *p > '9') return nullptr;
        while (p < end && *p >= '0' && *p <= '9') ++p;
    }
    return p;
}

// `[ n, n, ... ]` with at least one number: returns the position after ']' and the count, or nullptr.
inline const char* scan_row(const char* p, const char* end, size_t* count) {
    Cursor c{p, end};
    if (c.done() || *c.p != '[') return nullptr;
    ++c.p;
    size_t n = 0;
    for (;;) {
        skip_ws(c);
        const char* q = scan_number(c.p, c.end);
        if (!q) return nullptr;
        c.p = q;
        ++n;
        skip_ws(c);
        if (c.done()) return nullptr;
        if (*c.p == ',') { ++c.p; continue; }
        if (*c.p == ']') { ++c.p; break; }
        return nullptr;
    }
    *count = n;
    return c.p;
}

// `[ row, row, ... ]` with equally long rows: returns the position after the outer ']' or nullptr.
inline const char* scan_matrix(const char* p, const char* end, size_t* rows, size_t* cols) {
    Cursor c{p, end};
    if (c.done() || *c.p != '[') return nullptr;
    ++c.p;
    size_t r = 0, width = 0;
    for (;;) {
        skip_ws(c);
        size_t n = 0;
        const char* q = scan_row(c.p, c.end, &n);
        if (!q) return nullptr;
        if (r == 0) width = n;
        else if (n != width) return nullptr;
        c.p = q;
        ++r;
        skip_ws(c);
        if (c.done()) return nullptr;
        if (*c.p == ',') { ++c.p; continue; }
        if (*c.p == ']') { ++c.p; break; }
        return nullptr;
    }
    *rows = r;
    *cols = width;
    return c.p;
}

// Exact powers of ten in double: 10^22 = 2^22 * 5^22 and 5^22 < 2^53.
constexpr double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                               1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// (float)(double)literal without the correctly rounded decimal -> double conversion, when that is provably safe: the literal's
// first 19 significant digits as an integer, scaled by exact powers of ten, give a double within 4 ulp of the true value (one
// rounding of the integer, at most three of the scalings, 1e-18 relative from dropped digits).  If that approximation is more
// than 32 ulp away from both rounding boundaries of its float (the midpoints to the neighbouring floats), every double within
// 4 ulp -- the correctly rounded one included -- casts to the same float.  Doubles that came from fp32 values (what an event file
// holds) sit exactly ON a float, 2^28 ulp from the boundaries; anything closer than 32 ulp to a boundary returns false and takes
// the exact path.
inline bool convert_fast(const char* p, const char* q, float* out) {
    const bool neg = *p == '-';
    if (neg) ++p;
    uint64_t m = 0;
    int digits = 0, exp10 = 0;
    bool seen_point = false;
    for (; p < q; ++p) {
        uint64_t w;
        while (digits <= 11 && q - p >= 8 && eight_digits(p, &w)) {     // eight digits at a time while they all fit into m
            const uint32_t v = eight_digits_value(w);
            if (m == 0) {
                m = v;
                digits = v >= 10000000 ? 8 : v >= 1000000 ? 7 : v >= 100000 ? 6 : v >= 10000 ? 5 : v >= 1000 ? 4 : v >= 100 ? 3 : v >= 10 ? 2 : v ? 1 : 0;
            } else {
                m = m * 100000000ull + v;
                digits += 8;
            }
            if (seen_point) exp10 -= 8;
            p += 8;
        }
        if (p >= q) break;
        const char ch = *p;
        if (ch >= '0' && ch <= '9') {
            if (digits < 19) {
                if (m != 0 || ch != '0') { m = m * 10 + (uint64_t)(ch - '0'); ++digits; }
                if (seen_point) --exp10;
            } else if (!seen_point) {
                ++exp10;                                  // a dropped integer digit
            }
        } else if (ch == '.') {
            seen_point = true;
        } else {
            break;                                        // 'e' / 'E'
        }
    }
    const bool has_exp = p < q;
    if (has_exp) {                                        // exponent
        ++p;
        bool eneg = false;
        if (p < q && (*p == '+' || *p == '-')) { eneg = *p == '-'; ++p; }
        int e = 0;
        for (; p < q; ++p) {
            e = e * 10 + (*p - '0');
            if (e > 10000) return false;
        }
        exp10 += eneg ? -e : e;
    }
    // "-0.0" is a float, "-0" an int: json.load gives 0, which has no sign
    if (m == 0) { *out = neg && (seen_point || has_exp) ? -0.0f : 0.0f; return true; }
    if (exp10 < -66 || exp10 > 66) return false;
    double d = (double)m;
    for (int e = exp10; e > 0; e -= 22) d *= kPow10[e > 22 ? 22 : e];
    for (int e = -exp10; e > 0; e -= 22) d /= kPow10[e > 22 ? 22 : e];
    if (!(d < 3.0e38) || d < 1.0e-60) return false;       // near or past the float range ends: exact path
    const float f = (float)d;                             // 0 <= f < 3e38: its neighbours are the bit patterns +- 1
    uint32_t bits;
    std::memcpy(&bits, &f, 4);
    const uint32_t up_bits = bits + 1, down_bits = bits ? bits - 1 : 0x80000001u;      // below +0: the smallest negative subnormal
    float up, down;
    std::memcpy(&up, &up_bits, 4);
    std::memcpy(&down, &down_bits, 4);
    const double lo = ((double)down + (double)f) * 0.5;
    const double hi = ((double)f + (double)up) * 0.5;
    const double tol = d * (32.0 / 4503599627370496.0);   // 32 ulp of d (2^-52 relative)
    if (!(d - lo > tol && hi - d > tol)) return false;
    *out = neg ? -f : f;
    return true;
}

inline bool convert_number(const char* p, const char* q, float* out) {
    const bool neg = *p == '-';
    const char* s = p + (neg ? 1 : 0);
    if (*s == 'I') { *out = neg ? -__builtin_inff() : __builtin_inff(); return true; }
    if (*s == 'N') { *out = __builtin_nanf(""); return true; }
    if (convert_fast(p, q, out)) return true;
    double d = 0.0;
    const auto r = std::from_chars(p, q, d);
    if (r.ptr != q) return false;
    // a literal outside the double range (never produced from fp32 features): not converted here -- the caller falls back to
    // json.load for the whole file, which knows Python's rules for it
    if (r.ec != std::errc()) return false;
    *out = (float)d;
    return true;
}

}  // namespace

using namespace hmm;

// Every `[[numbers], [numbers], ...]` (equally long rows) of at least `min_values` numbers in text[0, len), outside string
// literals, in document order.  out[i] = {begin, end, rows, cols} (byte span of the outer brackets); at most `cap` are written,
// *n_found counts all of them (call again with a larger `out` if *n_found > cap).
extern "C" int hmm_json_find_matrices(const char* text, size_t len, size_t min_values, hmm_json_matrix* out, int cap, int* n_found) {
    HMM_REQUIRE(text && n_found && (out || cap == 0) && cap >= 0, HMM_E_INVALID, "json_find_matrices: null pointer");
    const char* p = text;
    const char* end = text + len;
    int found = 0;
    while (p < end) {
        const char ch = *p;
        if (ch == '"') {                                   // string literal: skip to the closing quote
            ++p;
            while (p < end && *p != '"') p += (*p == '\\' && p + 1 < end) ? 2 : 1;
            ++p;
            continue;
        }
        if (ch == '[') {
            const char* q = p + 1;
            while (q < end && (*q == ' ' || *q == '\n' || *q == '\t' || *q == '\r')) ++q;
            if (q < end && *q == '[') {
                size_t rows = 0, cols = 0;
                const char* after = scan_matrix(p, end, &rows, &cols);
                if (after && rows * cols >= min_values) {
                    if (found < cap) out[found] = hmm_json_matrix{(size_t)(p - text), (size_t)(after - text), rows, cols};
                    ++found;
                    p = after;
                    continue;
                }
            }
        }
        ++p;
    }
    *n_found = found;
    return HMM_OK;
}

// The matrix hmm_json_find_matrices reported at [begin, end) of `text`, as rows x cols fp32 (row-major) into out_host:
// out[r][c] = (float)(double)literal.  HMM_E_INVALID when the span is not such a matrix.
extern "C" int hmm_json_parse_matrix_f32(const char* text, size_t begin, size_t end, size_t rows, size_t cols, float* out_host) {
    HMM_REQUIRE(text && out_host && begin < end, HMM_E_INVALID, "json_parse_matrix: null pointer or empty span");
    Cursor c{text + begin, text + end};
    const size_t total = rows * cols;
    size_t k = 0;
    int depth = 0;
    for (;;) {
        skip_ws(c);
        if (c.done()) break;
        const char ch = *c.p;
        if (ch == ',') { ++c.p; continue; }
        if (ch == '[') { ++depth; ++c.p; continue; }
        if (ch == ']') { --depth; ++c.p; if (depth == 0) break; continue; }
        const char* q = scan_number(c.p, c.end);
        HMM_REQUIRE(q && depth == 2 && k < total, HMM_E_INVALID, "json_parse_matrix: not a %zu x %zu matrix of numbers at byte %zu",
                    rows, cols, (size_t)(c.p - text));
        HMM_REQUIRE(convert_number(c.p, q, out_host + k), HMM_E_INVALID, "json_parse_matrix: bad number at byte %zu", (size_t)(c.p - text));
        ++k;
        c.p = q;
    }
    HMM_REQUIRE(k == total && depth == 0, HMM_E_INVALID, "json_parse_matrix: %zu numbers in the span, expected %zu x %zu", k, rows, cols);
    return HMM_OK;
}

// ---- writer: the text json.dumps(rows, indent=2) gives a list of equally long lists of floats, nested `close_indent` deep ----
namespace {

// float.__repr__ (and therefore json.dumps) of a finite double: the shortest digit string that round-trips (std::to_chars gives the
// same digits as Python's dtoa mode 0), in fixed notation when -4 < decimal-point position <= 16, else d[.ddd]e+XX; "NaN" /
// "Infinity" / "-Infinity" as json.dumps spells them.  Returns the end of the text.
inline char* py_float_repr(double x, char* out) {
    if (x != x) { std::memcpy(out, "NaN", 3); return out + 3; }
    if (x < 0 || (x == 0 && __builtin_signbit(x))) { *out++ = '-'; x = -x; }
    if (x == __builtin_inf()) { std::memcpy(out, "Infinity", 8); return out + 8; }
    if (x == 0) { std::memcpy(out, "0.0", 3); return out + 3; }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof(sci), x, std::chars_format::scientific);      // d[.ddd]e[+-]XX
    char digits[24];
    int n = 0;
    const char* p = sci;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[n++] = *p;
    ++p;                                                       // past 'e'
    const bool eneg = *p == '-';
    ++p;
    int e10 = 0;
    for (; p < r.ptr; ++p) e10 = e10 * 10 + (*p - '0');
    const int decpt = (eneg ? -e10 : e10) + 1;                 // value = 0.d1d2... x 10^decpt
    if (decpt <= -4 || decpt > 16) {
        *out++ = digits[0];
        if (n > 1) { *out++ = '.'; std::memcpy(out, digits + 1, n - 1); out += n - 1; }
        *out++ = 'e';
        int e = decpt - 1;
        *out++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        if (e >= 100) { *out++ = (char)('0' + e / 100); e %= 100; *out++ = (char)('0' + e / 10); *out++ = (char)('0' + e % 10); }
        else { *out++ = (char)('0' + e / 10); *out++ = (char)('0' + e % 10); }
        return out;
    }
    if (decpt <= 0) {
        *out++ = '0'; *out++ = '.';
        for (int i = 0; i < -decpt; ++i) *out++ = '0';
        std::memcpy(out, digits, n);
        return out + n;
    }
    if (decpt >= n) {
        std::memcpy(out, digits, n); out += n;
        for (int i = n; i < decpt; ++i) *out++ = '0';
        *out++ = '.'; *out++ = '0';
        return out;
    }
    std::memcpy(out, digits, decpt); out += decpt;
    *out++ = '.';
    std::memcpy(out, digits + decpt, n - decpt);
    return out + (n - decpt);
}

inline char* put_spaces(char* out, int n) { std::memset(out, ' ', (size_t)n); return out + n; }

}  // namespace

// Upper bound of the bytes hmm_json_write_matrix_f64 writes.
extern "C" size_t hmm_json_matrix_text_bound(size_t rows, size_t cols, int close_indent) {
    const size_t ind = close_indent > 0 ? (size_t)close_indent : 0;
    return 8 + ind + rows * (2 * (ind + 2) + 8 + cols * (26 + ind + 4 + 2));
}

// rows x cols doubles (row-major, host) as the text json.dumps(m.tolist(), indent=2) produces for that list when its closing bracket
// sits at `close_indent` spaces (rows at +2, values at +4): byte for byte what save_theta_event writes
// (hippomm/core/hippocampal_memory.py:331-335) for a feature matrix.  rows >= 1, cols >= 1.
extern "C" int hmm_json_write_matrix_f64(const double* m, size_t rows, size_t cols, int close_indent, char* out_text, size_t cap,
                                         size_t* written) {
    HMM_REQUIRE(m && out_text && written && rows >= 1 && cols >= 1 && close_indent >= 0, HMM_E_INVALID, "json_write_matrix: bad arguments");
    HMM_REQUIRE(cap >= hmm_json_matrix_text_bound(rows, cols, close_indent), HMM_E_INVALID, "json_write_matrix: buffer of %zu bytes, need %zu",
                cap, hmm_json_matrix_text_bound(rows, cols, close_indent));
    char* o = out_text;
    *o++ = '[';
    for (size_t r = 0; r < rows; ++r) {
        if (r) *o++ = ',';
        *o++ = '\n';
        o = put_spaces(o, close_indent + 2);
        *o++ = '[';
        for (size_t c = 0; c < cols; ++c) {
            if (c) *o++ = ',';
            *o++ = '\n';
            o = put_spaces(o, close_indent + 4);
            o = py_float_repr(m[r * cols + c], o);
        }
        *o++ = '\n';
        o = put_spaces(o, close_indent + 2);
        *o++ = ']';
    }
    *o++ = '\n';
    o = put_spaces(o, close_indent);
    *o++ = ']';
    *written = (size_t)(o - out_text);
    return HMM_OK;
}
