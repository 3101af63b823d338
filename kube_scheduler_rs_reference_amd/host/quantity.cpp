#include "quantity.hpp"

#include <cctype>
#include <limits>

namespace ksched_host {

namespace {

// v *= 10^e with overflow detection (one multiplication: the powers up to 10^38 fit 128 bits)
bool scale10(__int128 &v, int e) {
    static const struct Pow10 {
        __int128 p[39];
        Pow10() {
            p[0] = 1;
            for (int i = 1; i < 39; ++i) p[i] = p[i - 1] * 10;
        }
    } t;
    if (e <= 0) return true;
    if (e > 38) return v == 0;
    return !__builtin_mul_overflow(v, t.p[e], &v);
}
// v /= 10^e, exactly: false when a non-zero digit would be dropped
bool unscale10(__int128 &v, int e) {
    for (; e > 18; e -= 18) {
        if (v % (__int128)1000000000000000000ll != 0) return false;
        v /= (__int128)1000000000000000000ll;
    }
    static const int64_t p[19] = {1ll, 10ll, 100ll, 1000ll, 10000ll, 100000ll, 1000000ll, 10000000ll, 100000000ll, 1000000000ll, 10000000000ll, 100000000000ll,
                                  1000000000000ll, 10000000000000ll, 100000000000000ll, 1000000000000000ll, 10000000000000000ll, 100000000000000000ll, 1000000000000000000ll};
    if (v % (__int128)p[e] != 0) return false;
    v /= (__int128)p[e];
    return true;
}

int decimal_suffix(char c, bool &ok) {
    ok = true;
    switch (c) {
        case 'n': return -9;
        case 'u': return -6;
        case 'm': return -3;
        case 'k': return 3;
        case 'M': return 6;
        case 'G': return 9;
        case 'T': return 12;
        case 'P': return 15;
        case 'E': return 18;
    }
    ok = false;
    return 0;
}

int binary_suffix(char c) {
    switch (c) {
        case 'K': return 10;
        case 'M': return 20;
        case 'G': return 30;
        case 'T': return 40;
        case 'P': return 50;
        case 'E': return 60;
    }
    return 0;
}

}  // namespace

// Kubernetes apimachinery grammar: <sign>? digits ('.' digits)? suffix, suffix one of
// "" n u m k M G T P E | Ki Mi Gi Ti Pi Ei | e<exp> E<exp>
ParsedQuantity ParsedQuantity::try_from(const std::string &text) {
    const auto bad = [&](const char *why) { return QuantityError("invalid quantity '" + text + "': " + why); };
    size_t i = 0;
    bool neg = false;
    if (i < text.size() && (text[i] == '+' || text[i] == '-')) neg = text[i++] == '-';
    __int128 mant = 0;
    uint64_t m64 = 0;  // the mantissa while it has at most 18 digits (no overflow possible): nearly every quantity of a real spec
    int digits = 0, frac = 0;
    auto eat = [&](bool fractional) {
        while (i < text.size() && (unsigned)(text[i] - '0') < 10u) {
            const unsigned d = (unsigned)(text[i] - '0');
            if (digits < 18) {
                m64 = m64 * 10u + d;
            } else {
                if (digits == 18) mant = (__int128)m64;
                if (__builtin_mul_overflow(mant, (__int128)10, &mant) || __builtin_add_overflow(mant, (__int128)d, &mant)) throw bad("mantissa too large");
            }
            ++digits;
            if (fractional) ++frac;
            ++i;
        }
    };
    eat(false);
    if (i < text.size() && text[i] == '.') {
        ++i;
        eat(true);
    }
    if (digits <= 18) mant = (__int128)m64;
    if (digits == 0) throw bad("no digits");
    const char *const suf = text.data() + i;  // (no copy of the suffix)
    const size_t nsuf = text.size() - i;
    auto isdig = [](char c) { return (unsigned)(c - '0') < 10u; };
    int exp10 = 0, shift = 0;
    if (nsuf == 0) {
    } else if ((suf[0] == 'e' || suf[0] == 'E') && nsuf > 1 && (isdig(suf[1]) || suf[1] == '+' || suf[1] == '-')) {
        size_t j = 1;
        bool eneg = false;
        if (suf[j] == '+' || suf[j] == '-') eneg = suf[j++] == '-';
        if (j >= nsuf) throw bad("empty exponent");
        int ev = 0;
        for (; j < nsuf; ++j) {
            if (!isdig(suf[j])) throw bad("bad exponent");
            ev = ev * 10 + (suf[j] - '0');
            if (ev > 100) throw bad("exponent too large");
        }
        exp10 = eneg ? -ev : ev;
    } else if (nsuf == 2 && suf[1] == 'i') {
        shift = binary_suffix(suf[0]);
        if (!shift) throw bad("unknown binary suffix");
    } else if (nsuf == 1) {
        bool ok;
        exp10 = decimal_suffix(suf[0], ok);
        if (!ok) throw bad("unknown suffix");
    } else {
        throw bad("unknown suffix");
    }
    __int128 v = mant;
    if (shift && __builtin_mul_overflow(v, ((__int128)1) << shift, &v)) throw bad("out of range");
    const int scale = 9 + exp10 - frac;
    if (scale > 0 && !scale10(v, scale)) throw bad("out of range");
    if (scale < 0 && v != 0 && !unscale10(v, -scale)) throw bad("finer than one nano-unit");
    ParsedQuantity q;
    q.nanos_ = neg ? -v : v;
    return q;
}

int64_t ParsedQuantity::to_milli() const {
    if (nanos_ % 1000000 != 0) throw QuantityError("quantity is not a whole number of milli-units");
    const __int128 m = nanos_ / 1000000;
    if (m > std::numeric_limits<int64_t>::max() || m < std::numeric_limits<int64_t>::min()) throw QuantityError("milli-units overflow int64");
    return (int64_t)m;
}

int64_t ParsedQuantity::to_units() const {
    if (nanos_ % 1000000000 != 0) throw QuantityError("quantity is not a whole number of units");
    const __int128 m = nanos_ / 1000000000;
    if (m > std::numeric_limits<int64_t>::max() || m < std::numeric_limits<int64_t>::min()) throw QuantityError("units overflow int64");
    return (int64_t)m;
}

}  // namespace ksched_host
