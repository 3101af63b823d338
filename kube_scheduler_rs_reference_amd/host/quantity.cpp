#include "quantity.hpp"

#include <cctype>
#include <limits>

namespace ksched_host {

namespace {

// 10^e as a 128-bit integer with overflow detection
bool scale10(__int128 &v, int e) {
    for (; e > 0; --e)
        if (__builtin_mul_overflow(v, (__int128)10, &v)) return false;
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
    int digits = 0, frac = 0;
    auto eat = [&](bool fractional) {
        while (i < text.size() && std::isdigit((unsigned char)text[i])) {
            if (__builtin_mul_overflow(mant, (__int128)10, &mant) || __builtin_add_overflow(mant, (__int128)(text[i] - '0'), &mant))
                throw bad("mantissa too large");
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
    if (digits == 0) throw bad("no digits");
    const std::string suf = text.substr(i);
    int exp10 = 0, shift = 0;
    if (suf.empty()) {
    } else if ((suf[0] == 'e' || suf[0] == 'E') && suf.size() > 1 && (std::isdigit((unsigned char)suf[1]) || suf[1] == '+' || suf[1] == '-')) {
        size_t j = 1;
        bool eneg = false;
        if (suf[j] == '+' || suf[j] == '-') eneg = suf[j++] == '-';
        if (j >= suf.size()) throw bad("empty exponent");
        int ev = 0;
        for (; j < suf.size(); ++j) {
            if (!std::isdigit((unsigned char)suf[j])) throw bad("bad exponent");
            ev = ev * 10 + (suf[j] - '0');
            if (ev > 100) throw bad("exponent too large");
        }
        exp10 = eneg ? -ev : ev;
    } else if (suf.size() == 2 && suf[1] == 'i') {
        shift = binary_suffix(suf[0]);
        if (!shift) throw bad("unknown binary suffix");
    } else if (suf.size() == 1) {
        bool ok;
        exp10 = decimal_suffix(suf[0], ok);
        if (!ok) throw bad("unknown suffix");
    } else {
        throw bad("unknown suffix");
    }
    __int128 v = mant;
    if (shift && __builtin_mul_overflow(v, ((__int128)1) << shift, &v)) throw bad("out of range");
    int scale = 9 + exp10 - frac;
    if (scale > 0 && !scale10(v, scale)) throw bad("out of range");
    for (; scale < 0; ++scale) {
        if (v % 10 != 0) throw bad("finer than one nano-unit");
        v /= 10;
    }
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
