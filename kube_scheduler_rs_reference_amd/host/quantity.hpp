// quantity.hpp -- exact Kubernetes resource.Quantity values for the host encoder.
//
// Plays the role `kube_quantity::ParsedQuantity` (0.6.1, Cargo.lock:787-797) plays in the
// reference: TryFrom<&str> / TryFrom<&Quantity> (src/util.rs:25,65; src/predicates.rs:29),
// AddAssign / SubAssign (src/util.rs:33,65) and PartialOrd (src/predicates.rs:42).  Values are
// exact nano-units in a 128-bit integer; every quantity of the canonical domain D (SURVEY.md
// section 8c) is an integer number of milli-cores / bytes, which is what the device compares.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>

namespace ksched_host {

struct QuantityError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class ParsedQuantity {
public:
    ParsedQuantity() = default;
    // text -> value; throws QuantityError where the reference's `.expect(...)` would panic
    static ParsedQuantity try_from(const std::string &text);

    ParsedQuantity &operator+=(const ParsedQuantity &o) { nanos_ += o.nanos_; return *this; }
    ParsedQuantity &operator-=(const ParsedQuantity &o) { nanos_ -= o.nanos_; return *this; }
    bool operator<=(const ParsedQuantity &o) const { return nanos_ <= o.nanos_; }
    bool operator==(const ParsedQuantity &o) const { return nanos_ == o.nanos_; }

    // exact integer views for the device columns; throw QuantityError if the value is not an
    // integer number of milli-units / units or does not fit in int64
    int64_t to_milli() const;
    int64_t to_units() const;
    __int128 nanos() const { return nanos_; }

private:
    __int128 nanos_ = 0;
};

}  // namespace ksched_host
