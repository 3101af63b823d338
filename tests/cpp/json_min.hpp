// json_min.hpp -- a minimal JSON reader for the C++ test tools (tests only; the product has no JSON in it).
// Parses objects / arrays / strings (with the common escapes) / numbers (kept as text + double) / true / false / null.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace jmin {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;  // String: the text; Number: the literal
    std::vector<Value> arr;
    std::map<std::string, Value> obj;

    bool has(const std::string &k) const { return kind == Object && obj.count(k) != 0; }
    const Value &at(const std::string &k) const {
        auto it = obj.find(k);
        if (kind != Object || it == obj.end()) throw std::runtime_error("json: missing key " + k);
        return it->second;
    }
    const Value *get(const std::string &k) const {
        if (kind != Object) return nullptr;
        auto it = obj.find(k);
        return (it == obj.end() || it->second.kind == Null) ? nullptr : &it->second;
    }
};

class Parser {
public:
    explicit Parser(const std::string &s) : s_(s) {}
    Value parse() {
        Value v = value();
        ws();
        if (i_ != s_.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string &s_;
    size_t i_ = 0;
    [[noreturn]] void fail(const char *why) const { throw std::runtime_error(std::string("json: ") + why + " at offset " + std::to_string(i_)); }
    void ws() {
        while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) ++i_;
    }
    bool eat(char c) {
        ws();
        if (i_ < s_.size() && s_[i_] == c) {
            ++i_;
            return true;
        }
        return false;
    }
    std::string string() {
        if (!eat('"')) fail("expected string");
        std::string out;
        while (i_ < s_.size() && s_[i_] != '"') {
            char c = s_[i_++];
            if (c == '\\') {
                if (i_ >= s_.size()) fail("bad escape");
                char e = s_[i_++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {  // \uXXXX: ASCII range only (the fixtures are ASCII)
                        if (i_ + 4 > s_.size()) fail("bad \\u escape");
                        out += (char)std::strtol(s_.substr(i_, 4).c_str(), nullptr, 16);
                        i_ += 4;
                        break;
                    }
                    default: out += e;
                }
            } else {
                out += c;
            }
        }
        if (i_ >= s_.size()) fail("unterminated string");
        ++i_;
        return out;
    }
    Value value() {
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        Value v;
        const char c = s_[i_];
        if (c == '{') {
            ++i_;
            v.kind = Value::Object;
            if (eat('}')) return v;
            do {
                std::string k = string();
                if (!eat(':')) fail("expected ':'");
                v.obj.emplace(std::move(k), value());
            } while (eat(','));
            if (!eat('}')) fail("expected '}'");
        } else if (c == '[') {
            ++i_;
            v.kind = Value::Array;
            if (eat(']')) return v;
            do v.arr.push_back(value());
            while (eat(','));
            if (!eat(']')) fail("expected ']'");
        } else if (c == '"') {
            v.kind = Value::String;
            v.str = string();
        } else if (s_.compare(i_, 4, "true") == 0) {
            v.kind = Value::Bool, v.b = true, i_ += 4;
        } else if (s_.compare(i_, 5, "false") == 0) {
            v.kind = Value::Bool, i_ += 5;
        } else if (s_.compare(i_, 4, "null") == 0) {
            i_ += 4;
        } else {
            const size_t b = i_;
            while (i_ < s_.size() && (std::isdigit((unsigned char)s_[i_]) || s_[i_] == '-' || s_[i_] == '+' || s_[i_] == '.' || s_[i_] == 'e' || s_[i_] == 'E')) ++i_;
            if (b == i_) fail("unexpected character");
            v.kind = Value::Number;
            v.str = s_.substr(b, i_ - b);
            v.num = std::strtod(v.str.c_str(), nullptr);
        }
        return v;
    }
};

inline Value parse(const std::string &text) { return Parser(text).parse(); }

}  // namespace jmin
