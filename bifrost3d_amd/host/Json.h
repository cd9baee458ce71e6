// Json.h -- a small JSON document model and recursive-descent parser (RFC 8259), what the glTF loader reads files with.
//
// The reference reads glTF through two vendored third-party headers (extensions/glTFLoader/glTFLoader/tiny_gltf.h and json.hpp);
// this is the repository's own replacement for the JSON half. Numbers are doubles, objects keep insertion order, lookups of
// missing members return a shared null value so that chains such as doc["a"]["b"].as_int(-1) never fault.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

namespace Json {

class Value {
public:
    enum class Type { Null, Bool, Number, String, Array, Object };

    Value() = default;
    Type type() const { return m_type; }
    bool is_null() const { return m_type == Type::Null; }
    bool is_number() const { return m_type == Type::Number; }
    bool is_string() const { return m_type == Type::String; }
    bool is_array() const { return m_type == Type::Array; }
    bool is_object() const { return m_type == Type::Object; }

    double as_double(double fallback = 0.0) const { return m_type == Type::Number ? m_number : fallback; }
    int as_int(int fallback = 0) const { return m_type == Type::Number ? int(m_number) : fallback; }
    bool as_bool(bool fallback = false) const { return m_type == Type::Bool ? m_number != 0.0 : fallback; }
    const std::string& as_string() const { static const std::string empty; return m_type == Type::String ? m_string : empty; }

    size_t size() const { return m_type == Type::Array ? m_elements.size() : (m_type == Type::Object ? m_members.size() : 0); }
    const Value& operator[](size_t index) const { return m_type == Type::Array && index < m_elements.size() ? m_elements[index] : null(); }
    const Value& operator[](const char* key) const {
        if (m_type == Type::Object)
            for (const auto& member : m_members)
                if (member.first == key) return member.second;
        return null();
    }
    bool has(const char* key) const { return !(*this)[key].is_null(); }
    const std::vector<Value>& elements() const { return m_elements; }
    const std::vector<std::pair<std::string, Value>>& members() const { return m_members; }

    static const Value& null() { static const Value v; return v; }

    // Parses a whole document; on failure returns false and describes the first problem in `error`.
    static bool parse(const char* begin, const char* end, Value& out, std::string& error) {
        Parser p{begin, end, begin, error};
        if (!p.value(out, 0)) return false;
        p.skip_whitespace();
        if (p.at != p.end) return p.fail("trailing characters after the document");
        return true;
    }

private:
    Type m_type = Type::Null;
    double m_number = 0.0;
    std::string m_string;
    std::vector<Value> m_elements;
    std::vector<std::pair<std::string, Value>> m_members;

    struct Parser {
        const char* begin; const char* end; const char* at; std::string& error;

        bool fail(const char* what) { error = std::string(what) + " at byte " + std::to_string(at - begin); return false; }
        void skip_whitespace() { while (at != end && (*at == ' ' || *at == '\t' || *at == '\n' || *at == '\r')) ++at; }
        bool literal(const char* text) {
            const char* p = at;
            for (; *text; ++text, ++p) if (p == end || *p != *text) return false;
            at = p;
            return true;
        }

        static void append_utf8(std::string& s, uint32_t cp) {
            if (cp < 0x80) s += char(cp);
            else if (cp < 0x800) { s += char(0xC0 | (cp >> 6)); s += char(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) { s += char(0xE0 | (cp >> 12)); s += char(0x80 | ((cp >> 6) & 0x3F)); s += char(0x80 | (cp & 0x3F)); }
            else { s += char(0xF0 | (cp >> 18)); s += char(0x80 | ((cp >> 12) & 0x3F)); s += char(0x80 | ((cp >> 6) & 0x3F)); s += char(0x80 | (cp & 0x3F)); }
        }
        bool hex4(uint32_t& out) {
            out = 0;
            for (int i = 0; i < 4; ++i, ++at) {
                if (at == end) return false;
                const char c = *at;
                out <<= 4;
                if (c >= '0' && c <= '9') out |= uint32_t(c - '0');
                else if (c >= 'a' && c <= 'f') out |= uint32_t(c - 'a' + 10);
                else if (c >= 'A' && c <= 'F') out |= uint32_t(c - 'A' + 10);
                else return false;
            }
            return true;
        }
        bool string(std::string& out) {
            ++at;   // opening quote
            out.clear();
            while (true) {
                if (at == end) return fail("unterminated string");
                const char c = *at++;
                if (c == '"') return true;
                if ((unsigned char)c < 0x20) return fail("control character in string");
                if (c != '\\') { out += c; continue; }
                if (at == end) return fail("unterminated escape");
                const char e = *at++;
                switch (e) {
                case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
                case 'r': out += '\r'; break; case 't': out += '\t'; break;
                case 'u': {
                    uint32_t cp;
                    if (!hex4(cp)) return fail("bad \\u escape");
                    if (cp >= 0xD800 && cp < 0xDC00 && end - at >= 6 && at[0] == '\\' && at[1] == 'u') {   // surrogate pair
                        at += 2;
                        uint32_t low;
                        if (!hex4(low) || low < 0xDC00 || low > 0xDFFF) return fail("bad surrogate pair");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (low - 0xDC00);
                    }
                    append_utf8(out, cp);
                    break;
                }
                default: return fail("unknown escape");
                }
            }
        }
        bool number(Value& out) {
            const char* start = at;
            if (at != end && *at == '-') ++at;
            if (at == end || *at < '0' || *at > '9') return fail("malformed number");
            if (*at == '0') ++at; else while (at != end && *at >= '0' && *at <= '9') ++at;
            if (at != end && *at == '.') {
                ++at;
                if (at == end || *at < '0' || *at > '9') return fail("malformed fraction");
                while (at != end && *at >= '0' && *at <= '9') ++at;
            }
            if (at != end && (*at == 'e' || *at == 'E')) {
                ++at;
                if (at != end && (*at == '+' || *at == '-')) ++at;
                if (at == end || *at < '0' || *at > '9') return fail("malformed exponent");
                while (at != end && *at >= '0' && *at <= '9') ++at;
            }
            out.m_type = Type::Number;
            out.m_number = std::strtod(std::string(start, at).c_str(), nullptr);
            return true;
        }
        bool value(Value& out, int depth) {
            if (depth > 256) return fail("document nested too deeply");
            skip_whitespace();
            if (at == end) return fail("unexpected end of document");
            const char c = *at;
            if (c == '{') {
                ++at;
                out.m_type = Type::Object;
                skip_whitespace();
                if (at != end && *at == '}') { ++at; return true; }
                while (true) {
                    skip_whitespace();
                    if (at == end || *at != '"') return fail("expected a member name");
                    std::string key;
                    if (!string(key)) return false;
                    skip_whitespace();
                    if (at == end || *at != ':') return fail("expected ':'");
                    ++at;
                    out.m_members.emplace_back(std::move(key), Value());
                    if (!value(out.m_members.back().second, depth + 1)) return false;
                    skip_whitespace();
                    if (at != end && *at == ',') { ++at; continue; }
                    if (at != end && *at == '}') { ++at; return true; }
                    return fail("expected ',' or '}'");
                }
            }
            if (c == '[') {
                ++at;
                out.m_type = Type::Array;
                skip_whitespace();
                if (at != end && *at == ']') { ++at; return true; }
                while (true) {
                    out.m_elements.emplace_back();
                    if (!value(out.m_elements.back(), depth + 1)) return false;
                    skip_whitespace();
                    if (at != end && *at == ',') { ++at; continue; }
                    if (at != end && *at == ']') { ++at; return true; }
                    return fail("expected ',' or ']'");
                }
            }
            if (c == '"') { out.m_type = Type::String; return string(out.m_string); }
            if (literal("true")) { out.m_type = Type::Bool; out.m_number = 1.0; return true; }
            if (literal("false")) { out.m_type = Type::Bool; out.m_number = 0.0; return true; }
            if (literal("null")) { out.m_type = Type::Null; return true; }
            return number(out);
        }
    };
};

} // namespace Json
