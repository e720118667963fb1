// mini_json.hpp — the little JSON the CLI needs (objects, arrays, strings, integers, booleans,
// null): a recursive-descent parser and a compact serializer.  Object keys keep insertion order.
#pragma once
#include <stdint.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mjson {

struct Value;
typedef std::shared_ptr<Value> Ptr;

struct Value {
  enum Kind { Null, Bool, Int, Double, String, Array, Object } kind = Null;
  bool b = false;
  int64_t i = 0;
  double d = 0.0;
  std::string s;
  std::vector<Ptr> a;
  std::vector<std::pair<std::string, Ptr>> o;

  const Ptr* find(const std::string& key) const {
    for (auto& kv : o) if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

inline Ptr make(Value::Kind k) { auto v = std::make_shared<Value>(); v->kind = k; return v; }
inline Ptr integer(int64_t x) { auto v = make(Value::Int); v->i = x; return v; }
inline Ptr string(const std::string& x) { auto v = make(Value::String); v->s = x; return v; }

class Parser {
 public:
  explicit Parser(const std::string& text) : t_(text), p_(0) {}
  Ptr parse() {
    Ptr v = value();
    ws();
    if (p_ != t_.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& t_;
  size_t p_;
  [[noreturn]] void fail(const std::string& m) const {
    throw std::runtime_error("JSON: " + m + " at offset " + std::to_string(p_));
  }
  void ws() { while (p_ < t_.size() && (t_[p_] == ' ' || t_[p_] == '\n' || t_[p_] == '\t' || t_[p_] == '\r')) ++p_; }
  bool eat(char c) { ws(); if (p_ < t_.size() && t_[p_] == c) { ++p_; return true; } return false; }
  Ptr value() {
    ws();
    if (p_ >= t_.size()) fail("unexpected end");
    char c = t_[p_];
    if (c == '{') return object();
    if (c == '[') return array();
    if (c == '"') { auto v = make(Value::String); v->s = str(); return v; }
    if (t_.compare(p_, 4, "true") == 0) { p_ += 4; auto v = make(Value::Bool); v->b = true; return v; }
    if (t_.compare(p_, 5, "false") == 0) { p_ += 5; return make(Value::Bool); }
    if (t_.compare(p_, 4, "null") == 0) { p_ += 4; return make(Value::Null); }
    return number();
  }
  Ptr number() {
    size_t b = p_;
    bool is_double = false;
    if (p_ < t_.size() && (t_[p_] == '-' || t_[p_] == '+')) ++p_;
    while (p_ < t_.size() && ((t_[p_] >= '0' && t_[p_] <= '9') || t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E' ||
                              t_[p_] == '-' || t_[p_] == '+')) {
      if (t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E') is_double = true;
      ++p_;
    }
    if (b == p_) fail("unexpected character");
    const std::string tok = t_.substr(b, p_ - b);
    if (is_double) { auto v = make(Value::Double); v->d = std::stod(tok); return v; }
    auto v = make(Value::Int);
    v->i = std::stoll(tok);
    return v;
  }
  std::string str() {
    if (t_[p_] != '"') fail("expected string");
    ++p_;
    std::string out;
    while (p_ < t_.size() && t_[p_] != '"') {
      char c = t_[p_++];
      if (c != '\\') { out.push_back(c); continue; }
      if (p_ >= t_.size()) fail("bad escape");
      char e = t_[p_++];
      switch (e) {
        case 'n': out.push_back('\n'); break;
        case 't': out.push_back('\t'); break;
        case 'r': out.push_back('\r'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'u': {
          if (p_ + 4 > t_.size()) fail("bad \\u escape");
          unsigned cp = (unsigned)std::stoul(t_.substr(p_, 4), nullptr, 16);
          p_ += 4;
          if (cp < 0x80) out.push_back((char)cp);
          else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3f))); }
          else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3f))); out.push_back((char)(0x80 | (cp & 0x3f))); }
          break;
        }
        default: out.push_back(e);
      }
    }
    if (p_ >= t_.size()) fail("unterminated string");
    ++p_;
    return out;
  }
  Ptr array() {
    ++p_;
    auto v = make(Value::Array);
    if (eat(']')) return v;
    for (;;) {
      v->a.push_back(value());
      if (eat(']')) return v;
      if (!eat(',')) fail("expected , or ]");
    }
  }
  Ptr object() {
    ++p_;
    auto v = make(Value::Object);
    if (eat('}')) return v;
    for (;;) {
      ws();
      std::string k = str();
      if (!eat(':')) fail("expected :");
      v->o.emplace_back(k, value());
      if (eat('}')) return v;
      if (!eat(',')) fail("expected , or }");
    }
  }
};

inline void escape(const std::string& s, std::string& out) {
  out.push_back('"');
  for (char c : s) {
    switch (c) {
      case '"': out += "\\\""; break;
      case '\\': out += "\\\\"; break;
      case '\n': out += "\\n"; break;
      case '\t': out += "\\t"; break;
      case '\r': out += "\\r"; break;
      default: out.push_back(c);
    }
  }
  out.push_back('"');
}

inline void dump(const Ptr& v, std::string& out) {
  switch (v->kind) {
    case Value::Null: out += "null"; break;
    case Value::Bool: out += v->b ? "true" : "false"; break;
    case Value::Int: out += std::to_string(v->i); break;
    case Value::Double: out += std::to_string(v->d); break;
    case Value::String: escape(v->s, out); break;
    case Value::Array:
      out.push_back('[');
      for (size_t k = 0; k < v->a.size(); ++k) { if (k) out.push_back(','); dump(v->a[k], out); }
      out.push_back(']');
      break;
    case Value::Object:
      out.push_back('{');
      for (size_t k = 0; k < v->o.size(); ++k) {
        if (k) out.push_back(',');
        escape(v->o[k].first, out);
        out.push_back(':');
        dump(v->o[k].second, out);
      }
      out.push_back('}');
      break;
  }
}

inline std::string dump(const Ptr& v) { std::string s; dump(v, s); return s; }

}  // namespace mjson
