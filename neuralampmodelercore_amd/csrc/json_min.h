// json_min.h — a small, dependency-free JSON reader for .nam files.
//
// The reference parses .nam files with nlohmann::json (NAM/nam_file.cpp:19-27). This host
// library must stand alone on the GPU box, so it carries its own reader: a recursive-descent
// parser into an immutable tree. Numbers are kept as double (weights are narrowed to float by
// the loader exactly as `std::vector<float> weights = j["weights"]` does, NAM/get_dsp.cpp:130-139).
#pragma once

#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace namhip
{
namespace json
{

struct ParseError : std::runtime_error
{
  explicit ParseError(const std::string& m)
  : std::runtime_error(m)
  {
  }
};

class Value
{
public:
  enum Type
  {
    Null,
    Bool,
    Number,
    String,
    Array,
    Object
  };

  Type type = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Value> arr;
  // insertion order does not matter for .nam files; keep lookups simple
  std::vector<std::pair<std::string, Value>> obj;

  bool is_null() const { return type == Null; }
  bool is_bool() const { return type == Bool; }
  bool is_number() const { return type == Number; }
  bool is_string() const { return type == String; }
  bool is_array() const { return type == Array; }
  bool is_object() const { return type == Object; }

  const Value* find(const std::string& key) const
  {
    if (type != Object)
      return nullptr;
    for (const auto& kv : obj)
      if (kv.first == key)
        return &kv.second;
    return nullptr;
  }
  bool contains(const std::string& key) const { return find(key) != nullptr; }

  const Value& at(const std::string& key) const
  {
    const Value* v = find(key);
    if (!v)
      throw std::runtime_error("JSON: missing key \"" + key + "\"");
    return *v;
  }
  const Value& at(size_t i) const
  {
    if (type != Array || i >= arr.size())
      throw std::runtime_error("JSON: array index out of range");
    return arr[i];
  }
  size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }

  double as_double() const
  {
    if (type == Number)
      return num;
    if (type == Bool)
      return b ? 1.0 : 0.0;
    throw std::runtime_error("JSON: value is not a number");
  }
  int as_int() const { return (int)as_double(); }
  bool as_bool() const
  {
    if (type == Bool)
      return b;
    if (type == Number)
      return num != 0.0;
    throw std::runtime_error("JSON: value is not a boolean");
  }
  const std::string& as_string() const
  {
    if (type != String)
      throw std::runtime_error("JSON: value is not a string");
    return str;
  }

  int value_int(const std::string& key, int dflt) const
  {
    const Value* v = find(key);
    return v ? v->as_int() : dflt;
  }
  double value_double(const std::string& key, double dflt) const
  {
    const Value* v = find(key);
    return v ? v->as_double() : dflt;
  }
  bool value_bool(const std::string& key, bool dflt) const
  {
    const Value* v = find(key);
    return v ? v->as_bool() : dflt;
  }
  std::string value_string(const std::string& key, const std::string& dflt) const
  {
    const Value* v = find(key);
    return (v && v->is_string()) ? v->str : dflt;
  }
};

class Parser
{
public:
  Parser(const char* s, size_t n)
  : p_(s)
  , end_(s + n)
  , begin_(s)
  {
  }

  Value parse()
  {
    Value v = value(0);
    ws();
    if (p_ != end_)
      fail("trailing characters after JSON value");
    return v;
  }

private:
  const char* p_;
  const char* end_;
  const char* begin_;

  [[noreturn]] void fail(const std::string& what) const
  {
    throw ParseError("JSON parse error at byte " + std::to_string((size_t)(p_ - begin_)) + ": " + what);
  }
  void ws()
  {
    while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r'))
      ++p_;
  }
  bool lit(const char* s)
  {
    const size_t n = std::strlen(s);
    if ((size_t)(end_ - p_) >= n && std::memcmp(p_, s, n) == 0)
    {
      p_ += n;
      return true;
    }
    return false;
  }

  Value value(int depth)
  {
    if (depth > 256)
      fail("nesting too deep");
    ws();
    if (p_ >= end_)
      fail("unexpected end of input");
    Value v;
    switch (*p_)
    {
      case '{':
      {
        ++p_;
        v.type = Value::Object;
        ws();
        if (p_ < end_ && *p_ == '}')
        {
          ++p_;
          return v;
        }
        for (;;)
        {
          ws();
          if (p_ >= end_ || *p_ != '"')
            fail("expected string key");
          std::string k = string();
          ws();
          if (p_ >= end_ || *p_ != ':')
            fail("expected ':'");
          ++p_;
          Value child = value(depth + 1);
          // last duplicate wins (nlohmann behaviour)
          bool replaced = false;
          for (auto& kv : v.obj)
            if (kv.first == k)
            {
              kv.second = std::move(child);
              replaced = true;
              break;
            }
          if (!replaced)
            v.obj.emplace_back(std::move(k), std::move(child));
          ws();
          if (p_ < end_ && *p_ == ',')
          {
            ++p_;
            continue;
          }
          if (p_ < end_ && *p_ == '}')
          {
            ++p_;
            return v;
          }
          fail("expected ',' or '}'");
        }
      }
      case '[':
      {
        ++p_;
        v.type = Value::Array;
        ws();
        if (p_ < end_ && *p_ == ']')
        {
          ++p_;
          return v;
        }
        for (;;)
        {
          v.arr.push_back(value(depth + 1));
          ws();
          if (p_ < end_ && *p_ == ',')
          {
            ++p_;
            continue;
          }
          if (p_ < end_ && *p_ == ']')
          {
            ++p_;
            return v;
          }
          fail("expected ',' or ']'");
        }
      }
      case '"':
        v.type = Value::String;
        v.str = string();
        return v;
      case 't':
        if (!lit("true"))
          fail("invalid literal");
        v.type = Value::Bool;
        v.b = true;
        return v;
      case 'f':
        if (!lit("false"))
          fail("invalid literal");
        v.type = Value::Bool;
        v.b = false;
        return v;
      case 'n':
        if (!lit("null"))
          fail("invalid literal");
        return v;
      case 'N': // tolerate NaN (Python json.dump emits it)
        if (!lit("NaN"))
          fail("invalid literal");
        v.type = Value::Number;
        v.num = std::nan("");
        return v;
      default: return number();
    }
  }

  Value number()
  {
    const char* s = p_;
    if (p_ < end_ && (*p_ == '-' || *p_ == '+'))
      ++p_;
    if (lit("Infinity"))
    {
      Value v;
      v.type = Value::Number;
      v.num = (*s == '-') ? -HUGE_VAL : HUGE_VAL;
      return v;
    }
    bool any = false;
    while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '-' || *p_ == '+'))
    {
      any = true;
      ++p_;
    }
    if (!any)
      fail("invalid value");
    std::string tmp(s, (size_t)(p_ - s));
    char* e = nullptr;
    const double d = std::strtod(tmp.c_str(), &e);
    if (e == tmp.c_str() || *e != '\0')
      fail("invalid number \"" + tmp + "\"");
    Value v;
    v.type = Value::Number;
    v.num = d;
    return v;
  }

  static void put_utf8(std::string& out, unsigned cp)
  {
    if (cp < 0x80)
      out.push_back((char)cp);
    else if (cp < 0x800)
    {
      out.push_back((char)(0xC0 | (cp >> 6)));
      out.push_back((char)(0x80 | (cp & 0x3F)));
    }
    else if (cp < 0x10000)
    {
      out.push_back((char)(0xE0 | (cp >> 12)));
      out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      out.push_back((char)(0x80 | (cp & 0x3F)));
    }
    else
    {
      out.push_back((char)(0xF0 | (cp >> 18)));
      out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
      out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      out.push_back((char)(0x80 | (cp & 0x3F)));
    }
  }

  unsigned hex4()
  {
    if (end_ - p_ < 4)
      fail("truncated \\u escape");
    unsigned v = 0;
    for (int i = 0; i < 4; i++)
    {
      const char c = *p_++;
      v <<= 4;
      if (c >= '0' && c <= '9')
        v |= (unsigned)(c - '0');
      else if (c >= 'a' && c <= 'f')
        v |= (unsigned)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F')
        v |= (unsigned)(c - 'A' + 10);
      else
        fail("bad hex digit in \\u escape");
    }
    return v;
  }

  std::string string()
  {
    ++p_; // opening quote
    std::string out;
    while (p_ < end_)
    {
      const char c = *p_++;
      if (c == '"')
        return out;
      if (c != '\\')
      {
        out.push_back(c);
        continue;
      }
      if (p_ >= end_)
        break;
      const char e = *p_++;
      switch (e)
      {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u':
        {
          unsigned cp = hex4();
          if (cp >= 0xD800 && cp <= 0xDBFF && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u')
          {
            p_ += 2;
            const unsigned lo = hex4();
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          put_utf8(out, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
    fail("unterminated string");
  }
};

inline Value parse(const std::string& text)
{
  Parser p(text.data(), text.size());
  return p.parse();
}

// Serialise a value back to JSON text (compact). Numbers print with 17 significant digits: a double read from a
// .nam file and a float weight both survive the round trip exactly.
inline void dump_to(const Value& v, std::string& out)
{
  switch (v.type)
  {
    case Value::Null: out += "null"; break;
    case Value::Bool: out += v.b ? "true" : "false"; break;
    case Value::Number:
    {
      // what the parser above reads back: NaN / Infinity / -Infinity as Python's json module writes them (non-finite
      // numbers have no JSON form), the sign of zero kept, and the integer test only inside the range where the cast is defined
      if (v.num != v.num)
        out += "NaN";
      else if (v.num == HUGE_VAL || v.num == -HUGE_VAL)
        out += v.num > 0 ? "Infinity" : "-Infinity";
      else if (v.num == 0.0 && std::signbit(v.num))
        out += "-0.0";
      else if (v.num > -1e15 && v.num < 1e15 && v.num == (double)(long long)v.num)
        out += std::to_string((long long)v.num);
      else
      {
        char buf[40];
        std::snprintf(buf, sizeof(buf), "%.17g", v.num);
        out += buf;
      }
      break;
    }
    case Value::String:
    {
      out += '"';
      for (unsigned char c : v.str)
      {
        switch (c)
        {
          case '"': out += "\\\""; break;
          case '\\': out += "\\\\"; break;
          case '\n': out += "\\n"; break;
          case '\r': out += "\\r"; break;
          case '\t': out += "\\t"; break;
          default:
            if (c < 0x20)
            {
              char buf[8];
              std::snprintf(buf, sizeof(buf), "\\u%04x", (unsigned)c);
              out += buf;
            }
            else
              out += (char)c;
        }
      }
      out += '"';
      break;
    }
    case Value::Array:
    {
      out += '[';
      for (size_t i = 0; i < v.arr.size(); i++)
      {
        if (i)
          out += ',';
        dump_to(v.arr[i], out);
      }
      out += ']';
      break;
    }
    case Value::Object:
    {
      out += '{';
      for (size_t i = 0; i < v.obj.size(); i++)
      {
        if (i)
          out += ',';
        Value k;
        k.type = Value::String;
        k.str = v.obj[i].first;
        dump_to(k, out);
        out += ':';
        dump_to(v.obj[i].second, out);
      }
      out += '}';
      break;
    }
  }
}
inline std::string dump(const Value& v)
{
  std::string out;
  dump_to(v, out);
  return out;
}


} // namespace json
} // namespace namhip
