// Minimal boost::property_tree::ptree stand-in with an INFO-format reader.
// The reference threads a boost ptree through every constructor
// (include/mfmg/common/hierarchy.hpp:159-172) and reads it from INFO files
// (tests/data/hierarchy_input.info, tests/test_hierarchy.cc:207-209); Boost is
// not available here, so the subset actually used is restated: get / get with
// default / get_optional / put / get_child / get_child_optional, '.'-separated paths.
#pragma once

#include <algorithm>
#include <memory>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace mfmg
{
class ptree
{
public:
  ptree() = default;

  // ---- typed access -----------------------------------------------------
  template <typename T>
  T get(std::string const &path) const
  {
    ptree const *node = find(path);
    if (node == nullptr)
      throw std::runtime_error("No such node (" + path + ")");
    return convert<T>(node->_data, path);
  }
  template <typename T>
  T get(std::string const &path, T const &default_value) const
  {
    ptree const *node = find(path);
    if (node == nullptr)
      return default_value;
    return convert<T>(node->_data, path);
  }
  std::string get(std::string const &path, char const *default_value) const
  {
    return get<std::string>(path, std::string(default_value));
  }
  template <typename T>
  std::optional<T> get_optional(std::string const &path) const
  {
    ptree const *node = find(path);
    if (node == nullptr)
      return std::nullopt;
    return convert<T>(node->_data, path);
  }
  template <typename T>
  void put(std::string const &path, T const &value)
  {
    ptree *node = find_or_create(path);
    node->_data = to_string(value);
  }
  ptree const &get_child(std::string const &path) const
  {
    ptree const *node = find(path);
    if (node == nullptr)
      throw std::runtime_error("No such node (" + path + ")");
    return *node;
  }
  ptree const *get_child_optional(std::string const &path) const { return find(path); }
  ptree &put_child(std::string const &path, ptree const &child)
  {
    ptree *node = find_or_create(path);
    *node = child;
    return *node;
  }
  bool empty() const { return _children.empty(); }
  std::string const &data() const { return _data; }
  std::vector<std::pair<std::string, ptree>> const &children() const { return _children; }

  // ---- INFO reader (boost::property_tree::info_parser::read_info) ----------
  static ptree parse_info(std::string const &text)
  {
    std::vector<std::string> tok = tokenize(text);
    size_t pos = 0;
    ptree root;
    parse_block(tok, pos, root, false);
    return root;
  }

  std::string to_info(int indent = 0) const
  {
    std::ostringstream os;
    for (auto const &kv : _children)
    {
      os << std::string(indent, ' ') << quote(kv.first);
      if (!kv.second._data.empty())
        os << ' ' << quote(kv.second._data);
      os << '\n';
      if (!kv.second._children.empty())
      {
        os << std::string(indent, ' ') << "{\n" << kv.second.to_info(indent + 2) << std::string(indent, ' ')
           << "}\n";
      }
    }
    return os.str();
  }

private:
  std::string _data;
  std::vector<std::pair<std::string, ptree>> _children;

  static std::string quote(std::string const &s)
  {
    if (s.find_first_of(" \t;{}") != std::string::npos || s.empty())
      return "\"" + s + "\"";
    return s;
  }
  template <typename T>
  static std::string to_string(T const &v)
  {
    if constexpr (std::is_same<T, bool>::value)
      return v ? "true" : "false";
    else if constexpr (std::is_convertible<T, std::string>::value)
      return std::string(v);
    else
    {
      std::ostringstream os;
      os.precision(17);
      os << v;
      return os.str();
    }
  }
  template <typename T>
  static T convert(std::string const &s, std::string const &path)
  {
    if constexpr (std::is_same<T, std::string>::value)
      return s;
    else if constexpr (std::is_same<T, bool>::value)
    {
      std::string l = s;
      std::transform(l.begin(), l.end(), l.begin(), ::tolower);
      if (l == "true" || l == "1")
        return true;
      if (l == "false" || l == "0")
        return false;
      throw std::runtime_error("conversion of data to type bool failed (" + path + ")");
    }
    else
    {
      std::istringstream is(s);
      T v;
      is >> v;
      if (is.fail())
        throw std::runtime_error("conversion of data failed (" + path + ")");
      return v;
    }
  }
  static std::vector<std::string> split(std::string const &path)
  {
    std::vector<std::string> out;
    std::string cur;
    for (char ch : path)
    {
      if (ch == '.')
      {
        out.push_back(cur);
        cur.clear();
      }
      else
        cur.push_back(ch);
    }
    out.push_back(cur);
    return out;
  }
  ptree const *find(std::string const &path) const
  {
    ptree const *node = this;
    for (auto const &key : split(path))
    {
      ptree const *next = nullptr;
      for (auto const &kv : node->_children)
        if (kv.first == key)
        {
          next = &kv.second;
          break;
        }
      if (next == nullptr)
        return nullptr;
      node = next;
    }
    return node;
  }
  ptree *find_or_create(std::string const &path)
  {
    ptree *node = this;
    for (auto const &key : split(path))
    {
      ptree *next = nullptr;
      for (auto &kv : node->_children)
        if (kv.first == key)
        {
          next = &kv.second;
          break;
        }
      if (next == nullptr)
      {
        node->_children.emplace_back(key, ptree());
        next = &node->_children.back().second;
      }
      node = next;
    }
    return node;
  }
  // tokens: "{", "}", "\n" and words (quotes removed, '\x01' prefix marks a quoted word)
  static std::vector<std::string> tokenize(std::string const &text)
  {
    std::vector<std::string> tok;
    size_t i = 0;
    while (i < text.size())
    {
      char ch = text[i];
      if (ch == ';')
      {
        while (i < text.size() && text[i] != '\n')
          ++i;
      }
      else if (ch == '\n')
      {
        tok.push_back("\n");
        ++i;
      }
      else if (ch == ' ' || ch == '\t' || ch == '\r')
        ++i;
      else if (ch == '{' || ch == '}')
      {
        tok.push_back(std::string(1, ch));
        ++i;
      }
      else if (ch == '"')
      {
        std::string w;
        ++i;
        while (i < text.size() && text[i] != '"')
        {
          if (text[i] == '\\' && i + 1 < text.size())
            ++i;
          w.push_back(text[i++]);
        }
        ++i;
        tok.push_back(std::string("\x01") + w);
      }
      else
      {
        std::string w;
        while (i < text.size() && std::string(" \t\r\n;{}").find(text[i]) == std::string::npos)
          w.push_back(text[i++]);
        tok.push_back(w);
      }
    }
    return tok;
  }
  static bool is_word(std::string const &t) { return t != "\n" && t != "{" && t != "}"; }
  static std::string word(std::string const &t) { return (!t.empty() && t[0] == '\x01') ? t.substr(1) : t; }
  static void parse_block(std::vector<std::string> const &tok, size_t &pos, ptree &node, bool nested)
  {
    ptree *last = nullptr;
    while (pos < tok.size())
    {
      std::string const &t = tok[pos];
      if (t == "\n")
      {
        ++pos;
      }
      else if (t == "}")
      {
        if (!nested)
          throw std::runtime_error("INFO parse error: unmatched '}'");
        ++pos;
        return;
      }
      else if (t == "{")
      {
        if (last == nullptr)
          throw std::runtime_error("INFO parse error: '{' without a key");
        ++pos;
        parse_block(tok, pos, *last, true);
        last = nullptr;
      }
      else
      {
        std::string key = word(t);
        ++pos;
        std::string value;
        if (pos < tok.size() && is_word(tok[pos]))
        {
          value = word(tok[pos]);
          ++pos;
        }
        node._children.emplace_back(key, ptree());
        last = &node._children.back().second;
        last->_data = value;
      }
    }
    if (nested)
      throw std::runtime_error("INFO parse error: missing '}'");
  }
};
} // namespace mfmg
