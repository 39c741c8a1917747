// tclap/CmdLine.h -- the subset of the TCLAP interface the reference's hot-path tools use
// (ValueArg, MultiArg, CmdLine::parse, ArgException; e.g. tools/ExtractFeatures.cxx:19-71),
// written for this build because TCLAP is not available.  Same flag syntax: -x value,
// --name value, --name=value; -h/--help and --version are handled like TCLAP does (print
// and exit 0).
#ifndef IFE_HOST_TCLAP_CMDLINE_H
#define IFE_HOST_TCLAP_CMDLINE_H

#include <cstdlib>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace TCLAP {

class ArgException : public std::exception {
 public:
  ArgException(const std::string &text, const std::string &id) : text_(text), id_(id) {}
  std::string error() const { return text_; }
  std::string argId() const { return id_; }
  const char *what() const noexcept override { return text_.c_str(); }

 private:
  std::string text_, id_;
};

class CmdLine;

class Arg {
 public:
  Arg(const std::string &flag, const std::string &name, const std::string &desc, bool req,
      const std::string &typeDesc)
      : flag_(flag), name_(name), desc_(desc), type_(typeDesc), req_(req) {}
  virtual ~Arg() {}
  virtual void take(const std::string &value) = 0;
  virtual bool multi() const { return false; }
  std::string id() const { return "-" + flag_ + " (--" + name_ + ")"; }
  std::string flag_, name_, desc_, type_;
  bool req_, set_ = false;
};

class CmdLine {
 public:
  CmdLine(const std::string &message, char = ' ', const std::string &version = "none")
      : message_(message), version_(version) {}
  void add(Arg *a) { args_.push_back(a); }
  void parse(int argc, const char *const *argv) {
    const std::string prog = argc > 0 ? argv[0] : "tool";
    for (int i = 1; i < argc; ++i) {
      std::string tok = argv[i], value;
      bool has_value = false;
      if (tok == "-h" || tok == "--help") { usage(prog); std::exit(0); }
      if (tok == "--version") { std::cout << prog << "  version: " << version_ << std::endl; std::exit(0); }
      Arg *hit = nullptr;
      if (tok.size() > 2 && tok[0] == '-' && tok[1] == '-') {
        std::string nm = tok.substr(2);
        const size_t eq = nm.find('=');
        if (eq != std::string::npos) { value = nm.substr(eq + 1); nm = nm.substr(0, eq); has_value = true; }
        for (Arg *a : args_) if (a->name_ == nm) hit = a;
      } else if (tok.size() == 2 && tok[0] == '-') {
        for (Arg *a : args_) if (a->flag_ == tok.substr(1)) hit = a;
      }
      if (!hit) throw ArgException("Couldn't find match for argument", tok);
      if (!has_value) {
        if (i + 1 >= argc) throw ArgException("Missing a value for this argument!", hit->id());
        value = argv[++i];
      }
      if (hit->set_ && !hit->multi()) throw ArgException("Argument already set!", hit->id());
      hit->take(value);
      hit->set_ = true;
    }
    for (Arg *a : args_)
      if (a->req_ && !a->set_) throw ArgException("Required argument missing: " + a->name_, a->id());
  }
  void usage(const std::string &prog) const {
    std::cout << "USAGE:\n   " << prog;
    for (Arg *a : args_) std::cout << (a->req_ ? " " : " [") << "-" << a->flag_ << " <" << a->type_ << ">" << (a->multi() ? " ..." : "") << (a->req_ ? "" : "]");
    std::cout << "\n\nWhere:\n";
    for (Arg *a : args_) std::cout << "   -" << a->flag_ << ",  --" << a->name_ << " <" << a->type_ << ">" << (a->req_ ? "  (required)" : "") << "\n     " << a->desc_ << "\n";
    std::cout << "\n   " << message_ << std::endl;
  }

 private:
  std::string message_, version_;
  std::vector<Arg *> args_;
};

namespace detail {
template <typename T>
inline T convert(const std::string &s, const Arg &a) {
  std::istringstream is(s);
  T v;
  is >> std::boolalpha >> v;
  if (is.fail()) {
    std::istringstream is2(s);  // "1"/"0" for bool
    is2 >> v;
    if (is2.fail()) throw ArgException("Couldn't read argument value from string '" + s + "'", a.id());
    return v;
  }
  return v;
}
template <>
inline std::string convert<std::string>(const std::string &s, const Arg &) { return s; }
}  // namespace detail

template <typename T>
class ValueArg : public Arg {
 public:
  ValueArg(const std::string &flag, const std::string &name, const std::string &desc, bool req, T def,
           const std::string &typeDesc, CmdLine &cmd)
      : Arg(flag, name, desc, req, typeDesc), value_(def) { cmd.add(this); }
  void take(const std::string &s) override { value_ = detail::convert<T>(s, *this); }
  const T &getValue() const { return value_; }
  bool isSet() const { return set_; }

 private:
  T value_;
};

template <typename T>
class MultiArg : public Arg {
 public:
  MultiArg(const std::string &flag, const std::string &name, const std::string &desc, bool req,
           const std::string &typeDesc, CmdLine &cmd)
      : Arg(flag, name, desc, req, typeDesc) { cmd.add(this); }
  void take(const std::string &s) override { values_.push_back(detail::convert<T>(s, *this)); }
  bool multi() const override { return true; }
  const std::vector<T> &getValue() const { return values_; }

 private:
  std::vector<T> values_;
};

}  // namespace TCLAP

#endif
