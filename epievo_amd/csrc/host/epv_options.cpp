#include "epv_options.hpp"

#include <cstdlib>
#include <sstream>
#include <stdexcept>

namespace epv {

void OptionParser::add(const std::string &l, char s, const std::string &h, bool req, Kind k, void *t) {
  opts_.push_back(Opt{l, h, s, req, false, k, t});
}
void OptionParser::add_opt(const std::string &l, char s, const std::string &h, bool r, bool &v) { add(l, s, h, r, BOOL, &v); }
void OptionParser::add_opt(const std::string &l, char s, const std::string &h, bool r, size_t &v) { add(l, s, h, r, SIZE, &v); }
void OptionParser::add_opt(const std::string &l, char s, const std::string &h, bool r, double &v) { add(l, s, h, r, DOUBLE, &v); }
void OptionParser::add_opt(const std::string &l, char s, const std::string &h, bool r, std::string &v) { add(l, s, h, r, STRING, &v); }

void OptionParser::parse(int argc, const char **argv, std::vector<std::string> &leftover) {
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-?" || a == "-help" || a == "--help" || a == "-h") { help_ = true; continue; }
    Opt *hit = nullptr;
    if (a.size() >= 2 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9') && a[1] != '.') {
      const std::string name = a.substr(a[1] == '-' ? 2 : 1);
      for (Opt &o : opts_)
        if ((name.size() == 1 && name[0] == o.sname) || name == o.lname) hit = &o;
      if (!hit) throw std::runtime_error("unknown option: " + a);
    }
    if (!hit) { leftover.push_back(a); continue; }
    hit->seen = true;
    if (hit->kind == BOOL) { *static_cast<bool *>(hit->target) = true; continue; }
    if (i + 1 >= argc) throw std::runtime_error("option " + a + " needs a value");
    const std::string v = argv[++i];
    if (hit->kind == SIZE) *static_cast<size_t *>(hit->target) = std::strtoull(v.c_str(), nullptr, 10);
    else if (hit->kind == DOUBLE) *static_cast<double *>(hit->target) = std::atof(v.c_str());
    else *static_cast<std::string *>(hit->target) = v;
  }
  for (const Opt &o : opts_)
    if (o.required && !o.seen) missing_ += (missing_.empty() ? "" : ", ") + std::string("-") + o.sname;
}

std::string OptionParser::help_message() const {
  std::ostringstream oss;
  oss << "Usage: " << prog_ << " [OPTIONS] " << args_ << "\n\nOptions:\n";
  for (const Opt &o : opts_)
    oss << "  -" << o.sname << ", -" << o.lname << "  " << o.help << (o.required ? " [REQUIRED]" : "") << "\n";
  return oss.str();
}

}  // namespace epv
