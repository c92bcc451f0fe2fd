// epv_options.hpp -- a ~100-line stand-in for smithlab_cpp's OptionParser (absent,
// un-vendored submodule).  Same surface as the calls the
// reference mains make: add_opt(long, short, help, required, variable), bool options
// are switches, "-x value" / "-long value" / "--long value", help on -? / -help.
#ifndef EPV_OPTIONS_HPP
#define EPV_OPTIONS_HPP

#include <string>
#include <vector>

namespace epv {

class OptionParser {
public:
  OptionParser(const std::string &prog, const std::string &descr, const std::string &args)
      : prog_(prog), descr_(descr), args_(args) {}
  void add_opt(const std::string &l, char s, const std::string &h, bool req, bool &v);
  void add_opt(const std::string &l, char s, const std::string &h, bool req, size_t &v);
  void add_opt(const std::string &l, char s, const std::string &h, bool req, double &v);
  void add_opt(const std::string &l, char s, const std::string &h, bool req, std::string &v);
  void parse(int argc, const char **argv, std::vector<std::string> &leftover);
  bool help_requested() const { return help_; }
  bool option_missing() const { return !missing_.empty(); }
  std::string option_missing_message() const { return "required argument missing: [" + missing_ + "]"; }
  std::string help_message() const;
  std::string about_message() const { return descr_; }

private:
  enum Kind { BOOL, SIZE, DOUBLE, STRING };
  struct Opt {
    std::string lname, help;
    char sname;
    bool required, seen;
    Kind kind;
    void *target;
  };
  void add(const std::string &l, char s, const std::string &h, bool req, Kind k, void *t);
  std::vector<Opt> opts_;
  std::string prog_, descr_, args_, missing_;
  bool help_ = false;
};

}  // namespace epv

#endif
