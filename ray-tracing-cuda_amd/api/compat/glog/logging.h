// LOG(INFO) / CHECK(cond) work-alike for the scene sources (glog is an un-fetched submodule
// of the reference).  LOG streams one line to stderr; CHECK aborts with the streamed message.
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>

namespace rt_log {
struct Line {
  std::ostringstream os;
  bool fatal;
  Line(const char *sev, const char *file, int line, bool f) : fatal(f) { os << sev << " " << file << ":" << line << "] "; }
  ~Line() {
    std::cerr << os.str() << std::endl;
    if (fatal) std::abort();
  }
  template <class T>
  Line &operator<<(const T &v) {
    os << v;
    return *this;
  }
};
struct Voidify {
  void operator&(const Line &) {}
};
}  // namespace rt_log

#define LOG(sev) ::rt_log::Line(#sev, __FILE__, __LINE__, false)
#define CHECK(cond) \
  (cond) ? (void)0 : ::rt_log::Voidify() & ::rt_log::Line("CHECK failed: " #cond, __FILE__, __LINE__, true)
