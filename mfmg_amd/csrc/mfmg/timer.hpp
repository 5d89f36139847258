// Stand-in for the optional dealii::TimerOutput threaded through Hierarchy
// (include/mfmg/common/hierarchy.hpp:36-47,161-164,369): same section names,
// wall time accumulated per section, summary table on request.
#pragma once

#include "../common.hpp"

#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace mfmg
{
class TimerOutput
{
public:
  void enter_subsection(std::string const &name)
  {
    _stack.push_back({name, std::chrono::steady_clock::now()});
    if (_order.find(name) == _order.end())
      _order[name] = _order.size();
  }
  void leave_subsection()
  {
    if (_stack.empty())
      return;
    auto e = _stack.back();
    _stack.pop_back();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - e.start).count();
    auto &s = _sections[e.name];
    s.first += dt;
    s.second += 1;
  }
  std::string summary() const
  {
    std::vector<std::string> names(_order.size());
    for (auto const &kv : _order)
      names[kv.second] = kv.first;
    std::ostringstream os;
    os << "+---------------------------------------------+------------+------------+\n";
    os << "| Section                                     | no. calls  |  wall time |\n";
    os << "+---------------------------------------------+------------+------------+\n";
    for (auto const &n : names)
    {
      auto it = _sections.find(n);
      if (it == _sections.end())
        continue;
      char buf[160];
      snprintf(buf, sizeof(buf), "| %-43s | %10ld | %9.4fs |\n", n.c_str(), it->second.second, it->second.first);
      os << buf;
    }
    os << "+---------------------------------------------+------------+------------+\n";
    return os.str();
  }
  double wall_time(std::string const &name) const
  {
    auto it = _sections.find(name);
    return it == _sections.end() ? 0. : it->second.first;
  }

private:
  struct Entry
  {
    std::string name;
    std::chrono::steady_clock::time_point start;
  };
  std::vector<Entry> _stack;
  std::map<std::string, std::pair<double, long>> _sections;
  std::map<std::string, size_t> _order;
};

// roctx ranges around the same sections (the reference's TimerOutput sections, include/mfmg/common/hierarchy.hpp:164-271),
// so that a `rocprofv3 --marker-trace` timeline shows "Setup: build restrictor", "Apply: fine levels", ... next to the
// kernels.  The marker library is resolved at run time: the copy a profiler has already loaded, or -- with
// MFMG_HIP_ROCTX=1 -- librocprofiler-sdk-roctx / libroctx64 from the ROCm installation; without either the calls are
// two null-pointer tests.  MFMG_HIP_ROCTX=0 switches the ranges off.
struct RoctxApi
{
  int (*push)(char const *) = nullptr;
  int (*pop)() = nullptr;
};
inline RoctxApi const &roctx_api()
{
  static const RoctxApi api = [] {
    RoctxApi a;
    char const *env = std::getenv("MFMG_HIP_ROCTX");
    if (env && env[0] == '0')
      return a;
    char const *names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    void *lib = nullptr;
    for (char const *n : names)
      if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr)
        break;
    if (!lib && env && env[0] == '1')
      for (char const *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr)
          break;
    if (lib)
    {
      a.push = reinterpret_cast<int (*)(char const *)>(dlsym(lib, "roctxRangePushA"));
      a.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
      if (!a.push || !a.pop)
        a.push = nullptr, a.pop = nullptr;
    }
    return a;
  }();
  return api;
}

// (the open section is also the PHASE device allocations are booked under: common.hpp, DeviceMemoryLedger)
inline std::vector<std::string> &memory_phase_stack()
{
  static thread_local std::vector<std::string> s;
  return s;
}
inline void timer_enter_subsection(std::shared_ptr<TimerOutput> timer, std::string const &section)
{
  if (roctx_api().push)
    roctx_api().push(section.c_str());
  if (timer)
    timer->enter_subsection(section);
  memory_phase_stack().push_back(DeviceMemoryLedger::phase());
  if (section != "Setup" && section.rfind("Apply", 0) != 0) // (the outer section and the apply sections keep the phase they run in)
    DeviceMemoryLedger::phase() = section;
}
inline void timer_leave_subsection(std::shared_ptr<TimerOutput> timer)
{
  if (!memory_phase_stack().empty())
  {
    DeviceMemoryLedger::phase() = memory_phase_stack().back();
    memory_phase_stack().pop_back();
  }
  if (timer)
    timer->leave_subsection();
  if (roctx_api().pop)
    roctx_api().pop();
}
} // namespace mfmg
