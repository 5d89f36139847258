// Stand-in for the optional dealii::TimerOutput threaded through Hierarchy
// (include/mfmg/common/hierarchy.hpp:36-47,161-164,369): same section names,
// wall time accumulated per section, summary table on request.
#pragma once

#include <chrono>
#include <map>
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

inline void timer_enter_subsection(std::shared_ptr<TimerOutput> timer, std::string const &section)
{
  if (timer)
    timer->enter_subsection(section);
}
inline void timer_leave_subsection(std::shared_ptr<TimerOutput> timer)
{
  if (timer)
    timer->leave_subsection();
}
} // namespace mfmg
