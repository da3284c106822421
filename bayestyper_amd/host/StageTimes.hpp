// Wall-clock per stage of a `bayesTyper cluster` / `genotype` run, printed to stderr at the end when BT_STAGE_TIMES is set (tools/e2e_c2.sh keeps the table
// under profiles/): which host stage a run waits for.  StageScope("name") adds the lifetime of the object to the stage's total; stages may nest
// (a nested stage's time is part of its parent's too).
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <sys/resource.h>
#include <string>
#include <utility>
#include <vector>

namespace bthost {

class StageTimes {
  public:
    static StageTimes &get() {
        static StageTimes t;
        return t;
    }
    void add(const std::string &name, double seconds) {
        std::lock_guard<std::mutex> lock(mu);
        for (auto &r : rows)
            if (r.first == name) {
                r.second += seconds;
                return;
            }
        rows.emplace_back(name, seconds);
    }
    void print(const char *title) {
        if (!getenv("BT_STAGE_TIMES")) return;
        std::lock_guard<std::mutex> lock(mu);
        std::fprintf(stderr, "\n## stage times: %s\n", title);
        for (auto &r : rows) std::fprintf(stderr, "%-52s %10.3f s\n", r.first.c_str(), r.second);
        struct rusage ru;
        if (getrusage(RUSAGE_SELF, &ru) == 0) std::fprintf(stderr, "%-52s %10.3f GB\n", "peak host memory (resident set)", ru.ru_maxrss / 1048576.0);
        rows.clear();
    }

  private:
    std::mutex mu;
    std::vector<std::pair<std::string, double>> rows;
};

class StageScope {
  public:
    explicit StageScope(std::string name) : name_(std::move(name)), t0(std::chrono::steady_clock::now()) {}
    ~StageScope() { StageTimes::get().add(name_, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }

  private:
    std::string name_;
    std::chrono::steady_clock::time_point t0;
};

}  // namespace bthost
