// See powers_dag.h (reference: common/apsu/powers.cpp:22-107).
#include "powers_dag.h"

#include <algorithm>

namespace apsu_he {

void PowersDag::reset()
{
    by_power_.clear();
    targets_.clear();
    ready_ = false;
    max_depth_ = 0;
    n_sources_ = 0;
}

bool PowersDag::configure(std::set<uint32_t> sources, std::set<uint32_t> targets)
{
    reset();
    if (sources.count(0) || !sources.count(1)) return false;
    if (targets.count(0) || !targets.count(1)) return false;
    if (!std::includes(targets.begin(), targets.end(), sources.begin(), sources.end())) return false;

    for (uint32_t s : sources) by_power_[s] = PowersNode{ s, 0, { 0, 0 } };
    uint32_t deepest = 0;
    for (uint32_t power : targets) {
        if (sources.count(power)) continue;
        // best split power = s1 + s2 over target powers; ties keep the earliest s1
        PowersNode best{ power, power - 1, { power - 1, 1 } };
        for (uint32_t s1 : targets) {
            if (s1 >= power) break;
            uint32_t s2 = power - s1;
            if (!targets.count(s2)) continue;
            uint32_t d = std::max(by_power_.at(s1).depth, by_power_.at(s2).depth) + 1;
            if (d < best.depth) best = PowersNode{ power, d, { s1, s2 } };
        }
        by_power_[power] = best;
        deepest = std::max(deepest, best.depth);
    }
    ready_ = true;
    targets_ = std::move(targets);
    max_depth_ = deepest;
    n_sources_ = (uint32_t)sources.size();
    return true;
}

std::vector<std::vector<PowersDag::PowersNode>> PowersDag::levels() const
{
    std::vector<std::vector<PowersNode>> out(max_depth_ + 1);
    for (auto &kv : by_power_) out[kv.second.depth].push_back(kv.second);
    return out;
}

} // namespace apsu_he
