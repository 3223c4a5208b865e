// PowersDag: depth-optimal DAG from the query's source powers to every target power.
// Mirrors apsu::PowersDag (common/apsu/powers.h:53-77,88,158 ; common/apsu/powers.cpp:22-107):
// same parent choice (first minimal-depth split in ascending s1 order), same node fields.
// parallel_apply's spin scheduler (powers.h:158-278) is replaced by level-synchronous batches:
// all nodes of equal depth are independent and run as one set of GPU launches.
#pragma once
#include <cstdint>
#include <map>
#include <set>
#include <utility>
#include <vector>

namespace apsu_he {

class PowersDag {
public:
    struct PowersNode {
        uint32_t power = 0;
        uint32_t depth = 0;
        std::pair<uint32_t, uint32_t> parents{ 0, 0 };
        bool is_source() const { return parents.first == 0 && parents.second == 0; }
    };

    bool configure(std::set<uint32_t> source_powers, std::set<uint32_t> target_powers);
    bool is_configured() const { return ready_; }
    uint32_t depth() const { return max_depth_; }
    uint32_t source_count() const { return n_sources_; }
    const std::set<uint32_t> &target_powers() const { return targets_; }
    const std::map<uint32_t, PowersNode> &nodes() const { return by_power_; }
    // nodes grouped by depth (index 0 = sources), ascending power inside a level
    std::vector<std::vector<PowersNode>> levels() const;
    void reset();

private:
    std::map<uint32_t, PowersNode> by_power_;
    std::set<uint32_t> targets_;
    bool ready_ = false;
    uint32_t max_depth_ = 0, n_sources_ = 0;
};

} // namespace apsu_he
