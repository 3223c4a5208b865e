#include "sharding.h"

#include <algorithm>
#include <stdexcept>

namespace apsu_he {

std::vector<int> partition_units(const std::vector<ShardUnit> &units, uint32_t bundle_idx_count, int world)
{
    if (world <= 0) throw std::invalid_argument("no devices");
    if (!bundle_idx_count) throw std::invalid_argument("bundle_idx_count is zero");
    constexpr uint64_t UNIT_OVERHEAD = 64;                       // relinearisation, epilogue
    std::vector<std::vector<int>> devs_of(bundle_idx_count);
    if ((uint32_t)world >= bundle_idx_count)
        for (int r = 0; r < world; r++) devs_of[(uint32_t)r % bundle_idx_count].push_back(r);
    else
        for (uint32_t b = 0; b < bundle_idx_count; b++) devs_of[b].push_back((int)(b % (uint32_t)world));
    std::vector<uint64_t> load(world, 0);
    std::vector<int> out(units.size(), -1);
    for (uint32_t b = 0; b < bundle_idx_count; b++) {
        std::vector<size_t> mine;
        for (size_t i = 0; i < units.size(); i++) {
            if (units[i].bundle_idx >= bundle_idx_count) throw std::invalid_argument("bundle_idx out of range");
            if (units[i].bundle_idx == b) mine.push_back(i);
        }
        std::stable_sort(mine.begin(), mine.end(), [&](size_t x, size_t y) {
            if (units[x].degree != units[y].degree) return units[x].degree > units[y].degree;
            return units[x].cache_idx < units[y].cache_idx;
        });
        for (size_t i : mine) {
            int best = devs_of[b][0];
            for (int r : devs_of[b]) if (load[r] < load[best] || (load[r] == load[best] && r < best)) best = r;
            out[i] = best;
            load[best] += units[i].degree + UNIT_OVERHEAD;
        }
    }
    return out;
}

} // namespace apsu_he
