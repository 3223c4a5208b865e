#include "sharding.h"

#include <algorithm>
#include <stdexcept>

namespace apsu_he {

std::vector<int> partition_units(const std::vector<ShardUnit> &units, uint32_t bundle_idx_count, int world, uint64_t compute_powers_cost)
{
    if (world <= 0) throw std::invalid_argument("no devices");
    if (!bundle_idx_count) throw std::invalid_argument("bundle_idx_count is zero");
    constexpr uint64_t UNIT_OVERHEAD = 64;                       // relinearisation, epilogue
    auto cost = [&](size_t i) { return (uint64_t)units[i].degree + UNIT_OVERHEAD; };
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
            load[best] += cost(i);
        }
    }
    if (!compute_powers_cost) return out;
    // Spill pass (bundle indices do not always divide over the devices: 3 indices on 8 devices leave one index with two devices
    // and 1.5x the BinBundles per device).  A device pays ComputePowers once per bundle index it holds, so moving a BinBundle
    // to a device of ANOTHER index only pays when what it takes off the slowest device outweighs a second ComputePowers there:
    // the busiest device hands its cheapest unit to whichever device ends up least loaded, while that lowers the maximum.
    std::vector<std::vector<uint32_t>> held(world, std::vector<uint32_t>(bundle_idx_count, 0));   // units of index b on device r
    for (size_t i = 0; i < units.size(); i++) held[out[i]][units[i].bundle_idx]++;
    auto total = [&](int r) {
        uint64_t t = load[r];
        for (uint32_t b = 0; b < bundle_idx_count; b++) if (held[r][b]) t += compute_powers_cost;
        return t;
    };
    for (size_t guard = 0; guard < units.size() * 4 + 16; guard++) {
        int rmax = 0;
        for (int r = 1; r < world; r++) if (total(r) > total(rmax)) rmax = r;
        const uint64_t tmax = total(rmax);
        // the unit of rmax whose move gives the lowest resulting maximum of (rmax, target)
        uint64_t best_peak = tmax;
        size_t best_unit = units.size();
        int best_dst = -1;
        for (size_t i = 0; i < units.size(); i++) {
            if (out[i] != rmax) continue;
            const uint32_t b = units[i].bundle_idx;
            const uint64_t src_after = tmax - cost(i) - (held[rmax][b] == 1 ? compute_powers_cost : 0);
            for (int r = 0; r < world; r++) {
                if (r == rmax) continue;
                const uint64_t dst_after = total(r) + cost(i) + (held[r][b] ? 0 : compute_powers_cost);
                const uint64_t peak = std::max(src_after, dst_after);
                if (peak < best_peak) { best_peak = peak; best_unit = i; best_dst = r; }   // ties keep the earliest (unit, device)
            }
        }
        if (best_dst < 0) break;
        const uint32_t b = units[best_unit].bundle_idx;
        held[rmax][b]--; held[best_dst][b]++;
        load[rmax] -= cost(best_unit); load[best_dst] += cost(best_unit);
        out[best_unit] = best_dst;
    }
    return out;
}

} // namespace apsu_he
