// BinBundle sharding across the GPUs of one node; host logic only (no HIP), shared by the in-process
// multi-GPU engine (multi.cpp) and mirrored by apsu_amd/sharding.py for the one-process-per-GPU bench.
#pragma once
#include <cstdint>
#include <vector>

namespace apsu_he {

struct ShardUnit { uint32_t bundle_idx, cache_idx, degree; };

// -> device slot of every unit.  The independent unit is one BinBundle (receiver/apsu/receiver_osn.cpp:334-359 treats
// them as independent tasks).  world >= bundle_idx_count: device r serves index r % count, else index b lives on device
// b % world (a device then needs the powers of few indices only); inside an index units go, largest degree first (ties:
// smaller cache_idx), to the least loaded of its devices (cost = degree + 64).
std::vector<int> partition_units(const std::vector<ShardUnit> &units, uint32_t bundle_idx_count, int world);

} // namespace apsu_he
