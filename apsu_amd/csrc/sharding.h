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
// compute_powers_cost > 0 (in the same unit, degree + 64 per BinBundle; about 110 per ciphertext product of the PowersDag on
// MI355X): a spill pass then moves BinBundles off the slowest device to devices of other bundle indices while that lowers the
// slowest device's cost INCLUDING the extra ComputePowers the receiving device has to run (break-even: DESIGN.md section 6).
std::vector<int> partition_units(const std::vector<ShardUnit> &units, uint32_t bundle_idx_count, int world, uint64_t compute_powers_cost = 0);

} // namespace apsu_he
