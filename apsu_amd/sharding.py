"""BinBundle sharding across the GPUs of one node.

The independent unit is one BinBundle (bundle_idx, cache_idx): the reference already treats them as
independent thread-pool tasks (receiver/apsu/receiver_osn.cpp:334-359).  GPUs are assigned to bundle
indices first (each GPU then only needs the powers of its indices), and an index's BinBundles are
split over its GPUs by cost ~ degree with a longest-processing-time greedy.  The only collective of
the path is the final gather of fixed-size result ciphertexts (2*n words each) to rank 0.
"""

UNIT_OVERHEAD = 64          # cost model: degree + constant (relinearisation, epilogue)


def partition(units, bundle_idx_count, world, compute_powers_cost=0):
    """units: [(bundle_idx, cache_idx, degree)] -> {rank: [unit, ...]} (deterministic on every rank).
    compute_powers_cost > 0 adds the spill pass of apsu_amd/csrc/sharding.cpp (same rule, same ties)."""
    ranks_of = {b: [] for b in range(bundle_idx_count)}
    if world >= bundle_idx_count:
        for r in range(world):
            ranks_of[r % bundle_idx_count].append(r)
    else:
        for b in range(bundle_idx_count):
            ranks_of[b].append(b % world)
    assign = {r: [] for r in range(world)}
    for b in range(bundle_idx_count):
        load = {r: sum(u[2] + UNIT_OVERHEAD for u in assign[r]) for r in ranks_of[b]}
        for u in sorted([u for u in units if u[0] == b], key=lambda u: (-u[2], u[1])):
            r = min(sorted(load), key=lambda k: load[k])
            assign[r].append(u)
            load[r] += u[2] + UNIT_OVERHEAD
    if not compute_powers_cost:
        return assign
    order = {u: i for i, u in enumerate(units)}                    # ties are broken by position in `units`, as in the C++ rule
    cost = lambda u: u[2] + UNIT_OVERHEAD

    def total(r):
        return sum(cost(u) for u in assign[r]) + compute_powers_cost * len({u[0] for u in assign[r]})

    for _ in range(len(units) * 4 + 16):
        rmax = max(range(world), key=lambda r: (total(r), -r))
        tmax = total(rmax)
        best = None                                                # (peak, unit position, destination)
        for u in sorted(assign[rmax], key=lambda u: order[u]):
            alone = sum(1 for v in assign[rmax] if v[0] == u[0]) == 1
            src_after = tmax - cost(u) - (compute_powers_cost if alone else 0)
            for r in range(world):
                if r == rmax:
                    continue
                has = any(v[0] == u[0] for v in assign[r])
                peak = max(src_after, total(r) + cost(u) + (0 if has else compute_powers_cost))
                if peak < tmax and (best is None or (peak, order[u], r) < best):
                    best = (peak, order[u], r)
        if best is None:
            break
        u = units[best[1]]
        assign[rmax].remove(u)
        assign[best[2]].append(u)
    return assign


def gather_slots(assign):
    """-> (max_local, {(bundle_idx, cache_idx): row}) rows of the all-gathered [world*max_local] result table."""
    world = len(assign)
    max_local = max((len(v) for v in assign.values()), default=0)
    rows = {}
    for r in range(world):
        for i, u in enumerate(assign[r]):
            rows[(u[0], u[1])] = r * max_local + i
    return max_local, rows
