"""GPU tier: the WHOLE bench workload of the two large named configurations -- every target power of every bundle index
(Receiver::ComputePowers, receiver_osn.cpp:395-488) and every BinBundle (eval_patstock, bin_bundle.cpp:192-360) -- on the GPU
against the CPU oracle, bit for bit.  16M-4096: 72 powers x 4 indices, 28 BinBundles (6.2 GB of DB).  256M-4096: 322 powers x 3
indices, 102 BinBundles of degree 3999, 80 GB of DB in HBM -- BASELINE.json's "max DB, HBM-resident SenderDB, 288 GB/GPU
sizing" on one GPU; it needs ~100 GB of free HBM and is skipped on a smaller or busy device.
The run is tests/full_query_parity.py in a child process (the database is freed with it)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config,need_gb,expect", [("16M-4096", 16, "72 target powers x 4 bundle indices and 28 BinBundles compared, 0 mismatches"),
                                                   ("256M-4096", 100, "322 target powers x 3 bundle indices and 102 BinBundles compared, 0 mismatches")])
def test_full_workload_bit_exact(config, need_gb, expect):
    import torch
    free, total = torch.cuda.mem_get_info()
    if free < need_gb * (1 << 30):
        pytest.skip("%s needs ~%d GB of free HBM (%.0f GB free of %.0f)" % (config, need_gb, free / 2**30, total / 2**30))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "full_query_parity.py"), config], capture_output=True, text=True,
                       timeout=840, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "RESULT %s: %s" % (config, expect) in r.stdout, r.stdout[-1500:]
