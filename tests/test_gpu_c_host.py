"""The C ABI driven from a plain C program (examples/c_abi_demo.c): built with gcc against include/apsu_he.h and the
shared library, run on the GPU, and compared with the same calls through the Python binding."""
import os
import subprocess
import sys

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import apsu_amd                                            # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK64 = (1 << 64) - 1


def fnv(a):
    h = 0xcbf29ce484222325
    for b in np.ascontiguousarray(a).view(np.uint8).tobytes():
        h = ((h ^ b) * 0x100000001b3) & MASK64
    return "%016x" % h


def mix(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def test_c_program_matches_python_binding(tmp_path):
    exe = str(tmp_path / "c_abi_demo")
    libdir = os.path.join(ROOT, "apsu_amd")
    subprocess.check_call(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + libdir, "-lapsu_he_gpu", "-L/opt/rocm/lib",
                           "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    params = os.path.join(ROOT, "tests", "params", "1M-1024-com.json")
    r = subprocess.run([exe, params], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines() if " " in l)
    assert lines["roundtrip"] == "ok" and lines["missing-powers"].startswith("status -1") and "done" in r.stdout
    assert lines["multi"].startswith("ok") and lines["wire"].startswith("ok"), r.stdout
    assert lines["pinned"].startswith("ok") and "Receiver::RunQuery" in lines["pinned"] and lines["seal"].startswith("ok"), r.stdout
    assert lines["dbfile"].startswith("ok (1 BinBundle"), r.stdout

    # the same inputs through the Python binding
    G = apsu_amd.HeContext(open(params).read())
    n, K, first = G.n, G.K, G.first_chain_idx
    Lf = first + 1
    q = [int(x) for x in G.q]

    def pseudo(base, shape, limb_axis):
        idx = np.arange(int(np.prod(shape)), dtype=np.uint64).reshape(shape)
        v = mix(np.uint64(base) + idx)
        mods = np.array(q, dtype=np.uint64)[:shape[limb_axis]].reshape([-1 if a == limb_axis else 1 for a in range(len(shape))])
        return v % mods

    ct = pseudo(1, (2, Lf, n), 1)
    G.transform_to_ntt_inplace(ct, first)
    assert lines["ntt"] == fnv(ct)
    ns = G.source_power_count
    src = pseudo(77, (ns, 2, Lf, n), 2)
    rk = None
    if K > 1:
        rk = G.upload_relin_keys(pseudo(1000003, (K - 1, 2, K, n), 2))
    pw = G.compute_powers([0], [[src[s] for s in range(ns)]], rk)
    bundle = G.random_bundle(0, 0, G.max_items_per_bin - 1, 4242)
    buf = torch.empty(n, dtype=torch.int64, device="cuda")
    _, blocks = G.mask_generate_blake2xb([int(mix(np.uint64(99 + i))) for i in range(8)], 1, buf.data_ptr(), want_values=False)
    assert lines["blocks"] == fnv(blocks)
    out = G.eval_bundles([bundle], pw, rk, [buf.data_ptr()], masks_on_device=True)
    assert lines["result"] == fnv(out)
    item = np.array([0x10 * (i % 8) + 0x0f - i for i in range(16)], dtype=np.uint8)
    assert lines["felts"] == fnv(G.algebraize_items(item)[0, :2])
