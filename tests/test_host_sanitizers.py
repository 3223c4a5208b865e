"""CPU tier: the product's host code that parses untrusted or structured input, built with AddressSanitizer +
UndefinedBehaviorSanitizer and driven through a fuzzing harness (tests/native/sanitize_harness.cpp): the N3 framing reader and the SEAL object codec (seeded / zlib ciphertexts, RelinKeys),
the PSUParams JSON reader with the derived constants, the PowersDag and the partition rule.  (GPU AddressSanitizer is not
available on this pool; the device side is covered by the bit-exact parity tests.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "apsu_amd", "csrc")


def test_host_parsers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_harness")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "native", "sanitize_harness.cpp")] + \
          [os.path.join(SRC, f) for f in ("wire.cpp", "seal_codec.cpp", "params.cpp", "powers_dag.cpp", "sharding.cpp")] + ["-lz", "-ldl", "-o", exe]
    subprocess.check_call(cmd)
    params = [os.path.join(ROOT, "tests", "params", f + ".json") for f in ("100K-1", "1M-1024-com", "1M-4096-32", "256M-4096")]
    # seeds for the saved-BinBundle reader come from the FlatBuffers model of tests/test_wire_framing.py (with and without a cache)
    from test_wire_framing import build_bin_bundle
    seeds = []
    for i, blobs in enumerate((None, [bytes([7]) * 40, bytes([9]) * 13])):
        path = tmp_path / ("seed%d.binbundle" % i)
        path.write_bytes(build_bin_bundle(2, 65537, [[5, 7, 11], [], [2**40 + 3], list(range(1, 9))], blobs))
        seeds.append(str(path))
    import struct
    from test_wire_framing import FbBuilder
    B = FbBuilder()
    hashed = B.struct_vector([struct.pack("<QQ", i, i + 1) for i in range(3)], 8)
    key = B.byte_vector(bytes(32))
    pv = B.byte_vector(b"params")
    hdr = tmp_path / "seed.dbheader"
    hdr.write_bytes(B.finish_size_prefixed(B.table([("off", pv), ("struct", struct.pack("<IIQ??", 0, 16, 3, False, False) + bytes(6), 8),
                                                    ("off", key), ("off", hashed), ("u32", 2)])))
    seeds.append(str(hdr))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe] + params + seeds, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert r.stdout.strip().endswith("ok") and "malformed buffers rejected" in r.stdout
