"""VGPRs / SGPRs / spills / LDS / scratch of the kernels in a gfx950 object file (runs without a GPU):
python tools/kernel_resources.py apsu_amd/csrc/build/kernels.o [name-substring ...]"""
import os, re, subprocess, sys, tempfile
obj = os.path.abspath(sys.argv[1]); want = sys.argv[2:]
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    tmp = os.path.join(d, "k.o")
    subprocess.run(["cp", obj, tmp], check=True)
    subprocess.run([llvm + "/llvm-objdump", "--offloading", tmp], check=True, capture_output=True)
    co = [f for f in os.listdir(d) if "amdgcn" in f]
    if not co:
        sys.exit("no gfx code object in " + obj)
    notes = subprocess.run([llvm + "/llvm-readelf", "--notes", os.path.join(d, co[0])], check=True, capture_output=True, text=True).stdout
for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    m = re.search(r"\.name:\s+(\S+)", e)
    if not m:
        continue
    dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void apsu_he::", "")
    if want and not any(w in dem for w in want):
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", e).group(1)
    print("%-64s vgpr %3s sgpr %3s spill %2s lds %6s scratch %4s wg %4s" % (dem[:64], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"),
                                                                          g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
