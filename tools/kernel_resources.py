"""Register / LDS / scratch use of the library's kernels, from the code-object metadata of the built .so (no GPU needed):
    python tools/kernel_resources.py [substring ...]
Unbundles the gfx950 code object (llvm-objcopy + clang-offload-bundler) and reads the AMDGPU metadata notes."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.path.join(ROOT, "council-gan_amd", "lib", "libcouncilgan_hip.so")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "cg.co")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + co])
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    dem = "c++filt"
rows = []
for k in re.split(r'\n\s*- \.agpr_count:', notes)[1:]:
    name = re.search(r'\.name:\s*(\S+)', k).group(1)
    f = lambda key: int(re.search(r'\.%s:\s*(\d+)' % key, k).group(1))
    rows.append((name, int(k.split('\n')[0].strip()), f('vgpr_count'), f('vgpr_spill_count'), f('sgpr_count'),
                 f('group_segment_fixed_size'), f('private_segment_fixed_size')))
names = subprocess.run([dem] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
pats = sys.argv[1:]
print("%-86s %5s %5s %5s %5s %7s %7s %6s" % ("kernel", "agpr", "vgpr", "spill", "sgpr", "lds", "scratch", "w/SIMD"))
for (raw, ag, vg, sp, sg, lds, scr), nm in zip(rows, names):
    nm = re.sub(r'\(anonymous namespace\)::', '', nm).split('(')[0].replace('void ', '')
    if pats and not any(p in nm for p in pats):
        continue
    alloc = (vg + 7) // 8 * 8
    print("%-86s %5d %5d %5d %5d %7d %7d %6d" % (nm[:86], ag, vg, sp, sg, lds, scr, min(8, 512 // max(alloc, 1))))
