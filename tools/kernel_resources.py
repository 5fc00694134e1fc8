"""Registers, LDS and occupancy of every kernel as the compiler reports them (no GPU needed):
    python tools/kernel_resources.py profiles/<tag>/kernel_resources.md"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bevy_amd import build as mi_build  # the library's own flags, per-file ones included (FILE_FLAGS)
out = []
for f in ("kernels_flat.hip", "kernels_tree.hip", "kernels_cluster.hip", "kernels_batch.hip", "kernels_sorted.hip", "kernels_cells.hip"):
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + [x for x in mi_build.FLAGS if x != "-shared"] + mi_build.FILE_FLAGS.get(f, [])
                       + ["-x", "hip", "-c", os.path.join(ROOT, "bevy_amd", "csrc", f), "-o", "/tmp/_kr.o",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur, d = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur, d = m.group(1), {}
        for k, rx in (("vgpr", r" VGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                      ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m2 = re.search(rx, line)
            if m2 and cur:
                d[k] = int(m2.group(1))
        if "LDS Size" in line and cur:
            dem = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", "").replace("void ", "")).replace("mi::", "")
            out.append((f, name, d))
dst = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
with open(dst, "w") as fh:
    fh.write("# Registers, LDS and occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; tools/kernel_resources.py)\n\n")
    fh.write("| file | kernel | VGPRs | VGPR spills | scratch B/lane | LDS B/workgroup | waves / SIMD |\n|---|---|---|---|---|---|---|\n")
    for f, name, d in out:
        fh.write(f"| {f} | `{name}` | {d.get('vgpr')} | {d.get('spill')} | {d.get('scratch')} | {d.get('lds')} | {d.get('occ')} |\n")
print(len(out), "kernels")
