"""Filter hipcc -Rpass-analysis=kernel-resource-usage output down to this repo's kernels."""
import re
import sys

cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        continue
    if cur is None or "msfl" not in cur or "rocprim" in cur:
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m:
        rows.setdefault(cur, {})[m.group(1).split(" ")[0]] = int(m.group(2))
    if "error" in line:
        print(line, end="")
print(f"{'kernel':70s} VGPR SGPR scratch occ  LDS")
for k, v in rows.items():
    name = re.sub(r"^_ZN4msfl\d+", "", k)[:70]
    print(f"{name:70s} {v.get('VGPRs',0):4d} {v.get('SGPRs',0):4d} {v.get('ScratchSize',0):7d} {v.get('Occupancy',0):3d} {v.get('LDS',0):5d}")
