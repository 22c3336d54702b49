#!/usr/bin/env python
"""MFMA pipe utilisation per kernel from a tools/pmc_passes.sh pass that collected SQ_VALU_MFMA_BUSY_CYCLES and
GRBM_GUI_ACTIVE together: utilisation = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 x 1024) on the `steady` column.
usage: mfma_util.py pass_1.txt "title" > table.md"""
import re, sys
rows = {}
cmd = ""
for line in open(sys.argv[1]):
    if line.startswith("#"):
        cmd = line[1:].strip(); continue
    m = re.match(r"(.{72}) (\S+)\s+n=(\d+)\s+mean=(\S+) max=(\S+) steady\(n=(\d+)\)=(\S+)", line)
    if m:
        rows.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(7)))
print("# %s\n" % (sys.argv[2] if len(sys.argv) > 2 else "MFMA pipe utilisation per kernel"))
print("`%s`\n" % cmd)
print("`SQ_VALU_MFMA_BUSY_CYCLES` is summed over the 1024 SIMDs, `GRBM_GUI_ACTIVE` over the 8 XCDs; utilisation = MFMA_BUSY / "
      "(GRBM_GUI_ACTIVE / 8 x 1024), both from the steady column (the dispatches within 20 % of the largest).\n")
print("| kernel | launches | MFMA busy cycles | GUI active cycles | MFMA utilisation |\n|---|---:|---:|---:|---:|")
out = []
for k, v in rows.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"][1] > 0:
        b, g = v["SQ_VALU_MFMA_BUSY_CYCLES"][1], v["GRBM_GUI_ACTIVE"][1]
        if b > 0:
            out.append((b, "| `%s` | %d | %.3g | %.3g | %.1f %% |" % (k, v["GRBM_GUI_ACTIVE"][0], b, g, 100.0 * b / (g / 8 * 1024))))
for _, line in sorted(out, reverse=True):
    print(line)
