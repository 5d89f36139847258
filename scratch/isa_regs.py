"""Register use of the kernels in a gfx950 assembly listing (hipcc --cuda-device-only -S): name, VGPRs, spills.
usage: python scratch/isa_regs.py file.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or pat not in name.group(1):
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)
    print(name.group(1)[:110], "vgpr", g("vgpr_count"), "vspill", g("vgpr_spill_count"), "sgpr", g("sgpr_count"), "sspill", g("sgpr_spill_count"), "lds", g("group_segment_fixed_size"))
