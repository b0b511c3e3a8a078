"""Register / spill / LDS summary of the kernels in a hipcc -S (--cuda-device-only) listing, filtered by a substring.

    hipcc -O3 ... --cuda-device-only -S -o x.s file.hip && python tools/kernel_regs.py x.s k_project_mfma
"""
import re
import sys

text = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in text.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or pat not in name.group(1):
        continue
    get = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)  # noqa: E731
    short = re.sub(r"^_ZN5msgat\d+", "", name.group(1))
    short = re.sub(r"EEvNS.*$|EvPK.*$|EEvPK.*$", "", short)
    print(f"{short:48s} vgpr {get('vgpr_count'):>4s} spill {get('vgpr_spill_count'):>3s} sgpr_spill {get('sgpr_spill_count'):>3s} "
          f"lds {get('group_segment_fixed_size'):>6s}")
