"""Static view of one kernel's hot basic blocks: python scripts/isa_blocks.py file.s kernel_substring [min_mem_ops]
(per block: instructions, 12-byte loads, 8-byte stores, scratch traffic, v_writelane / v_readlane, LDS ops, VALU)."""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 2
text = open(src).read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and key in l and l.rstrip().endswith(("E:", ":")) or (l.startswith("_Z") and key in l and ":" in l))
blocks, cur = [], None
for i in range(start, len(text)):
    l = text[i]
    if "s_endpgm" in l:
        break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = dict(name=m.group(1), n=0, ld3=0, st2=0, scr=0, wl=0, rl=0, ds=0, valu=0)
        blocks.append(cur)
        continue
    t = l.strip()
    if cur is None or not t or t[0] in ";.":
        continue
    cur["n"] += 1
    cur["ld3"] += "buffer_load_dwordx3" in t
    cur["st2"] += "buffer_store_dwordx2" in t
    cur["scr"] += "scratch_" in t
    cur["wl"] += "v_writelane" in t
    cur["rl"] += "v_readlane" in t
    cur["ds"] += t.startswith("ds_")
    cur["valu"] += t.startswith("v_")
tot = dict(n=0, scr=0)
for b in blocks:
    tot["n"] += b["n"]; tot["scr"] += b["scr"]
    if b["ld3"] >= floor or b["st2"] >= floor:
        print(b)
print("kernel: %d blocks, %d instructions, %d scratch ops" % (len(blocks), tot["n"], tot["scr"]))
