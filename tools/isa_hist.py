"""Mnemonic histogram of the branch-free pair body of a stage-D kernel in a hipcc -S listing.
usage: python tools/isa_hist.py file.s [kernel-name substring] [marker mnemonic, default v_sqrt_f32]
The body is the basic block (between labels / branches) holding 24 marker instructions."""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_cdc_partial_groupedIfLi1ELi1ELi8ELi6ELb0E"
marker = sys.argv[3] if len(sys.argv) > 3 else "v_sqrt_f32"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if want in l and re.match(r"^_Z\S+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur = [], []
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(".LBB"):
            blocks.append(cur); cur = []
        continue
    if re.match(r"^\S+:(\s|$)", t):
        blocks.append(cur); cur = []
        continue
    cur.append(t.split()[0])
    if t.startswith("s_cbranch") or t.startswith("s_branch"):
        blocks.append(cur); cur = []
blocks.append(cur)
for b in blocks:
    n = sum(1 for m in b if m.startswith(marker))
    if n >= 12:
        print(f"== block with {n} x {marker}: {len(b)} instructions")
        for m, c in collections.Counter(b).most_common():
            print(f"  {m:34s} {c}")
name = lines[start].split(":")[0]
k = next(i for i, l in enumerate(lines) if l.strip().startswith(".amdhsa_kernel") and name in l)
for l in lines[k:k + 60]:
    if "next_free_vgpr" in l or "private_segment_fixed_size" in l:
        print(l.strip())
