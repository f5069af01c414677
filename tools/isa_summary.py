"""The pair body of every stage-D variant the bench and the facade run, from a `hipcc -S` listing of csrc/coldeltacor.hip: the
branch-free basic block that evaluates ONE (cell, neighbour) pair on ONE gene chunk (mnemonic histogram), its VALU count split into
the instructions the arithmetic of the elements needs (per-element recipe x elements per lane) and the rest (the transposing wave
reduction of the three / four moments, the LDS accumulation, pair bookkeeping), and VGPR / scratch use.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only velocyto.py_amd/csrc/coldeltacor.hip -o /tmp/cdc.s
       python tools/isa_summary.py /tmp/cdc.s > profiles/rNN_cdc_grouped_isa.txt"""
import collections
import re
import sys

path = sys.argv[1]
lines = open(path).read().split("\n")
# (label, mangled-name fragment, marker mnemonic, elements per lane and chunk, per-element recipe)
VARIANTS = [
    ("f32, no-pseudocount rule (production), single control: 8 cells x 1536 genes", "k_cdc_partial_groupedIfLi1ELi2ELi8ELi6ELb0E", "v_rsq_f32", 24,
     "v_sub (t), v_rsq_f32, v_mul_legacy (A = t rsq|t|), v_add (sum A), v_add |t| (sum A^2 = sum |t|), v_fmac (sum A b)"),
    ("f32, literal rule, single control: 8 cells x 1536 genes", "k_cdc_partial_groupedIfLi1ELi1ELi8ELi6ELb0E", "v_sqrt_f32", 24,
     "v_sub (t), v_mul |t| 2^54 clamp (c), v_fma (psc c + |t|), v_sqrt_f32, v_bfi (sign), v_add (sum A), 2 x v_fmac (sum A^2, sum A b)"),
    ("f64, literal rule (the reference's arithmetic, the bench headline), single control: 6 cells x 1024 genes", "k_cdc_partial_groupedIdLi1ELi1ELi6ELi8ELb0E", "v_rsq_f32", 16,
     "v_add_f64 (t), v_add_f64 (|t| + psc), v_cvt_f32_f64, v_cmp_f64 + v_cndmask (zero rule, on the argument of the seed), v_rsq_f32, 2 x v_mul_f32 (s0 = x_f y_f, h = y_f / 2), "
     "2 x v_cvt_f64_f32, 4 x v_fma_f64 (two Newton corrections), v_bfi (sign), v_add_f64 + 2 x v_fma_f64 (moments)"),
    ("f32, no-pseudocount rule, dual control (estimate_transition_prob's default): 6 cells x 1536 genes", "k_cdc_partial_groupedIfLi1ELi2ELi6ELi6ELb1E", "v_rsq_f32", 24,
     "as the single-control element + v_fmac (sum A b2)"),
    ("f64, literal rule, dual control: 4 cells x 1024 genes", "k_cdc_partial_groupedIdLi1ELi1ELi4ELi8ELb1E", "v_rsq_f32", 16, "as the single-control element + v_fma_f64 (sum A b2)"),
]
PER_ELEM = {0: 6, 1: 8, 2: 18, 3: 7, 4: 19}


def body(want, marker, n_marker):
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
    name = lines[start].split(":")[0]
    k = next(i for i, l in enumerate(lines) if l.strip().startswith(".amdhsa_kernel") and name in l)
    res = [l.strip() for l in lines[k:k + 60] if "next_free_vgpr" in l or "private_segment_fixed_size" in l]
    hits = [b for b in blocks if sum(1 for m in b if m.startswith(marker)) == n_marker]
    return hits[0], len(hits), res


print("# Pair bodies of k_cdc_partial_grouped (csrc/coldeltacor.hip), from the compiler's listing: tools/isa_summary.py")
print("# One body = one (cell, neighbour) pair on one gene chunk, evaluated by one wave (64 lanes x the elements per lane shown).")
for vi, (label, frag, marker, nel, recipe) in enumerate(VARIANTS):
    b, copies, res = body(frag, marker, nel)
    cnt = collections.Counter(b)
    valu = sum(c for m, c in cnt.items() if m.startswith("v_"))
    lds = sum(c for m, c in cnt.items() if m.startswith("ds_"))
    salu = sum(c for m, c in cnt.items() if m.startswith("s_") and not m.startswith("s_waitcnt") and not m.startswith("s_nop"))
    elem = PER_ELEM[vi] * nel
    print(f"\n== {label}")
    print(f"   kernel ...{frag}  ({copies} unrolled copies of the body in the row loop); {'; '.join(res)}")
    print(f"   {len(b)} instructions per pair-chunk: {valu} VALU, {lds} LDS reads, {salu} SALU, {cnt.get('s_waitcnt', 0)} s_waitcnt, {cnt.get('s_nop', 0)} s_nop")
    print(f"   elements per lane: {nel}; per element {PER_ELEM[vi]} VALU instructions ({recipe})")
    print(f"   => {elem} element instructions + {valu - elem} for the wave reduction of the moments (v_permlane*_swap, DPP adds), the first-vector initialisation "
          f"and pair bookkeeping = {100.0 * (valu - elem) / valu:.1f} % of the body's VALU instructions")
    for m, c in cnt.most_common():
        print(f"     {m:34s} {c}")
