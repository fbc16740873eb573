#!/usr/bin/env python3
"""Static estimate of the vector unit's cycles per instruction of each kernel in a `hipcc -S` listing, from the issue costs
measured by tools/micro/issue_cost.hip on gfx950: 8 cycles for transcendentals (v_rcp, v_sqrt, ...), 2 cycles for plain fp32 /
integer add, subtract, multiply, fma, move, and, or, xor, right shift when no operand is a scalar register, 4 cycles for
everything else (compares, selects, min / max, left shifts, three-operand integer forms, conversions, all fp64, packed,
cross-lane) and for the 2-cycle forms when an operand is a scalar register.
usage: isa_cycles.py listing.s [KERNEL_SUBSTRING]
       isa_cycles.py --json listing.s > profiles/isa_cycles.json   (kernels by demangled name; tools/make_isa_cycles.sh)"""
import collections, json, re, subprocess, sys
as_json = len(sys.argv) > 1 and sys.argv[1] == "--json"
if as_json: del sys.argv[1]
table = {}
FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32")
SLOW8 = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
def cost(op, args):
    if op.startswith(SLOW8): return 8
    base = op.replace("_e32", "").replace("_e64", "")
    if base in FAST:
        if re.search(r"\bs\d+\b|\bs\[|\bvcc|\bexec|\bm0\b|\bttmp", args) and not base.endswith("co_u32"): return 4
        if base.endswith("co_u32"): return 4
        return 2
    return 4
want = sys.argv[2] if len(sys.argv) > 2 else ""
name, acc = None, None
def flush():
    if name and acc and acc["n"] and as_json:
        table[name] = {"valu_instructions_static": acc["n"], "cycles_per_instruction": round(acc["c"] / acc["n"], 3),
                       "by_cost": {"2": acc[2], "4": acc[4], "8": acc[8]}}
    elif name and want in name and acc["n"]:
        print(f"{name[:46]:46s} valu {acc['n']:5d}  cycles/instruction {acc['c'] / acc['n']:.2f}  (2-cycle {acc[2]:4d}, 4-cycle {acc[4]:4d}, 8-cycle {acc[8]:3d})")
for l in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        flush(); name, acc = m.group(1), collections.Counter(); continue
    if l.startswith(".Lfunc_end"):
        flush(); name = None; continue
    if name is None or not l.startswith("\t"): continue
    t = l.strip().split(None, 1)
    if not t or not t[0].startswith("v_"): continue
    c = cost(t[0], t[1] if len(t) > 1 else "")
    acc["n"] += 1; acc["c"] += c; acc[c] += 1
flush()
if as_json:
    names = list(table)
    plain = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for mangled, nice in zip(names, plain):
        nice = nice.split("(")[0].replace("void ", "").strip()
        if nice.startswith(("rt_shade_kernel", "rt_tree_kernel", "rt_finish_kernel")): out[nice] = table[mangled]
    json.dump({"source": "tools/isa_cycles.py over `hipcc -S` of rt_gpu.hip with the library's flags: every vector instruction of a kernel priced with the issue "
                         "costs tools/micro/issue_cost.hip measured on gfx950 (profiles/r3c_issue_cost.txt): 2 cycles for fp32 / integer add, subtract, multiply, "
                         "fma, move, and, or, xor, right shift on vector-register operands, 8 for transcendentals, 4 for everything else and for any "
                         "instruction with a scalar-register operand; the static mix of the whole kernel stands in for the executed mix",
               "kernels": out}, sys.stdout, indent=1)
    print()
