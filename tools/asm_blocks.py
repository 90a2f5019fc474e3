"""Basic blocks of one kernel in a device assembly dump, with instruction counts and a rough issue-cycle estimate
(f64 / 64-bit-integer VALU 8 cycles, transcendental and 32-bit integer multiplies 16, everything else 4 per wave64).
   python tools/asm_blocks.py /tmp/k.s <mangled-name regex> [min instructions]"""
import re
import sys

t = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 12
st = next(i for i, l in enumerate(t) if re.match(r"^" + pat + r"\S*:", l))
end = next(i for i in range(st, len(t)) if "s_endpgm" in t[i])
blocks, cur = [], None
for l in t[st:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), []]
        blocks.append(cur)
        continue
    if cur is None:
        cur = ["entry", []]
        blocks.append(cur)
    s = l.strip()
    if l.startswith("\t") and s and not s.startswith((".", ";")):
        cur[1].append(s)


def cost(ins):
    c = 0
    for i in ins:
        op = i.split()[0]
        if op.startswith("v_") and ("f64" in op or "u64" in op or "i64" in op):
            c += 16 if any(k in op for k in ("rcp", "sqrt", "rsq")) else 8
        elif op.startswith("v_") and any(k in op for k in ("mul_lo", "mul_hi", "exp_f32", "log_f32", "rcp_f32", "sqrt_f32", "rsq_f32", "sin_", "cos_")):
            c += 16
        else:
            c += 4
    return c


tot = 0
for name, ins in blocks:
    tot += len(ins)
    if len(ins) >= least:
        v = sum(i.startswith("v_") for i in ins)
        f64 = sum("f64" in i.split()[0] for i in ins)
        tr = [i.split()[0] for i in ins if any(k in i.split()[0] for k in ("exp_f", "log_f", "rcp_f", "sqrt_f", "rsq_f", "sin_f", "cos_f", "div_scale"))]
        print(f"{name:12s} n={len(ins):4d} valu={v:4d} f64={f64:4d} ~cycles={cost(ins):5d}  {' '.join(sorted(set(tr)))}")
print("total instructions", tot)
