"""Instruction mix per kernel (and of its largest loop body) from hipcc -S output.  usage: isa_stats.py file.s [substr ...]"""
import collections
import re
import sys

src = open(sys.argv[1]).read()
subs = sys.argv[2:]
# kernels: from "name:" label to ".Lfunc_end"
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", src, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if subs and not any(s in name for s in subs):
        continue
    lines = [l.strip() for l in body.split("\n")]
    labels = {}
    ins = []
    for l in lines:
        if not l or l.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", l):
            continue
        if re.match(r"^\.LBB\d+_\d+:", l):
            labels[l[:-1]] = len(ins)
            continue
        ins.append(l)
    # loops = backward branches
    loops = []
    for i, l in enumerate(ins):
        mm = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] <= i:
            loops.append((labels[mm.group(1)], i))

    def mix(seq):
        g = collections.Counter()
        for l in seq:
            op = l.split()[0]
            if op.startswith("v_pk_"):
                g["v_pk"] += 1
            elif op in ("v_rcp_f32_e32", "v_rsq_f32_e32", "v_sqrt_f32_e32"):
                g["v_trans"] += 1
            elif op.startswith("v_mov") or op.startswith("v_accvgpr"):
                g["v_mov"] += 1
            elif op.startswith("v_"):
                g["v_other"] += 1
            elif op.startswith("s_waitcnt"):
                g["s_waitcnt"] += 1
            elif op.startswith("s_nop"):
                g["s_nop"] += 1
            elif op.startswith("s_load"):
                g["s_load"] += 1
            elif op.startswith("s_"):
                g["salu"] += 1
            elif op.startswith("ds_"):
                g["lds"] += 1
            elif op.startswith("scratch"):
                g["scratch"] += 1
            elif op.startswith(("global", "flat", "buffer")):
                g["vmem"] += 1
            else:
                g["other"] += 1
        return dict(g)

    print(name[:90])
    print("  whole kernel:", len(ins), mix(ins))
    for a, b in sorted(loops, key=lambda x: x[0] - x[1])[:3]:
        print(f"  loop [{a}:{b}] {b - a + 1} instrs:", mix(ins[a:b + 1]))
