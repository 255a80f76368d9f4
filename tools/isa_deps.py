"""Static look at a kernel's ISA: for every straight-line segment (between labels / branches / barriers) longer than
MIN instructions, the histogram of the distance (in instructions) from each instruction to the nearest producer of one
of its VGPR sources.  A lone gfx950 wavefront issues an independent VALU op every ~2.5 cycles but a dependent one only
every ~5-8 (tools/micro/issue_rate.hip), so distance-1/2 pairs are stalls.
    python tools/isa_deps.py file.s kernel_name_substring [MIN]
"""
import collections, re, sys

def regs(tok):
    r = []
    for m in re.finditer(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b", tok):
        if m.group(1): r += [f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)]
        else: r.append(f"{m.group(4)}{m.group(5)}")
    return r

def analyse(ins):
    last = {}; hist = collections.Counter(); kinds = collections.Counter(); est = 0.0
    for i, l in enumerate(ins):
        op, _, rest = l.partition(" ")
        ops = rest.split(",")
        kinds[op.replace("_e32", "").replace("_e64", "")] += 1
        dst = regs(ops[0]) if ops else []
        srcs = [r for o in ops[1:] for r in regs(o)]
        if op.startswith(("v_fmac", "v_mac", "v_pk_fmac")): srcs += dst
        d = min([i - last[s] for s in srcs if s in last and s[0] in "va"], default=99)
        hist[min(d, 6)] += 1
        if not op.startswith(("s_", "ds_write", "global_store", "buffer_store", "scratch_store")):
            for r in dst: last[r] = i
    return hist, kinds

if __name__ == "__main__":
    txt = open(sys.argv[1]).read(); key = sys.argv[2]; MIN = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    f = [f for f in re.split(r"\n(?=_Z\w+:)", txt) if key in f.split(":")[0]][0]
    seg = []; k = 0
    for l in f.split("\n") + [".LBBend:"]:
        st = l.strip()
        brk = re.match(r"^\.LBB\d+_\d+:", l) or st.startswith(("s_cbranch", "s_branch", "s_barrier", ".LBBend"))
        if l.startswith("\t") and st and not st.startswith((".", ";")) and not brk: seg.append(st)
        if brk:
            if len(seg) >= MIN:
                h, kinds = analyse(seg); n = len(seg)
                print(f"segment {k:3d} ending at '{st[:28]}': {n:4d} instr; producer distance 1:{h[1]} 2:{h[2]} 3:{h[3]} 4:{h[4]} 5:{h[5]} 6+:{h[6]}   "
                      f"accvgpr moves {sum(v for o, v in kinds.items() if 'accvgpr' in o)}, v_mov {kinds['v_mov_b32']}, ds {sum(v for o, v in kinds.items() if o.startswith('ds_'))}")
            seg = []; k += 1
