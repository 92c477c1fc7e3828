"""Instruction mix per basic block of one kernel in a hipcc -S listing: which blocks form loops (backward branches), and
what is in them (f64 FMA/mul/add, other VALU, SALU, SMEM, LDS, VMEM, scratch, waits, barriers).
    python tools/isa_mix.py file.s <kernel-name-substring> [min_instructions]"""
import re, sys, collections
path, want = sys.argv[1], sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and want in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur, order = collections.OrderedDict(), "entry", {}
blocks[cur] = []
for i in range(start + 1, end + 1):
    l = lines[i].split(";")[0].rstrip()
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        cur = m.group(1); blocks[cur] = []; continue
    t = l.strip()
    if not t or t.startswith(".") or t.startswith(";"): continue
    blocks[cur].append(t)
names = list(blocks)
pos = {n: i for i, n in enumerate(names)}
def cls(t):
    op = t.split()[0]
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"): return "fma64"
    if op.startswith(("v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64")): return "f64"
    if op.startswith(("v_cmp", "v_cndmask")): return "cmp/sel"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith(("v_mov", "v_accvgpr")): return "mov"
    if op.startswith(("v_div", "v_rcp", "v_trig", "v_ldexp", "v_frexp")): return "div"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    return "other"
back = collections.defaultdict(list)
for n, ins in blocks.items():
    for t in ins:
        m = re.match(r"s_cbranch\S*\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)", t)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in pos and pos[tgt] <= pos[n]: back[tgt].append(n)
print("loops (header <- latch): instructions in [header, latch]")
for h, ls in back.items():
    last = max(ls, key=lambda x: pos[x])
    body = [t for n in names[pos[h]:pos[last] + 1] for t in blocks[n]]
    if len(body) < minins: continue
    c = collections.Counter(cls(t) for t in body)
    valu = sum(c[k] for k in ("fma64", "f64", "cmp/sel", "lane", "mov", "div", "valu"))
    print(f"{h:>14} <- {last:<14} {len(body):6d} ins  VALU {valu:5d} | " + "  ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
