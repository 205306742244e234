"""Compress the ISA of one kernel (hipcc -S output, one function) into per-barrier segments: G global load, S global store,
R ds_read, W ds_write, M mfma, v VALU, s SALU, [vmN] / [lgN] waits, | barrier.  usage: isa_segments.py file.s [first_seg [last_seg]]"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
seg, out = 0, []
cur = []
def flush():
    global cur, seg
    s = "".join(cur)
    # run-length compress
    s = re.sub(r"(v{4,})", lambda m: f"v{len(m.group(1))} ", s)
    if lo <= seg <= hi:
        cnt = {c: sum(1 for x in cur if x == c) for c in "GSRWMv"}
        print(f"--- segment {seg}: {cnt}")
        print(s)
    seg += 1
    cur = []
for ln in lines:
    t = ln.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_barrier":
        cur.append("|"); flush()
    elif op.startswith("v_mfma"):
        cur.append("M")
    elif op.startswith("global_load") or op.startswith("buffer_load"):
        cur.append("G")
    elif op.startswith("global_store") or op.startswith("buffer_store"):
        cur.append("S")
    elif op.startswith("ds_read") or op.startswith("ds_load"):
        cur.append("R")
    elif op.startswith("ds_write") or op.startswith("ds_store"):
        cur.append("W")
    elif op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", t); l = re.search(r"lgkmcnt\((\d+)\)", t)
        cur.append("[" + (f"vm{m.group(1)}" if m else "") + (f"lg{l.group(1)}" if l else "") + "]")
    elif op.startswith("v_"):
        cur.append("v")
    elif op.startswith("s_nop"):
        cur.append("n")
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        cur.append("B")
    elif op.startswith("s_"):
        cur.append("s")
flush()
