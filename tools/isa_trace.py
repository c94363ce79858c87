"""Compressed instruction trace of one kernel in a hipcc -save-temps assembly file: runs of the same class are counted.
python tools/isa_trace.py <file.s> <mangled-name substring> [max lines]"""
import re, sys
s = open(sys.argv[1]).read()
name = [m for m in re.findall(r'^(_Z\S+):', s, re.M) if sys.argv[2] in m][0]
body = s[s.index("\n" + name + ":"):]
body = body[:body.index(".Lfunc_end")]
def cls(l):
    op = l.split()[0]
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "dsR"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "dsW"
    if op.startswith("buffer_load"): return "bufL"
    if op.startswith("global_load"): return "glbL"
    if op.startswith("global_store"): return "glbS"
    if op.startswith("s_waitcnt"): return "WAIT " + " ".join(l.split()[1:])
    if op.startswith("s_barrier"): return "BARRIER"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "BR " + l.split()[1]
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return op
out, last, cnt = [], None, 0
for l in body.split("\n"):
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        if l.startswith(".LBB"): 
            if last: out.append("%s x%d" % (last, cnt)); last, cnt = None, 0
            out.append(l)
        continue
    c = cls(l)
    if c == last: cnt += 1
    else:
        if last: out.append("%s x%d" % (last, cnt))
        last, cnt = c, 1
if last: out.append("%s x%d" % (last, cnt))
print(name, len(out), "lines")
print("\n".join(out[:int(sys.argv[3]) if len(sys.argv) > 3 else 400]))
