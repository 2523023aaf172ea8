"""Design aid: per kernel, the order of global loads (L), stores (S) and full memory-queue drains (W = s_waitcnt vmcnt(0))
in the compiled gfx950 code (`make asm` in csrc/ writes /tmp/sactd3_engine.s).  `LWLWLW` = serialised loads, `SWSWSW` or
`SLW` inside an epilogue = stores waiting for each other."""
import re, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/sactd3_engine.s").read()
for name in re.findall(r"^(_Z[\w]+):\s*;?.*$", txt, re.M):
    i = txt.index("\n" + name + ":"); j = txt.index(".Lfunc_end", i)
    ev = []
    for line in txt[i:j].splitlines():
        l = line.strip()
        if l.startswith("global_load") or l.startswith("buffer_load"): ev.append("L")
        elif l.startswith("global_store") or l.startswith("buffer_store"): ev.append("S")
        elif l.startswith("global_atomic"): ev.append("A")
        elif l.startswith("s_waitcnt") and "vmcnt(0)" in l: ev.append("W")
    seq = re.sub(r"(L{4,}|S{4,})", lambda m: "%s%d " % (m.group(0)[0], len(m.group(0))), "".join(ev))
    print("%-52s %s" % (name[:52], seq))
