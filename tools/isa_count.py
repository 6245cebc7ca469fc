#!/usr/bin/env python3
"""Instruction census of the device code: tools/isa_count.py <file.s> [name-substring]."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
filt = sys.argv[2] if len(sys.argv) > 2 else "osfir"
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\S+):\s', s, re.M)]
for i, (pos, name) in enumerate(starts):
    if filt not in name:
        continue
    end = s.find("s_endpgm", pos)
    body = s[pos:end]
    ins = []
    for l in body.split("\n"):
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        ins.append(t.split()[0])
    c = collections.Counter(ins)
    f64 = sum(v for k, v in c.items() if "_f64" in k)
    print(name[:70])
    print("  total %d  f64-valu %d  other-valu %d  ds %d  global %d  salu %d  waitcnt %d  barrier %d" % (
        len(ins), f64, sum(v for k, v in c.items() if k.startswith("v_") and "_f64" not in k),
        sum(v for k, v in c.items() if k.startswith("ds_")), sum(v for k, v in c.items() if k.startswith("global_")),
        sum(v for k, v in c.items() if k.startswith("s_") and k not in ("s_waitcnt", "s_barrier")),
        c["s_waitcnt"], c["s_barrier"]))
    print("  ", c.most_common(22))
