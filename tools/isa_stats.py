#!/usr/bin/env python3
"""Instruction histogram of the main loop of a kernel in hipcc -S output."""
import re
import sys
from collections import Counter

path, sym = sys.argv[1], sys.argv[2]
s = open(path).read()
i = s.index("\n" + sym)
j = s.index("s_endpgm", i)
body = s[i:j].split("\n")
hdr = [n for n, l in enumerate(body) if "Loop Header" in l]
start = hdr[0] if hdr else 0
end = max(n for n, l in enumerate(body) if re.search(r"s_cbranch|s_branch", l))
loop = body[start:end]
cnt = Counter()
for l in loop:
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if m:
        cnt[m.group(1)] += 1
valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
print("loop lines %d..%d: %d instructions, %d VALU" % (start, end, sum(cnt.values()), valu))
for k, v in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print("  %-28s %d" % (k, v))
