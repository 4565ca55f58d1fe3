#!/usr/bin/env python3
"""Static instruction mix per kernel of an assembly listing (hipcc -S --cuda-device-only):
   python tests/probe/isa_mix.py file.s [name-substring]"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
name, body = None, []


def report(name, body):
    lines = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.strip().split(";")[0].strip().endswith(":")]
    c = Counter(l.split()[0] for l in lines)
    grp = Counter()
    for k, v in c.items():
        if k.startswith("v_"): grp["valu"] += v
        elif k.startswith("s_waitcnt"): grp["waitcnt"] += v
        elif k.startswith("s_"): grp["salu"] += v
        elif k.startswith("ds_"): grp["lds"] += v
        elif k.startswith(("global_", "buffer_", "flat_", "scratch_")): grp["vmem"] += v
        else: grp["other"] += v
    print(name[:90], len(lines), dict(grp))
    print("    ", c.most_common(16))


for l in txt:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        if name and want in name:
            report(name, body)
        name, body = m.group(1), []
    elif l.startswith(".Lfunc_end"):
        if name and want in name:
            report(name, body)
        name, body = None, []
    elif name:
        body.append(l)


def loops(name, body):
    """innermost loops: (first line, instructions in the body) for every backward branch"""
    labels, out, n = {}, [], 0
    for l in body:
        s = l.strip().split(";")[0].strip()
        if not s or s.startswith((".", "//")) and not s.endswith(":"):
            continue
        if s.endswith(":"):
            labels[s[:-1]] = n
            continue
        n += 1
        m = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", s)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels:
                out.append((labels[tgt], n - labels[tgt]))
    return out


if len(sys.argv) > 3 and sys.argv[3] == "loops":
    name, body = None, []
    for l in txt:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, body = m.group(1), []
        elif l.startswith(".Lfunc_end"):
            if name and want in name:
                print(name[:80], sorted(loops(name, body)))
            name = None
        elif name is not None:
            body.append(l)
