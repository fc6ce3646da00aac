#!/usr/bin/env python3
"""One-off source transformation (round 4): every __global__ kernel of csrc/*.hip.h gets a leading `AsmBt bt` parameter and an
ASM_BARGS(bt, <its parameters>) prologue, so that one launch can serve several scenarios (asm_bt.hip.h).  Idempotent: kernels that
already have the parameter are left alone."""
import re
import sys

def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out

def pname(p):
    p = p.split("=")[0].strip()
    m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*$", p)
    return m.group(1)

def transform(txt):
    pos, out, n = 0, [], 0
    for m in re.finditer(r"__global__", txt):
        # the kernel name: first "void NAME(" after the attribute list
        mv = re.compile(r"void\s+([A-Za-z_][A-Za-z0-9_]*)\s*\(").search(txt, m.end())
        if not mv:
            continue
        start = mv.end()            # just after '('
        depth, i = 1, start
        while depth:
            c = txt[i]
            depth += c == "("
            depth -= c == ")"
            i += 1
        params = txt[start:i - 1]
        if "AsmBt" in params:
            continue
        j = txt.index("{", i)
        names = [pname(p) for p in split_params(params)]
        # strip default arguments: the batched launcher passes every argument explicitly
        clean = ", ".join(re.sub(r"\s+", " ", p.split("=")[0]).strip() for p in split_params(params))
        out.append(txt[pos:start])
        out.append("AsmBt abt, " + clean.lstrip())
        out.append(txt[i - 1:j + 1])
        # '#pragma clang fp ...' must stay the first thing of the compound statement
        mp = re.compile(r"\s*\n\s*#pragma clang fp[^\n]*").match(txt, j + 1)
        if mp:
            out.append(txt[j + 1:mp.end()])
            j = mp.end() - 1
        out.append("\n    ASM_BARGS(abt, " + ", ".join(names) + ");")
        pos = j + 1
        n += 1
    out.append(txt[pos:])
    return "".join(out), n

for path in sys.argv[1:]:
    src = open(path).read()
    new, n = transform(src)
    new = new.replace("blockIdx.z", "asm_bz(abt)")
    open(path, "w").write(new)
    print(path, n, "kernels")
