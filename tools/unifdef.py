#!/usr/bin/env python3
"""Resolve #ifdef / #ifndef / #if defined(..) [&& defined(..)] / #else / #endif blocks of the given files IN PLACE for a set of
macros known to be defined (-DNAME) or undefined (-UNAME); conditionals on other macros are left alone.
    python tools/unifdef.py -DB9_LATE_OBS -UB9_ABL_NOBIN ... file ..."""
import re, sys
defs, files = {}, []
for a in sys.argv[1:]:
    if a.startswith("-D"): defs[a[2:]] = True
    elif a.startswith("-U"): defs[a[2:]] = False
    else: files.append(a)

def evaluate(line):
    m = re.match(r"\s*#\s*ifdef\s+(\w+)", line)
    if m: return defs.get(m.group(1))
    m = re.match(r"\s*#\s*ifndef\s+(\w+)", line)
    if m: return None if m.group(1) not in defs else not defs[m.group(1)]
    m = re.match(r"\s*#\s*if\s+(.*)", line)
    if m:
        terms = [t.strip() for t in m.group(1).split("//")[0].split("&&")]
        vals = []
        for t in terms:
            mm = re.fullmatch(r"defined\s*\(\s*(\w+)\s*\)", t)
            if not mm or mm.group(1) not in defs: return None
            vals.append(defs[mm.group(1)])
        return all(vals)
    return None

for path in files:
    out, stack = [], []          # stack entries: [known (bool or None), taking_now, parent_emitting]
    emitting = True
    for line in open(path).read().split("\n"):
        s = line.strip()
        if re.match(r"#\s*(ifdef|ifndef|if)\b", s):
            v = evaluate(line)
            stack.append([v, v if v is not None else True, emitting])
            if v is None and emitting: out.append(line)
            emitting = emitting and (v is None or v)
            continue
        if re.match(r"#\s*elif\b", s) and stack:
            assert stack[-1][0] is None, f"{path}: #elif on a resolved conditional is not supported"
            if emitting or stack[-1][2]: out.append(line)
            continue
        if re.match(r"#\s*else\b", s) and stack:
            v, taking, parent = stack[-1]
            if v is None:
                if parent: out.append(line)
            else:
                stack[-1][1] = not taking
                emitting = parent and stack[-1][1]
            continue
        if re.match(r"#\s*endif\b", s) and stack:
            v, taking, parent = stack.pop()
            if v is None and parent: out.append(line)
            emitting = parent
            continue
        if emitting: out.append(line)
    assert not stack, f"{path}: unbalanced conditionals"
    open(path, "w").write("\n".join(out))
