#!/usr/bin/env python3
"""A small `unifdef`: resolves the preprocessor conditionals of a source file for a given set of macro values and removes the
conditionals (and the `#ifndef X / #define X v / #endif` default blocks of those macros).  Used once per round to take the
experiment switches of a kernel out of the shipped source once the measurements they served are written down
(LABNOTES.md): python tools/strip_switches.py FILE NAME=VALUE ... [NAME=undef ...]
Conditionals that mention any macro NOT given on the command line are left alone."""
import re
import sys


def main():
    path = sys.argv[1]
    vals, undef = {}, set()
    for a in sys.argv[2:]:
        k, v = a.split("=", 1)
        if v == "undef":
            undef.add(k)
        else:
            vals[k] = v
    known = set(vals) | undef
    src = open(path).read().split("\n")

    def evaluate(expr):
        """-> True / False, or None if the expression mentions an unknown identifier"""
        e = re.sub(r"//.*$", "", expr).strip()
        e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: ("1" if m.group(1) in vals else "0") if m.group(1) in known else "?" + m.group(1), e)
        ids = set(re.findall(r"[A-Za-z_]\w*", e))
        if "?" in e or not ids <= known:
            return None
        for k in sorted(ids, key=len, reverse=True):
            e = re.sub(r"\b%s\b" % k, vals.get(k, "0"), e)
        e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
        return bool(eval(e))

    out = []
    stack = []   # entries: [resolved (bool), taking (bool: emit lines), any_taken (bool)] or [False, ...] for untouched conditionals
    i = 0

    def emitting():
        return all(s[1] for s in stack if s[0])

    while i < len(src):
        line = src[i]
        m = re.match(r"\s*#\s*(ifndef|ifdef|if|elif|else|endif)\b(.*)", line)
        if not m:
            # a default definition of a known macro outside any conditional we resolved away: drop it
            d = re.match(r"\s*#\s*define\s+(\w+)\b", line)
            if d and d.group(1) in known and emitting():
                i += 1
                continue
            if emitting():
                out.append(line)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            if kind == "if":
                v = evaluate(rest)
            else:
                name = rest.split()[0]
                v = None if name not in known else ((name in vals) == (kind == "ifdef"))
                # the `#ifndef X / #define X default / #endif` block of a macro we were GIVEN a value for: drop the whole block
                if kind == "ifndef" and name in vals:
                    v = False
            if v is None:
                stack.append([False, True, True])
                if emitting():
                    out.append(line)
            else:
                stack.append([True, v, v])
        elif kind in ("elif", "else"):
            top = stack[-1]
            if not top[0]:
                if emitting():
                    out.append(line)
            else:
                v = (evaluate(rest) if kind == "elif" else True)
                if v is None:
                    raise SystemExit("%s:%d: #elif on unknown macros inside a resolved conditional" % (path, i + 1))
                top[1] = (not top[2]) and v
                top[2] = top[2] or top[1]
        else:
            top = stack.pop()
            if not top[0] and emitting():
                out.append(line)
        i += 1
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
