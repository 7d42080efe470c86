#!/usr/bin/env python3
"""Resolve the preprocessor conditionals that test a given set of macros and leave every other line alone.

    tools/dev/unifdef.py FILE NAME=VALUE ... NAME=undef ...   (rewrites FILE in place)

Used in round 6 to take settled build switches out of the product sources (the switches live on as patches under
tools/dev/experiments/).  A conditional is resolved only when EVERY identifier it tests is in the given set; `#define NAME ...`
lines of resolved macros inside a resolved `#ifndef NAME` default block go with the block.  Uses of a valued macro in ordinary
code lines are replaced by the value."""
import re
import sys


def main():
    path = sys.argv[1]
    known = {}
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        known[k] = None if v == "undef" else v
    ident = re.compile(r"[A-Za-z_]\w*")

    def evaluate(kind, rest):
        """-> True/False if resolvable, else None"""
        rest = rest.split("//")[0].strip()
        if kind in ("ifdef", "ifndef"):
            name = rest.split()[0]
            if name not in known:
                return None
            d = known[name] is not None
            return d if kind == "ifdef" else not d
        expr = rest
        names = set(ident.findall(re.sub(r"defined\s*\(\s*\w+\s*\)", "", expr))) - {"defined"}
        dnames = set(re.findall(r"defined\s*\(\s*(\w+)\s*\)", expr))
        if not (names | dnames) or not (names | dnames) <= set(known):
            return None
        expr = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if known[m.group(1)] is not None else "0", expr)
        expr = ident.sub(lambda m: known[m.group(0)] if known[m.group(0)] is not None else "0", expr)
        expr = expr.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
        return bool(eval(expr))

    out = []
    stack = []   # [resolved, keep_now, any_taken]
    for ln in open(path).read().split("\n"):
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", ln)
        live = all(s[1] for s in stack)
        if m:
            kind, rest = m.group(1), m.group(2)
            if kind in ("ifdef", "ifndef", "if"):
                v = evaluate(kind, rest)
                if v is None:
                    stack.append([False, True, True])
                    if live:
                        out.append(ln)
                else:
                    stack.append([True, v, v])
                continue
            top = stack[-1]
            if kind == "elif":
                if top[0]:
                    v = evaluate("if", rest)
                    if v is None:
                        # every earlier branch resolved false: the chain continues as an ordinary #if
                        assert not top[2], "half-resolved #elif behind a taken branch: " + ln
                        stack[-1] = [False, True, True]
                        if all(s[1] for s in stack):
                            out.append(re.sub(r"#(\s*)elif", r"#\1if", ln, count=1))
                        continue
                    top[1] = (not top[2]) and v
                    top[2] = top[2] or v
                elif live:
                    out.append(ln)
                continue
            if kind == "else":
                if top[0]:
                    top[1] = not top[2]
                    top[2] = True
                elif live:
                    out.append(ln)
                continue
            stack.pop()
            if not top[0] and all(s[1] for s in stack):
                out.append(ln)
            continue
        if not live:
            continue
        dm = re.match(r"\s*#\s*define\s+(\w+)\b", ln)
        if dm and dm.group(1) in known:
            continue
        if any(k in ln for k, v in known.items() if v is not None) and not ln.lstrip().startswith("//"):
            code, sep, comment = ln.partition("//")
            code = ident.sub(lambda mm: known[mm.group(0)] if known.get(mm.group(0)) is not None else mm.group(0), code)
            ln = code + sep + comment
        out.append(ln)
    assert not stack
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
