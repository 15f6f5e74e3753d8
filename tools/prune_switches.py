#!/usr/bin/env python3
"""A small `unifdef`: resolves the preprocessor conditionals of the given sources for a fixed set of macros and deletes the
branches that can no longer be taken (round 6, VERDICT r5 weak 18: ~60 compile-time switches lived in the product kernels, most of
them measured-and-rejected variants whose story DESIGN_LOG.md already tells).

    tools/prune_switches.py [--dry] NAME[=VALUE] ... -- file ...

NAME alone: the macro is never defined; NAME=VALUE: it is defined to VALUE (an integer) and its `#ifndef NAME / #define NAME VALUE /
#endif` block goes too.  A conditional is resolved only when its condition has ONE value under every assignment of the macros this
tool was NOT told about; anything else is left as it is and reported."""
import itertools
import re
import sys

TOK = re.compile(r"\s*(defined\s*\(\s*\w+\s*\)|defined\s+\w+|\w+|&&|\|\||==|!=|<=|>=|[()!<>+\-*/%])")


def tokenize(e):
    e = re.sub(r"/\*.*?\*/", " ", e)
    e = e.split("//")[0]
    out, i = [], 0
    while i < len(e):
        if e[i].isspace():
            i += 1
            continue
        m = TOK.match(e, i)
        if not m:
            return None
        out.append(m.group(1))
        i = m.end()
    return out


def evaluate(expr, known):
    """True / False when the expression has one value whatever the unknown macros are; None otherwise"""
    toks = tokenize(expr)
    if toks is None:
        return None
    atoms, py = {}, []
    for t in toks:
        m = re.match(r"defined\s*\(?\s*(\w+)\s*\)?", t)
        if m:
            n = m.group(1)
            if n in known:
                py.append("1" if known[n] is not None else "0")
            else:
                py.append(atoms.setdefault("D_" + n, "a%d" % len(atoms)))
        elif re.match(r"[A-Za-z_]\w*$", t):
            if t in known:
                py.append(str(known[t]) if known[t] is not None else "0")      # (an undefined macro evaluates to 0 in #if)
            else:
                py.append(atoms.setdefault("V_" + t, "a%d" % len(atoms)))
        elif t == "&&":
            py.append(" and ")
        elif t == "||":
            py.append(" or ")
        elif t == "!":
            py.append(" not ")
        elif t == "/":
            py.append("//")
        else:
            py.append(t)
    src = "".join(py)
    names = sorted(atoms.values())
    vals = set()
    for combo in itertools.product((0, 1, 7), repeat=len(names)):
        try:
            vals.add(bool(eval(src, {"__builtins__": {}}, dict(zip(names, combo)))))
        except Exception:
            return None
    return vals.pop() if len(vals) == 1 else None


def prune(lines, known, path, report):
    out = []
    stack = []      # frames: dict(resolved, emitting, taken, parent_emit)
    i = 0
    mentions = re.compile(r"\b(" + "|".join(map(re.escape, known)) + r")\b") if known else None
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        emit = all(f["emit"] for f in stack)
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            # the `#define NAME VALUE` of a resolved `#ifndef NAME` block for a valued macro disappears with its frame
            if emit:
                out.append(ln)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2).strip()
        # line continuations
        while ln.rstrip().endswith("\\") and i + 1 < len(lines):
            i += 1
            ln = ln.rstrip()[:-1] + " " + lines[i]
            rest = re.match(r"#\s*\w+\b(.*)", ln.strip(), re.S).group(1).strip()
        if kind in ("ifdef", "ifndef", "if"):
            cond = rest
            if kind == "ifdef":
                cond = "defined(%s)" % rest.split()[0]
            elif kind == "ifndef":
                cond = "!defined(%s)" % rest.split()[0]
            v = evaluate(cond, known) if (mentions and mentions.search(cond)) else None
            # `#ifndef NAME / #define NAME v / #endif` of a valued macro: drop the whole block
            if kind == "ifndef" and rest.split()[0] in known and known[rest.split()[0]] is not None:
                v = False
            if v is None and mentions and mentions.search(cond) and emit:
                report.append("%s:%d: left as it is: #%s %s" % (path, i + 1, kind, rest))
            f = dict(resolved=v is not None, emit=True, taken=False, any_unknown=False)
            if v is None:
                if emit:
                    out.append(ln)
            else:
                f["emit"] = bool(v)
                f["taken"] = bool(v)
            stack.append(f)
        elif kind == "elif":
            f = stack[-1]
            if not f["resolved"]:
                if all(g["emit"] for g in stack[:-1]):
                    out.append(ln)
                    if mentions and mentions.search(rest):
                        report.append("%s:%d: left as it is: #elif %s" % (path, i + 1, rest))
            else:
                v = evaluate(rest, known) if not f["taken"] else False
                if v is None:
                    raise SystemExit("%s:%d: #elif with an open condition behind a resolved #if: edit by hand" % (path, i + 1))
                f["emit"] = bool(v)
                f["taken"] = f["taken"] or bool(v)
        elif kind == "else":
            f = stack[-1]
            if not f["resolved"]:
                if all(g["emit"] for g in stack[:-1]):
                    out.append(ln)
            else:
                f["emit"] = not f["taken"]
                f["taken"] = True
        else:  # endif
            f = stack.pop()
            if not f["resolved"] and all(g["emit"] for g in stack):
                out.append(ln)
        i += 1
    if stack:
        raise SystemExit("%s: unbalanced conditionals" % path)
    return out


def main(argv):
    dry = "--dry" in argv
    argv = [a for a in argv if a != "--dry"]
    k = argv.index("--")
    known = {}
    for spec in argv[:k]:
        n, _, v = spec.partition("=")
        known[n] = int(v) if v else None
    report = []
    for path in argv[k + 1:]:
        src = open(path).read().split("\n")
        new = prune(src, known, path, report)
        if new != src:
            print("%s: %d -> %d lines" % (path, len(src), len(new)))
            if not dry:
                open(path, "w").write("\n".join(new))
    for r in report:
        print(r)


if __name__ == "__main__":
    main(sys.argv[1:])
