"""A compiler's first complaints, without a compiler: go/fennec_hip.go has never met a Go toolchain (there is none in this image;
DESIGN.md section 1), so the mistakes `go build` refuses outright are looked for here -- unbalanced delimiters, imports that are
not used, package qualifiers that are not imported, and locals that are declared and never read (all hard errors in Go).  A
tokenizer (comments, interpreted / raw strings, runes) and scope-blind name counting: a heuristic, not a type checker;
tests/test_go_shim_matches_header.py holds the C side of the same file against include/fennec_hip.h."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "go", "fennec_hip.go")

KEYWORDS = {"break", "case", "chan", "const", "continue", "default", "defer", "else", "fallthrough", "for", "func", "go", "goto", "if",
            "import", "interface", "map", "package", "range", "return", "select", "struct", "switch", "type", "var"}


def tokenize(src):
    """-> list of (kind, text, line); kinds: id, num, str, rune, op.  The cgo preamble comment in front of `import "C"` is a comment."""
    toks, i, line, n = [], 0, 1, len(src)
    while i < n:
        c = src[i]
        if c == "\n":
            line += 1; i += 1
        elif c in " \t\r":
            i += 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            assert j >= 0, f"line {line}: comment never closed"
            line += src.count("\n", i, j); i = j + 2
        elif c == "`":
            j = src.find("`", i + 1)
            assert j >= 0, f"line {line}: raw string never closed"
            toks.append(("str", src[i:j + 1], line)); line += src.count("\n", i, j); i = j + 1
        elif c == '"' or c == "'":
            j = i + 1
            while j < n and src[j] != c:
                assert src[j] != "\n", f"line {line}: literal runs past its line"
                j += 2 if src[j] == "\\" else 1
            assert j < n, f"line {line}: literal never closed"
            toks.append(("str" if c == '"' else "rune", src[i:j + 1], line)); i = j + 1
        elif c.isalpha() or c == "_":
            j = i + 1
            while j < n and (src[j].isalnum() or src[j] == "_"):
                j += 1
            toks.append(("id", src[i:j], line)); i = j
        elif c.isdigit():
            j = i + 1
            while j < n and (src[j].isalnum() or src[j] in "._"):
                j += 1
            toks.append(("num", src[i:j], line)); i = j
        else:
            for op in (":=", "...", "<<=", ">>=", "&^=", "&&", "||", "<-", "++", "--", "==", "!=", "<=", ">=", "+=", "-=", "*=", "/=",
                       "%=", "&=", "|=", "^=", "<<", ">>", "&^"):
                if src.startswith(op, i):
                    toks.append(("op", op, line)); i += len(op)
                    break
            else:
                toks.append(("op", c, line)); i += 1
    return toks


@pytest.fixture(scope="module")
def toks():
    with open(SHIM, encoding="utf-8") as f:
        return tokenize(f.read())


def test_delimiters_balance(toks):
    pairs = {")": "(", "]": "[", "}": "{"}
    stack = []
    for kind, t, line in toks:
        if kind != "op":
            continue
        if t in "([{":
            stack.append((t, line))
        elif t in pairs:
            assert stack and stack[-1][0] == pairs[t], f"line {line}: `{t}` closes nothing (open: {stack[-1] if stack else None})"
            stack.pop()
    assert not stack, f"never closed: {stack[-3:]}"


def _imports(toks):
    names, i = {}, 0
    while i < len(toks):
        if toks[i][:2] == ("id", "import"):
            if toks[i + 1][:2] == ("op", "("):
                j = i + 2
                while toks[j][:2] != ("op", ")"):
                    alias = None
                    if toks[j][0] == "id":
                        alias = toks[j][1]; j += 1
                    assert toks[j][0] == "str", f"line {toks[j][2]}: import path expected"
                    path = toks[j][1].strip('"')
                    names[alias or path.rsplit("/", 1)[-1]] = path
                    j += 1
                i = j
            elif toks[i + 1][0] == "str":
                path = toks[i + 1][1].strip('"')
                names[path.rsplit("/", 1)[-1]] = path
                i += 1
        i += 1
    return names


def test_imports_are_used_and_qualifiers_are_imported(toks):
    imp = _imports(toks)
    assert "C" in imp and "unsafe" in imp, imp             # cgo, and the pointer conversions at the boundary
    used = set()
    for k in range(len(toks) - 2):
        if toks[k][0] == "id" and toks[k + 1][:2] == ("op", ".") and toks[k + 2][0] == "id" and (k == 0 or toks[k - 1][:2] != ("op", ".")):
            used.add(toks[k][1])
    unused = [n for n in imp if n not in used]
    assert not unused, f"imported and not used (a compile error in Go): {unused}"
    # what the file qualifies with a standard-library package name must be imported
    std = {"image", "color", "log", "math", "os", "runtime", "sync", "unsafe", "fmt", "errors", "bytes", "sort", "strings", "atomic", "time",
           "jpeg", "png", "io", "context", "strconv", "reflect", "binary"}
    missing = sorted(n for n in used if n in std and n not in imp)
    assert not missing, f"used as a package and not imported: {missing}"


def _functions(toks):
    """top-level `func` bodies: (name, line, index of `{`, index of the matching `}`)"""
    out, depth, i = [], 0, 0
    while i < len(toks):
        kind, t, line = toks[i]
        if kind == "op" and t in "([{":
            depth += 1
        elif kind == "op" and t in ")]}":
            depth -= 1
        elif depth == 0 and (kind, t) == ("id", "func"):
            j = i + 1
            name = "<literal>"
            if toks[j][:2] == ("op", "("):                      # a receiver -- or the parameters of a function literal
                d, e = 0, j
                while True:
                    d += toks[e][1] == "("; d -= toks[e][1] == ")"; e += 1
                    if d == 0:
                        break
                if toks[e][0] == "id" and toks[e + 1][:2] == ("op", "("):
                    j = e
                    name = toks[j][1]
            else:
                name = toks[j][1]
            d = 0
            while not (toks[j][:2] == ("op", "{") and d == 0):  # the signature's parentheses
                d += toks[j][1] in "(["; d -= toks[j][1] in ")]"; j += 1
            k, d = j, 0
            while True:
                d += toks[k][1] == "{" and toks[k][0] == "op"; d -= toks[k][1] == "}" and toks[k][0] == "op"; k += 1
                if d == 0:
                    break
            out.append((name, line, j, k - 1))
            i = k                                               # past the body's closing brace
            continue
        i += 1
    return out


def test_locals_declared_are_read(toks):
    """`x := ...` / `var x ...` inside a function and no later mention of x in that function: `declared and not used`."""
    funcs = _functions(toks)
    assert len(funcs) >= 30, len(funcs)
    bad = []
    for name, line, b0, b1 in funcs:
        body = toks[b0:b1 + 1]
        for k, (kind, t, ln) in enumerate(body):
            declared = []
            if (kind, t) == ("op", ":="):
                # identifiers of the left-hand side: back to the statement's start, commas between
                j = k - 1
                while j >= 0 and (body[j][0] == "id" or body[j][:2] == ("op", ",")):
                    if body[j][0] == "id" and body[j][1] not in KEYWORDS:
                        declared.append(body[j][1])
                    j -= 1
                if j >= 0 and body[j][:2] == ("op", ".") :      # `a.b := ` cannot be: a selector on the left means this scan went too far
                    declared = []
            elif (kind, t) == ("id", "var") and body[k + 1][0] == "id":
                declared.append(body[k + 1][1])
            for d in declared:
                if d == "_":
                    continue
                later = [x for x in body[k + 1:] if x[0] == "id" and x[1] == d]
                # `if v := f(); v != 0` and `for i := range x` read the name right behind the declaration: counted by the same scan
                if not later:
                    bad.append(f"{name} (line {ln}): `{d}` declared and never read")
    assert not bad, bad


def test_every_function_with_results_ends_in_a_return(toks):
    """a function with result types whose body's last statement is neither `return` nor `panic(...)`: `missing return`."""
    funcs = _functions(toks)
    bad = []
    for name, line, b0, b1 in funcs:
        # results: anything between the parameter list's `)` and the body's `{`
        j = b0 - 1
        has_results = toks[j][:2] != ("op", ")") or False
        if toks[j][:2] == ("op", ")"):
            # `) {`: either no results, or a parenthesised result list -- look for `) (` pattern before it
            d, k = 0, j
            while True:
                d += toks[k][1] == ")"; d -= toks[k][1] == "("; k -= 1
                if d == 0:
                    break
            has_results = toks[k][:2] == ("op", ")")            # a second parenthesised group = the result list
        if not has_results:
            continue
        # the last top-level statement of the body
        depth, last_start = 0, b0 + 1
        prev_line = None
        for k in range(b0 + 1, b1):
            kind, t, ln = toks[k]
            if depth == 0 and prev_line is not None and ln != prev_line and not (toks[k - 1][0] == "op" and toks[k - 1][1] in ",(+-*/|&.=" or toks[k - 1][1] in ("&&", "||", ":=")):
                last_start = k
            if kind == "op" and t in "([{":
                depth += 1
            elif kind == "op" and t in ")]}":
                depth -= 1
            prev_line = ln
        first = toks[last_start][1]
        if first not in ("return", "panic"):
            # an if/else or switch whose every branch returns is legal; accept only when a `return` is the body's very last statement inside it
            tail = [t for _, t, _ in toks[last_start:b1]]
            # (an `if` terminates only through its `else`)
            if not (first in ("if", "switch", "for", "select") and "return" in tail and (first != "if" or "else" in tail)):
                bad.append(f"{name} (line {line}): ends in `{first}`")
    assert not bad, bad


def test_the_lint_is_not_vacuous():
    """the same checks on the shim with one mistake planted each: every one of them must be caught"""
    with open(SHIM, encoding="utf-8") as f:
        src = f.read()

    def fails(check, text):
        try:
            check(tokenize(text))
        except AssertionError:
            return True
        return False
    assert fails(test_imports_are_used_and_qualifiers_are_imported, src.replace('\t"os"\n', '\t"os"\n\t"strings"\n', 1))
    assert fails(test_imports_are_used_and_qualifiers_are_imported, src.replace('\t"math"\n', "", 1))
    assert fails(test_delimiters_balance, src.replace("func (p *hipPool) put(c *C.fnx_ctx) {", "func (p *hipPool) put(c *C.fnx_ctx) {{", 1))
    planted = src.replace("func fellBack(fn string) {", "func fellBack(fn string) {\n\tunusedLocal := 3", 1)
    assert planted != src and fails(test_locals_declared_are_read, planted)
    planted = src.replace("func pix(img *image.NRGBA) *C.uint8_t {", "func pix(img *image.NRGBA) *C.uint8_t {\n\tif img == nil {\n\t\treturn nil\n\t}\n}\nfunc pix2(img *image.NRGBA) *C.uint8_t {", 1)
    assert planted != src and fails(test_every_function_with_results_ends_in_a_return, planted)
