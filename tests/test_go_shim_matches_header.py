"""A mechanical check of go/fennec_hip.go against include/fennec_hip.h.

There is no Go toolchain in the image, so the cgo shim has never met a compiler.  What can be checked without one:

* every `C.fnx_*` / `C.fennec_*` call names a function the header declares, with the header's argument count;
* every `C.FNX_*` constant is a `#define` of the header, every `C.fnx_*` type a type of the header;
* the Go functions that shadow the reference's (ssim.go:24,48,244,313, resize.go:37,
  effects.go:10,49,146, exif.go:178, analyze.go:26, targetsize.go:488) carry the reference's signatures
  character for character -- the table below is data copied from the reference's declarations, and where
  /root/reference is present (this container, not the GPU box) it is re-read from there.

Any drift between header and shim fails here.
"""
from __future__ import annotations

import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "go", "fennec_hip.go")
HEADER = os.path.join(ROOT, "include", "fennec_hip.h")

# the reference's declarations the shim must keep (file, line at the survey commit, signature)
REFERENCE_SIGNATURES = {
    "SSIM": ("ssim.go", "func SSIM(img1, img2 image.Image) float64"),
    "SSIMFast": ("ssim.go", "func SSIMFast(img1, img2 *image.NRGBA) float64"),
    "MSSSIM": ("ssim.go", "func MSSSIM(img1, img2 image.Image) float64"),
    "boxDownsample": ("ssim.go", "func boxDownsample(img *image.NRGBA, dstW, dstH int) *image.NRGBA"),
    "lanczosResize": ("resize.go", "func lanczosResize(img *image.NRGBA, dstW, dstH int) *image.NRGBA"),
    "GaussianBlur": ("effects.go", "func GaussianBlur(img *image.NRGBA, sigma float64) *image.NRGBA"),
    "Sharpen": ("effects.go", "func Sharpen(img *image.NRGBA, strength float64) *image.NRGBA"),
    "AdaptiveSharpen": ("effects.go", "func AdaptiveSharpen(img *image.NRGBA, strength float64) *image.NRGBA"),
    "ApplyOrientation": ("exif.go", "func ApplyOrientation(img *image.NRGBA, orient Orientation) *image.NRGBA"),
    "Analyze": ("analyze.go", "func Analyze(img image.Image) ImageStats"),
    "applyPalette": ("targetsize.go", "func applyPalette(src *image.NRGBA, palette color.Palette) *image.Paletted"),
}


def _strip_c_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _strip_go_comments_and_strings(text: str) -> str:
    """Blank out // and /* */ comments, "..." and `...` strings and rune literals, keeping offsets' line structure."""
    out = []
    i, n = 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            j = text.find("\n", i)
            j = n if j < 0 else j
            i = j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("\n" * text.count("\n", i, j))
            i = j
        elif c == '"':
            j = i + 1
            while j < n and text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            out.append('""')
            i = j + 1
        elif c == "`":
            j = text.find("`", i + 1)
            j = n if j < 0 else j
            out.append('""' + "\n" * text.count("\n", i, j))
            i = j + 1
        elif c == "'":
            j = i + 1
            while j < n and text[j] != "'":
                j += 2 if text[j] == "\\" else 1
            out.append("0")
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def _split_top_level(args: str) -> list[str]:
    parts, depth, cur = [], 0, []
    for ch in args:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    last = "".join(cur).strip()
    if last or parts:
        parts.append(last)
    return parts


def _matching_paren(text: str, open_at: int) -> int:
    depth = 0
    for j in range(open_at, len(text)):
        if text[j] == "(":
            depth += 1
        elif text[j] == ")":
            depth -= 1
            if depth == 0:
                return j
    raise AssertionError("unbalanced parentheses")


def header_prototypes() -> dict[str, int]:
    """name -> parameter count of every function the header declares."""
    text = _strip_c_comments(open(HEADER).read())
    protos: dict[str, int] = {}
    for m in re.finditer(r"\b((?:fnx|fennec)_\w+)\s*\(", text):
        name = m.group(1)
        close = _matching_paren(text, m.end() - 1)
        tail = text[close + 1:close + 40].lstrip()
        if not tail.startswith(";"):
            continue                                   # a function-pointer typedef or a macro use, not a prototype
        params = text[m.end():close].strip()
        n = 0 if params in ("", "void") else len(_split_top_level(params))
        assert protos.get(name, n) == n, f"{name} declared twice with different arity"
        protos[name] = n
    return protos


def header_defines() -> set[str]:
    return set(re.findall(r"^\s*#\s*define\s+(FNX_\w+)", open(HEADER).read(), flags=re.M))


def header_types() -> set[str]:
    text = _strip_c_comments(open(HEADER).read())
    names = set(re.findall(r"\btypedef\s+struct\s+\w+\s+(\w+)\s*;", text))
    names |= set(re.findall(r"\}\s*((?:fnx|fennec)_\w+)\s*;", text))
    names |= set(re.findall(r"\bstruct\s+((?:fnx|fennec)_\w+)", text))
    return names


def shim_text() -> str:
    return _strip_go_comments_and_strings(open(SHIM).read())


def shim_calls() -> list[tuple[str, int, int]]:
    """(name, argument count, line) of every C.fnx_* / C.fennec_* call in the shim."""
    text = shim_text()
    calls = []
    for m in re.finditer(r"\bC\.((?:fnx|fennec)_\w+)\s*\(", text):
        close = _matching_paren(text, m.end() - 1)
        args = text[m.end():close].strip()
        n = 0 if args == "" else len(_split_top_level(args))
        calls.append((m.group(1), n, text.count("\n", 0, m.start()) + 1))
    return calls


def test_parsers_see_the_files():
    protos = header_prototypes()
    assert len(protos) >= 80, "the header's prototypes were not parsed"
    assert protos["fnx_ctx_create"] == 2 and protos["fnx_device_count"] == 0
    calls = shim_calls()
    assert len(calls) >= 30, "the shim's C calls were not parsed"


def test_every_c_call_of_the_shim_is_declared_with_the_same_arity():
    protos = header_prototypes()
    types = header_types()
    bad = []
    for name, nargs, line in shim_calls():
        if name in types and name not in protos:
            continue                                   # a conversion C.fnx_type(x)
        if name not in protos:
            bad.append(f"go/fennec_hip.go:{line}: C.{name} is not declared in include/fennec_hip.h")
        elif protos[name] != nargs:
            bad.append(f"go/fennec_hip.go:{line}: C.{name} called with {nargs} arguments, the header declares {protos[name]}")
    assert not bad, "\n".join(bad)


def test_every_c_constant_and_type_of_the_shim_exists():
    text = shim_text()
    defines, types, protos = header_defines(), header_types(), header_prototypes()
    consts = set(re.findall(r"\bC\.(FNX_\w+)", text))
    assert consts, "no C.FNX_* constant found in the shim"
    assert not consts - defines, f"constants the header does not define: {sorted(consts - defines)}"
    idents = set(re.findall(r"\bC\.((?:fnx|fennec)_\w+)", text))
    unknown = {i for i in idents if i not in protos and i not in types}
    assert not unknown, f"C identifiers the header does not declare: {sorted(unknown)}"


def test_the_shim_calls_go_through_real_exports():
    """the symbols the shim binds are in the built library's export list (the header's, checked by test_abi)"""
    import fennec_amd
    exported = set(fennec_amd.exported_symbols())
    used = {name for name, _, _ in shim_calls() if name in header_prototypes()}
    assert used <= exported


@pytest.mark.parametrize("name", sorted(REFERENCE_SIGNATURES))
def test_shadowed_functions_keep_the_reference_signature(name):
    _, want = REFERENCE_SIGNATURES[name]
    src = open(SHIM).read()
    decls = [ln.rstrip() for ln in src.splitlines() if re.match(rf"func {name}\(", ln)]
    assert len(decls) == 1, f"{name}: {len(decls)} declarations in the shim"
    assert decls[0] == want + " {", f"{name}: shim has `{decls[0]}`, the reference `{want}`"


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is only present in the build container")
def test_signature_table_is_the_reference_s():
    for name, (fname, want) in REFERENCE_SIGNATURES.items():
        lines = open(os.path.join("/root/reference", fname)).read().splitlines()
        assert want + " {" in lines, f"{fname}: `{want}` not found -- the table above is stale"


def test_every_fallback_names_a_go_twin_and_is_counted():
    """each shadowed function falls back to the reference's renamed Go body (`...Go(`) through fellBack(): never silently"""
    src = open(SHIM).read()

    def body_of(fn):
        m = re.search(rf"^func {fn}\(.*?^}}", src, flags=re.S | re.M)
        return m.group(0) if m else ""

    for name in REFERENCE_SIGNATURES:
        body = body_of(name)
        assert body, name
        # helpers shared by two entry points (Sharpen / AdaptiveSharpen -> sharpenHIP) hold part of the path
        for helper in set(re.findall(r"\b(\w+HIP)\(", body)):
            body += body_of(helper)
        assert re.search(r"\b\w+Go\(", body), f"{name}: no fallback to the reference's Go body"
        assert "fellBack(" in body, f"{name}: fallback is not counted"
