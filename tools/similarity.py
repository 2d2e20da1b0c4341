#!/usr/bin/env python3
"""Function-level similarity of tike_amd/ to the reference (build container only).

For every function / method of `tike_amd/` with at least MIN_LINES body lines,
compare with EVERY function under the reference's `src/tike` whose token count
is within 0.5x .. 2x (round 5; rounds 3-4: namesakes only, `--namesakes-only`)
and report the highest similarity of their normalised sources: docstrings
stripped, `ast.unparse` formatting (so comments, blank lines and line breaks do
not count), `tike.` / `tike_amd.` prefixes removed, compared with
`difflib.SequenceMatcher` over the token stream.  This is the check the
round-3 verdict asked for ("similarity of every function >= 8 lines to its
reference namesake < 0.6"); it reads /root/reference and therefore never runs
on the GPU box or in the tests.

    python tools/similarity.py [--threshold 0.6] [--min-lines 8] [--all]
"""
import argparse
import ast
import difflib
import io
import os
import sys
import tokenize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/src/tike"


def _strip_docstrings(node):
    for n in ast.walk(node):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef,
                          ast.Module)):
            body = n.body
            if (body and isinstance(body[0], ast.Expr)
                    and isinstance(body[0].value, ast.Constant)
                    and isinstance(body[0].value.value, str)):
                n.body = body[1:] or [ast.Pass()]
    return node


def _tokens(fn, body_only=False):
    """Normalised token stream of a function (signature annotations dropped);
    body_only: without the `def name(parameters):` header, which the drop-in
    contract fixes."""
    fn = _strip_docstrings(fn)
    if body_only:
        fn = ast.Module(body=fn.body, type_ignores=[])
        text = ast.unparse(fn).replace("tike_amd.", "").replace("tike.", "")
        return _lex(text)
    fn.returns = None
    for a in (fn.args.args + fn.args.kwonlyargs + fn.args.posonlyargs +
              [x for x in (fn.args.vararg, fn.args.kwarg) if x]):
        a.annotation = None
    fn.decorator_list = []
    text = ast.unparse(fn).replace("tike_amd.", "").replace("tike.", "")
    return _lex(text)


def _lex(text):
    out = []
    for tok in tokenize.generate_tokens(io.StringIO(text).readline):
        if tok.type in (tokenize.NEWLINE, tokenize.NL, tokenize.INDENT,
                        tokenize.DEDENT, tokenize.COMMENT, tokenize.ENDMARKER):
            continue
        out.append(tok.string)
    return out


def _functions(path):
    try:
        tree = ast.parse(open(path).read())
    except SyntaxError:
        return
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            body = _strip_docstrings(node).body
            lines = (body[-1].end_lineno - body[0].lineno + 1) if body else 0
            yield node.name, lines, node


def _walk(root):
    for d, _, files in os.walk(root):
        if "__pycache__" in d:
            continue
        for f in sorted(files):
            if f.endswith(".py"):
                yield os.path.join(d, f)


def _anonymise(tokens):
    """Identifiers -> ID, numbers -> NUM (keywords and operators kept): what
    is left is the statement structure."""
    import keyword
    out = []
    for t in tokens:
        if t.isidentifier() and not keyword.iskeyword(t):
            out.append("ID")
        elif t[:1].isdigit():
            out.append("NUM")
        else:
            out.append(t)
    return out


def _best(mine, candidates, floor=0.0):
    """Highest SequenceMatcher ratio of `mine` against the candidate token
    streams [(tokens, tag)], pruned by the cheap upper bounds."""
    best, tag = floor, None
    for toks, t in candidates:
        m = difflib.SequenceMatcher(None, mine, toks, autojunk=False)
        if m.real_quick_ratio() <= best or m.quick_ratio() <= best:
            continue
        r = m.ratio()
        if r > best:
            best, tag = r, t
    return best, tag


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threshold", type=float, default=0.6)
    ap.add_argument("--min-lines", type=int, default=8)
    ap.add_argument("--all", action="store_true",
                    help="list every compared function, not only offenders")
    ap.add_argument("--namesakes-only", action="store_true",
                    help="rounds 3-4: compare with reference functions of the "
                    "same name only")
    ap.add_argument("--package", default=os.path.join(ROOT, "tike_amd"))
    a = ap.parse_args()
    if not os.path.isdir(REFERENCE):
        print(f"{REFERENCE} not present (build container only)")
        return 0
    import copy
    ref = []  # (name, where, whole tokens, body tokens)
    for path in _walk(REFERENCE):
        for name, lines, node in _functions(path):
            ref.append((name, f"{os.path.relpath(path, REFERENCE)}:{node.lineno}",
                        _tokens(copy.deepcopy(node)), _tokens(node, True)))
    rows = []
    for path in _walk(a.package):
        for name, lines, node in _functions(path):
            if lines < a.min_lines:
                continue
            mine = _tokens(copy.deepcopy(node))
            body = _tokens(node, True)
            # EVERY reference function of comparable size (0.5x .. 2x the
            # tokens), not only the namesakes: a renamed copy has no namesake
            pool = [r for r in ref
                    if (r[0] == name if a.namesakes_only else
                        0.5 * len(mine) <= len(r[2]) <= 2.0 * len(mine))]
            if not pool:
                continue
            whole, where = _best(mine, [(r[2], r[1]) for r in pool])
            bod, _ = _best(body, [(r[3], r[1]) for r in pool])
            anon, awhere = _best(_anonymise(body),
                                 [(_anonymise(r[3]), r[1]) for r in pool])
            rows.append((whole, os.path.relpath(path, ROOT), node.lineno, name,
                         lines, where, bod, anon, awhere))
    rows.sort(reverse=True)
    bad = [r for r in rows if r[0] >= a.threshold or r[6] >= a.threshold]
    for r in (rows if a.all else bad):
        print(f"{r[0]:.3f} (body {r[6]:.3f}, names anonymised {r[7]:.3f})  "
              f"{r[1]}:{r[2]}  {r[3]} ({r[4]} lines)  vs  {r[5]}")
    top = sorted(rows, key=lambda r: -r[7])[:5]
    print("highest with identifiers anonymised (structure only; informative): "
          + "; ".join(f"{r[3]} {r[7]:.2f} vs {r[8]}" for r in top))
    print(f"{len(rows)} functions of >= {a.min_lines} lines compared with "
          f"{'their reference namesakes' if a.namesakes_only else 'every reference function of comparable size'}"
          f"; {len(bad)} at or above {a.threshold}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
