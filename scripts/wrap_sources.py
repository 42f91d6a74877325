#!/usr/bin/env python
"""Wraps over-long lines of the repository's sources (product sources at <= 140 columns) WITHOUT changing what they mean,
and proves it:
  C / C++ / HIP   the token stream (comments and white space dropped, adjacent string literals merged) must be identical before and after;
  Python          the syntax tree (docstrings compared with white space collapsed) must be identical before and after;
a file whose check fails is left untouched and reported.  Lines that cannot be broken safely (inside raw strings, tables in comments wider
than the limit with no spaces, ...) are left as they are and listed.

What it does to a long line: a trailing comment moves to its own line(s) ABOVE the code; comment text is wrapped at spaces (a paragraph of
running text is re-wrapped as a whole, a table row keeps its columns); code is broken at the best of -- a statement boundary, the opening
brace of a one-line block, a comma (the shallower the better), a logical / ternary operator, an assignment, an arithmetic operator, any
space outside literals -- at or before the limit, continuation lines indented by 4 (a new statement keeps the indent); a preprocessor
definition gets backslashes; a string literal that alone exceeds the limit is split into adjacent literals at a space.  Python: breaks
only inside brackets, f-strings split outside their braces, docstring text wrapped.

usage: python scripts/wrap_sources.py [--limit 140] [--check] files..."""
import argparse
import ast
import io
import re
import sys
import tokenize

LIMIT = 140
MIN_FIRST = 24            # a break must leave at least this much on the first line (beyond its indent)


# ------------------------------------------------------------------------------------------------------------------------------------
# C-family lexing of one line
# ------------------------------------------------------------------------------------------------------------------------------------
def c_scan(line, in_block):
    """Marks every character of `line`: 'c' code, 's' inside a string / char literal (quotes included), 'L' line comment, 'B' block comment.
    Returns (marks, in_block at the end of the line)."""
    marks, i, n = [], 0, len(line)
    state = "B" if in_block else "c"
    quote = ""
    while i < n:
        ch = line[i]
        if state == "B":
            if line.startswith("*/", i):
                marks += ["B", "B"]; i += 2; state = "c"; continue
            marks.append("B"); i += 1; continue
        if state == "s":
            if ch == "\\" and i + 1 < n:
                marks += ["s", "s"]; i += 2; continue
            marks.append("s"); i += 1
            if ch == quote:
                state = "c"
            continue
        if line.startswith("//", i):
            marks += ["L"] * (n - i); i = n; break
        if line.startswith("/*", i):
            marks += ["B", "B"]; i += 2; state = "B"; continue
        if ch in "\"'":
            # a digit separator (1'000) is no char literal
            if ch == "'" and i > 0 and line[i - 1].isalnum() and i + 1 < n and line[i + 1].isalnum() and re.search(r"\d[\d']*$", line[:i]):
                marks.append("c"); i += 1; continue
            state, quote = "s", ch
            marks.append("s"); i += 1; continue
        marks.append("c"); i += 1
    return marks, state == "B"


def c_tokens(text):
    """Token stream of C-family source for the equivalence check: identifiers / numbers / punctuation, strings with adjacent literals
    merged; comments, white space and backslash-newlines dropped."""
    text = text.replace("\\\n", " ")
    out, in_block = [], False
    for line in text.split("\n"):
        directive = not in_block and line.lstrip().startswith("#")
        marks, in_block = c_scan(line, in_block)
        i, n = 0, len(line)
        while i < n:
            m = marks[i]
            if m in "LB" or line[i].isspace():
                i += 1; continue
            if m == "s":
                j = i
                while j < n and marks[j] == "s":
                    j += 1
                    if j < n and marks[j] == "s" and line[j - 1] == line[i] and j - i > 1 and line[j - 2] != "\\" and line[j] == line[i]:
                        break                                     # "a""b": two literals back to back
                lit = line[i:j]
                if out and out[-1][0] == "S" and lit[0] == '"' and out[-1][1][0] == '"':
                    out[-1] = ("S", out[-1][1][:-1] + lit[1:])    # adjacent string literals are one literal
                else:
                    out.append(("S", lit))
                i = j; continue
            if line[i].isalnum() or line[i] == "_":
                j = i
                while j < n and marks[j] == "c" and (line[j].isalnum() or line[j] in "_.'"):
                    j += 1
                out.append(("W", line[i:j])); i = j; continue
            out.append(("P", line[i])); i += 1
        if directive:
            out.append(("EOL", ""))                                # a preprocessor directive ends with its (logical) line
    return out


# ------------------------------------------------------------------------------------------------------------------------------------
# wrapping text and code
# ------------------------------------------------------------------------------------------------------------------------------------
def wrap_words(text, first_prefix, next_prefix, limit):
    words, lines, cur = text.split(" "), [], first_prefix
    empty = True
    for w in words:
        if not empty and len(cur) + 1 + len(w) > limit:
            lines.append(cur.rstrip()); cur, empty = next_prefix, True
        cur += ("" if empty else " ") + w
        empty = False
    lines.append(cur.rstrip())
    return lines


ASSIGN = re.compile(r" (=|\+=|-=|\*=|/=|\|=|&=|\^=|<<=|>>=) $")


def c_candidates(code, marks):
    """Break positions of a C-family code line: (position = index where the continuation starts, class weight, paren depth, brace depth)."""
    cands, depth, braces, angles = [], 0, 0, []
    n = len(code)
    for i, ch in enumerate(code):
        if marks[i] != "c":
            continue
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth = max(0, depth - 1)
        elif ch == "<" and i > 0 and (code[i - 1].isalnum() or code[i - 1] == "_") and code[i + 1:i + 2] not in (" ", "=", "<"):
            depth += 1; angles.append(depth)                    # a template argument list (binary operators are written with spaces here)
        elif ch == ">" and angles and angles[-1] == depth and code[i - 1] not in " -":
            depth -= 1; angles.pop()
        elif ch == "{":
            braces += 1
            if depth == 0 and code[i + 1:i + 2] == " " and i + 2 < n:
                cands.append((i + 2, 8.0, depth, braces))
        elif ch == "}":
            braces = max(0, braces - 1)
        elif ch == ";" and depth == 0 and code[i + 1:i + 2] == " " and i + 2 < n:
            cands.append((i + 2, 9.0, depth, braces))
        elif ch == "," and code[i + 1:i + 2] == " " and i + 2 < n:
            cands.append((i + 2, 6.0 - 1.5 * depth, depth, braces))
        elif ch == " " and i + 1 < n and marks[i + 1] == "c":
            rest = code[i + 1:]
            if rest.startswith(("&& ", "|| ")):
                cands.append((i + 1, 6.0 - 1.2 * depth, depth, braces))
            elif rest.startswith((": ", "? ")) and depth > 0 or rest.startswith("? "):
                cands.append((i + 1, 5.0 - 1.2 * depth, depth, braces))
            elif ASSIGN.search(code[:i + 1]) and not code[:i + 1].rstrip().endswith(("==", "<=", ">=", "!=")):
                cands.append((i + 1, 4.0 - 0.5 * depth, depth, braces))
            elif rest.startswith(("+ ", "- ", "* ", "/ ", "<< ", ">> ", "| ", "& ", "^ ", "< ", "> ", "<= ", ">= ", "== ", "!= ")):
                cands.append((i + 1, 3.0 - 0.5 * depth, depth, braces))
            else:
                cands.append((i + 1, 2.0 if depth == 0 else 1.0 - 0.2 * depth, depth, braces))
    return cands


def split_string_literal(code, marks, limit, indent):
    """`code` holds a string literal that runs past the limit: split it at a space inside into two adjacent literals."""
    i = limit - 2
    while i > len(indent) + MIN_FIRST:
        if marks[i] == "s" and code[i] == " " and marks[i + 1] == "s" and code[i - 1] != "\\":
            j = i
            while j >= 0 and marks[j] == "s":
                j -= 1
            if code[j + 1] == '"':                               # (not a char literal, not a prefixed / raw literal)
                return code[:i + 1] + '"', '"' + code[i + 1:]
        i -= 1
    return None


def break_c_code(code, indent, limit, suffix=""):
    """Pieces of one C-family code line, each (with `suffix` appended: the backslash of a preprocessor definition) within the limit where
    possible. A new statement of the line's own block goes back to `indent`, anything else continues at indent + 4."""
    pieces, room, cont, carry = [], limit - len(suffix), indent + "    ", 0
    while len(code) > room:
        marks, _ = c_scan(code, False)
        own = len(code) - len(code.lstrip())
        cands = [c for c in c_candidates(code, marks) if own + MIN_FIRST <= c[0] <= room and code[:c[0]].strip()]
        if cands:
            pos, w, depth, braces = max(cands, key=lambda c: (c[1] + 1.5 * c[0] / room, c[0]))
            head, tail = code[:pos].rstrip(), code[pos:]
            carry += braces
            same = w == 9.0 and carry == 0
        else:
            sp = split_string_literal(code, marks, room, code[:own])
            if not sp:
                break
            head, tail, same = sp[0], sp[1], False
        pieces.append(head + suffix)
        code = (indent if same else cont) + tail.lstrip()
    pieces.append(code)
    return pieces


COMMENT_LINE = re.compile(r"^(\s*)(//[/!]?)( ?)(.*)$")


def plain_text(body):
    """A comment line that may be joined with its neighbours: running text -- no table columns, no list item, no deeper indent."""
    return bool(body) and not body.startswith((" ", "-", "*", "#",
        "|")) and "   " not in body.rstrip() and not re.match(r"^(\d+\.|\w\)|[A-Za-z_]+:$)", body)


def wrap_comment_line(indent, lead, body, limit):
    """One over-long comment line on its own: a table row keeps its columns (the overflow hangs under the last column), text wraps."""
    m = None
    for m in re.finditer(r"\S {3,}(?=\S)", body):
        pass
    if m and len(indent + lead) + m.end() < limit - 40:
        hang = " " * m.end()
        return wrap_words(body[m.end():], indent + lead + body[:m.end()], indent + lead + hang, limit)
    return wrap_words(body, indent + lead, indent + lead, limit)


def reflow_comment_paragraphs(lines, limit):
    """Consecutive //-comment lines of running text (same indent) form a paragraph; a paragraph with an over-long line is re-wrapped as a
    whole."""
    out, i, in_block = [], 0, False
    while i < len(lines):
        m = COMMENT_LINE.match(lines[i]) if not in_block else None
        if not m or not plain_text(m.group(4)) or m.group(3) != " ":
            _, in_block = c_scan(lines[i], in_block)
            out.append(lines[i]); i += 1; continue
        j = i
        while j < len(lines):
            mj = COMMENT_LINE.match(lines[j])
            if not mj or mj.group(1) != m.group(1) or mj.group(2) != m.group(2) or mj.group(3) != " " or not plain_text(mj.group(4)):
                break
            j += 1
        block = lines[i:j]
        prefix = m.group(1) + m.group(2) + " "
        fitting = [len(l) for l in block if len(l) <= limit]
        width = max(fitting) if fitting else limit
        # the author's own paragraph breaks inside the block: a line that ends a sentence although the next line's first word would have
        # fitted
        para = []
        for k, l in enumerate(block):
            para.append(l)
            last = k == len(block) - 1
            if not last:
                first_word = block[k + 1][len(prefix):].split(" ")[0]
                ends = l.rstrip()[-1:] in ".:;)" and len(l.rstrip()) + 1 + len(first_word) <= width - 2
            if last or ends:
                if any(len(x) > limit for x in para):
                    out += wrap_words(" ".join(x[len(prefix):].rstrip() for x in para), prefix, prefix, limit)
                else:
                    out += para
                para = []
        i = j
    return out


def wrap_c_file(text, limit):
    out, in_block, skipped = [], False, []
    lines = reflow_comment_paragraphs(text.split("\n"), limit)
    skip_until = 0
    # the previous line ended with a backslash: this one belongs to its directive
    continued = False
    for ln, line in enumerate(lines, 1):
        if ln <= skip_until:
            continue
        in_macro, continued = continued, line.rstrip().endswith("\\")
        marks, after = c_scan(line, in_block)
        if len(line) <= limit:
            out.append(line); in_block = after; continue
        indent = line[:len(line) - len(line.lstrip())]
        stripped = line.strip()
        # 1. text inside a block comment
        if in_block or (stripped.startswith("/*") and all(m == "B" for m in marks[len(indent):])):
            body = stripped
            if body.startswith("* "):
                first, nxt = indent + "* ", indent + "* "; body = body[2:]
            elif body.startswith("/*"):
                first, nxt = indent, indent + (" * " if not in_block else "   ")
            else:
                first, nxt = indent, indent
            out += wrap_words(body, first, nxt, limit); in_block = after; continue
        # 2. a line comment on its own
        if stripped.startswith("//"):
            m = COMMENT_LINE.match(line)
            out += wrap_comment_line(indent, m.group(2) + m.group(3), m.group(4), limit); in_block = after; continue
        # 3. code, possibly with a trailing comment
        macro = stripped.startswith("#define") or line.rstrip().endswith("\\") or in_macro
        if stripped.startswith("#") and not macro:
            out.append(line); skipped.append(ln); in_block = after; continue
        code, comment = line.rstrip(), None
        if marks and marks[-1] == "L":
            k = marks.index("L")
            code, comment = line[:k].rstrip(), line[k:].strip()
            # comment lines below that continue this trailing comment (their // in the same column) move up with it
            nxt = ln
            while nxt < len(lines) and lines[nxt][:k].strip() == "" and lines[nxt][k:k + 2] == "//" and len(lines[nxt]) > k:
                comment += " " + lines[nxt][k + 2:].strip(); nxt += 1
            skip_until = nxt
        elif marks and marks[-1] == "B" and not after and line.rstrip().endswith("*/"):
            k = len(marks) - 1
            while k > 0 and marks[k - 1] == "B":
                k -= 1
            if line[:k].strip():
                code, comment = line[:k].rstrip(), line[k:].strip()
        if macro and comment:
            out.append(line); skipped.append(ln); in_block = after; continue
        if comment and comment.startswith("//"):
            m = re.match(r"(//[/!]?\s*)", comment)
            out += wrap_words(comment[len(m.group(1)):], indent + m.group(1), indent + m.group(1), limit)
        elif comment:
            inner = comment[2:-2].strip()
            w = wrap_words(inner, indent + "/* ", indent + " * ", limit - 3)
            w[-1] += " */"
            out += w
        if macro:
            body = code[:-1].rstrip() if code.endswith("\\") else code
            ends = code.endswith("\\")
            pcs = break_c_code(body, indent, limit, suffix=" \\")
            pcs = [p if p.endswith(" \\") else p + (" \\" if ends else "") for p in pcs]
            out += pcs
        else:
            out += break_c_code(code, indent, limit)
        in_block = after
    new = "\n".join(out)
    long_left = [i for i, l in enumerate(out, 1) if len(l) > limit]
    return new, long_left


# ------------------------------------------------------------------------------------------------------------------------------------
# Python
# ------------------------------------------------------------------------------------------------------------------------------------
def py_norm(tree):
    for node in ast.walk(tree):
        if isinstance(node, (ast.Module, ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)) and node.body and isinstance(node.body[0],
            ast.Expr) \
                and isinstance(getattr(node.body[0], "value", None), ast.Constant) and isinstance(node.body[0].value.value, str):
            node.body[0].value.value = " ".join(node.body[0].value.value.split())
    return ast.dump(tree)


STR_PREFIX = re.compile(r"^([rRbBfFuU]{0,2})(\"|')")


def split_py_string(tok, room):
    """Splits one single-quoted (not triple-quoted) string token into adjacent literals, the first at most `room` characters long."""
    m = STR_PREFIX.match(tok)
    if not m or tok[m.end() - 1] * 3 == tok[m.end() - 1:m.end() + 2]:
        return None
    prefix, q = m.group(1), m.group(2)
    is_f = "f" in prefix.lower()
    body_start = m.end()
    i = min(room - 1, len(tok) - 2)
    while i > body_start + 8:
        if tok[i] == " " and tok[i - 1] != "\\":
            if is_f and tok[body_start:i].count("{") - tok[body_start:i].count("}") != 0:
                i -= 1; continue
            return tok[:i + 1] + q, prefix + q + tok[i + 1:]
        i -= 1
    return None


def break_py_line(line, limit, toks, base_depth):
    """`toks`: the tokens of this physical line (type, string, start col, end col, depth BEFORE the token).  Returns the pieces."""
    indent = line[:len(line) - len(line.lstrip())]
    pieces, cont = [], indent + "    "
    cur_line, cur_toks = line, toks
    while len(cur_line) > limit:
        best = None
        for k, (typ, s, a, b, depth) in enumerate(cur_toks):
            if a < len(indent) + MIN_FIRST or k == 0:
                continue
            prev = cur_toks[k - 1]
            if a > limit:
                break
            w = None
            if depth > 0:
                if prev[1] == ",":
                    w = 6.0 - 0.8 * depth
                elif s in ("and", "or", "if", "else", "for") and typ == tokenize.NAME:
                    w = 5.0 - 0.8 * depth
                elif typ == tokenize.OP and s in ("+", "-", "*", "/", "%", "|", "&", "==", "!=", "<", ">", "<=",
                    ">=") and prev[0] != tokenize.OP:
                    w = 3.0 - 0.5 * depth
                elif prev[1] in ("(", "[", "{"):
                    w = 2.0 - 0.5 * depth
                elif typ == tokenize.STRING and prev[0] == tokenize.STRING:
                    w = 4.0 - 0.5 * depth
            if w is not None:
                score = w + 1.5 * a / limit
                if best is None or score >= best[0]:
                    best = (score, k)
        if best is None:
            # a string token that runs past the limit: split it
            done = False
            for k, (typ, s, a, b, depth) in enumerate(cur_toks):
                if typ == tokenize.STRING and a < limit - 16 and b > limit:
                    sp = split_py_string(s, limit - a - (0 if depth > 0 else 2))
                    if sp:
                        head = cur_line[:a] + sp[0] + ("" if depth > 0 else " \\")
                        tail_src = cont + sp[1] + cur_line[b:]
                        pieces.append(head)
                        shift = len(cont) + len(sp[1]) - b
                        cur_toks = [(typ, sp[1], len(cont), len(cont) + len(sp[1]), depth)] + [(t, s2, a2 + shift, b2 + shift, d2) for t,
                            s2, a2, b2, d2 in cur_toks[k + 1:]]
                        cur_line = tail_src
                        done = True
                    break
            if not done:
                break
            continue
        k = best[1]
        a = cur_toks[k][2]
        pieces.append(cur_line[:a].rstrip())
        shift = len(cont) - a
        cur_line = cont + cur_line[a:]
        cur_toks = [(t, s2, a2 + shift, b2 + shift, d2) for t, s2, a2, b2, d2 in cur_toks[k:]]
    pieces.append(cur_line)
    return pieces


def wrap_py_file(text, limit):
    lines = text.split("\n")
    # the lines of docstrings (their white space is free: the syntax-tree comparison collapses it); other multi-line strings are data
    doc_lines = set()
    for node in ast.walk(ast.parse(text)):
        if isinstance(node, (ast.Module, ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)) and node.body and isinstance(node.body[0],
            ast.Expr) \
                and isinstance(getattr(node.body[0], "value", None), ast.Constant) and isinstance(node.body[0].value.value, str):
            doc_lines.update(range(node.body[0].lineno, node.body[0].end_lineno + 1))
    # tokens per physical line with bracket depth; which lines lie inside a multi-line string
    per_line, in_string, depth = {}, set(), 0
    try:
        for tok in tokenize.generate_tokens(io.StringIO(text).readline):
            typ, s, (r0, c0), (r1, c1), _ = tok
            if typ == tokenize.STRING and r1 > r0:
                in_string.update(range(r0, r1 + 1))
            if typ in (tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENDMARKER):
                continue
            if r0 == r1:
                per_line.setdefault(r0, []).append((typ, s, c0, c1, depth))
            if typ == tokenize.OP and s in "([{":
                depth += 1
            elif typ == tokenize.OP and s in ")]}":
                depth -= 1
    except tokenize.TokenError:
        return text, []
    out = []
    for ln, line in enumerate(lines, 1):
        if len(line) <= limit:
            out.append(line); continue
        indent = line[:len(line) - len(line.lstrip())]
        stripped = line.strip()
        if ln in doc_lines:                                       # (a one-line docstring becomes a multi-line one)
            out += wrap_words(stripped, indent, indent, limit); continue
        if ln in in_string:
            out.append(line); continue
        if stripped.startswith("#"):
            m = re.match(r"(#+\s*)", stripped)
            out += wrap_words(stripped[len(m.group(1)):], indent + m.group(1), indent + m.group(1), limit); continue
        toks = per_line.get(ln, [])
        code = line
        if toks and toks[-1][0] == tokenize.COMMENT:
            c = toks[-1]
            comment, code, toks = c[1], line[:c[2]].rstrip(), toks[:-1]
            m = re.match(r"(#+\s*)", comment)
            if code.strip():
                out += wrap_words(comment[len(m.group(1)):], indent + m.group(1), indent + m.group(1), limit)
            else:
                out.append(line); continue
        out += break_py_line(code, limit, toks, 0)
    return "\n".join(out), [i for i, l in enumerate(out, 1) if len(l) > limit]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--limit", type=int, default=LIMIT)
    ap.add_argument("--check", action="store_true", help="report only")
    a = ap.parse_args()
    bad = 0
    for f in a.files:
        text = open(f).read()
        if not any(len(l) > a.limit for l in text.split("\n")):
            continue
        if f.endswith(".py"):
            new, left = wrap_py_file(text, a.limit)
            try:
                same = py_norm(ast.parse(text)) == py_norm(ast.parse(new))
            except SyntaxError as e:
                same = False; print(f"{f}: the wrapped text does not parse ({e})")
        else:
            new, left = wrap_c_file(text, a.limit)
            same = c_tokens(text) == c_tokens(new)
        before = sum(len(l) > a.limit for l in text.split("\n"))
        if not same:
            print(f"{f}: NOT EQUIVALENT after wrapping -- left untouched ({before} long lines)"); bad += 1; continue
        print(f"{f}: {before} long lines -> {len(left)}" + (f" (lines {left[:12]})" if left else ""))
        if not a.check:
            open(f, "w").write(new)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
