#!/usr/bin/env python
"""Re-wraps the paragraphs and list items of Markdown files so that no line exceeds 140 columns -- counted in BYTES of UTF-8 as `awk 'length
> 140'` counts them in the C locale (an arrow or a multiplication sign is three bytes) -- leaving code blocks, tables and headings as they
are. Consecutive lines of a paragraph (or of a list item with its hanging indent) are joined first, so repeated runs are stable. usage:
python scripts/wrap_markdown.py files..."""
import re
import sys

LIMIT = 140
ITEM = re.compile(r"^(\s*)([-*]|\d+\.)\s+")


def blen(s):
    return len(s.encode("utf-8"))


def wrap(text, first, hang):
    lines, cur = [], first
    empty = True
    for w in text.split():
        if not empty and blen(cur) + 1 + blen(w) > LIMIT:
            lines.append(cur); cur, empty = hang, True
        cur += ("" if empty else " ") + w
        empty = False
    lines.append(cur)
    return lines


def reflow(lines):
    out, i, code = [], 0, False
    while i < len(lines):
        line = lines[i]
        if line.startswith("```"):
            code = not code
        if code or line.startswith(("```", "|", "#", ">")) or not line.strip():
            out.append(line); i += 1; continue
        m = ITEM.match(line)
        first = m.group(0) if m else line[:len(line) - len(line.lstrip())]
        hang = " " * len(first)
        block = [line[len(first):]]
        j = i + 1
        while j < len(lines) and lines[j].strip() and not lines[j].startswith(("```", "|", "#", ">")) and not ITEM.match(lines[j]) \
                and lines[j].startswith(hang) and (not hang or not lines[j][len(hang):].startswith(" ")):
            block.append(lines[j][len(hang):]); j += 1
        if any(blen(l) > LIMIT for l in lines[i:j]):
            out += wrap(" ".join(block), first, hang)
        else:
            out += lines[i:j]
        i = j
    return out


for path in sys.argv[1:]:
    new = reflow(open(path).read().split("\n"))
    open(path, "w").write("\n".join(new))
    print(path, sum(blen(l) > LIMIT for l in new), "lines over 140 bytes left (tables / code)")
