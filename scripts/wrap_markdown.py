#!/usr/bin/env python
"""Wraps the paragraphs and list items of Markdown files at 140 columns (code blocks and table rows are left as they are).
usage: python scripts/wrap_markdown.py files..."""
import re
import sys
import textwrap

for path in sys.argv[1:]:
    out, code = [], False
    for line in open(path).read().split("\n"):
        if line.startswith("```"):
            code = not code
        if code or line.startswith(("```", "|")) or len(line) <= 140:
            out.append(line); continue
        m = re.match(r"^(\s*(?:[-*]|\d+\.)\s+|\s+)", line)
        pre = m.group(1) if m else ""
        out += textwrap.wrap(line[len(pre):], width=138, initial_indent=pre, subsequent_indent=" " * len(pre), break_long_words=False,
                             break_on_hyphens=False)
    open(path, "w").write("\n".join(out))
    print(path, sum(len(l) > 140 for l in out), "lines over 140 left (tables / code)")
