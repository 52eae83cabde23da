"""Rewrites the fenced blocks of INTEGRATION.md that show the Go shim so that they equal the files under shim/go/ verbatim.

A block is everything between `<!-- shim:<path> -->` and `<!-- /shim -->`; tests/test_go_shim_static.py fails when a block
and its file differ and names this script.  usage: python tools/go_static/sync_integration_md.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PAT = re.compile(r"(<!-- shim:(?P<path>[^ ]+) -->\n)(?P<body>.*?)(<!-- /shim -->)", re.S)


def render(path):
    with open(os.path.join(ROOT, path)) as fh:
        src = fh.read()
    lang = "go" if path.endswith(".go") else ""
    return f"```{lang}\n{src}```\n"


def blocks(doc):
    return [(m.group("path"), m.group("body")) for m in PAT.finditer(doc)]


def main():
    p = os.path.join(ROOT, "INTEGRATION.md")
    doc = open(p).read()
    new = PAT.sub(lambda m: m.group(1) + render(m.group("path")) + m.group(4), doc)
    if "--check" in sys.argv:
        sys.exit(0 if new == doc else 1)
    if new != doc:
        open(p, "w").write(new)
        print("INTEGRATION.md: shim blocks rewritten from shim/go/")


if __name__ == "__main__":
    main()
