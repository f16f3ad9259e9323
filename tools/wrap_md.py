"""Keeps a markdown file within a column limit (DESIGN.md at <= 200 columns): prose and list items are re-wrapped (continuation lines
indented under the item's text), and a table any of whose rows is longer than the limit becomes a nested list — one top-level item per row, headed by the
first column, one sub-item per further column ("<column header>: <cell>") — because a markdown table row cannot be wrapped.  Code fences, headings and
tables that fit are left alone.   python tools/wrap_md.py DESIGN.md [--width 200] [--check]"""
import re
import sys
import textwrap


def split_row(line):
    cells, cur, in_code, i = [], "", False, 0
    line = line.strip()
    if line.startswith("|"):
        line = line[1:]
    if line.endswith("|") and not line.endswith("\\|"):
        line = line[:-1]
    while i < len(line):
        ch = line[i]
        if ch == "`":
            in_code = not in_code
            cur += ch
        elif ch == "\\" and i + 1 < len(line) and line[i + 1] == "|":
            cur += "\\|"
            i += 1
        elif ch == "|" and not in_code:
            cells.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    cells.append(cur.strip())
    return cells


def wrap(text, width, first, rest):
    return textwrap.fill(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def convert_table(rows, width):
    header = split_row(rows[0])
    out = []
    for r in rows[2:]:
        cells = split_row(r)
        out.append(wrap("**" + (cells[0] if cells else "") + "**" + (" (" + header[0] + ")" if header and header[0] and not cells[0].startswith("**") else ""), width, "- ", "  "))
        for h, c in zip(header[1:], cells[1:]):
            if c.strip():
                out.append(wrap((h + ": " if h else "") + c, width, "  - ", "    "))
    return out


def process(lines, width):
    out, i, in_code = [], 0, False
    while i < len(lines):
        ln = lines[i]
        if ln.lstrip().startswith("```"):
            in_code = not in_code
            out.append(ln)
            i += 1
            continue
        if in_code or ln.startswith("#"):
            out.append(ln)
            i += 1
            continue
        if ln.lstrip().startswith("|") and i + 1 < len(lines) and re.match(r"^\s*\|[\s:|-]+\|\s*$", lines[i + 1]):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            tbl = lines[i:j]
            if max(len(t) for t in tbl) <= width:
                out.extend(tbl)
            elif max(len(t) for t in tbl[2:] or [""]) <= width - 20:
                # only the header is too long (a numeric table): the column titles move into a legend, the header row keeps letters
                header = split_row(tbl[0])
                letters = [chr(ord("A") + k) if k < 26 else "A" + chr(ord("A") + k - 26) for k in range(len(header))]
                out.append(wrap("Columns — " + "; ".join("**%s** = %s" % (l, h) for l, h in zip(letters, header)) + ".", width, "", ""))
                out.append("")
                out.append("| " + " | ".join(letters) + " |")
                out.extend(tbl[1:])
            else:
                out.extend(convert_table(tbl, width))
                out.append("")
            i = j
            continue
        if not ln.strip():
            out.append(ln)
            i += 1
            continue
        # a paragraph or a list item with its continuation lines: re-flowed as a whole (so that re-wrapping an edited, already wrapped text leaves no ragged lines)
        m = re.match(r"^(\s*)((?:[-*]|\d+\.)\s+)?", ln)
        indent, bullet = m.group(1), m.group(2) or ""
        cont = indent + " " * len(bullet)
        text, j = ln[len(indent) + len(bullet):].strip(), i + 1
        while j < len(lines):
            nx = lines[j]
            if (not nx.strip() or nx.startswith("#") or nx.lstrip().startswith("|") or nx.lstrip().startswith("```") or re.match(r"^\s*(?:[-*]|\d+\.)\s+", nx)
                    or (len(nx) - len(nx.lstrip())) != len(cont)):
                break
            text += " " + nx.strip()
            j += 1
        out.append(wrap(text, width, indent + bullet, cont))
        i = j
    return out


def main():
    path = sys.argv[1]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 200
    lines = open(path).read().split("\n")
    new = "\n".join(process(lines, width))
    over = [k + 1 for k, l in enumerate(new.split("\n")) if len(l) > width]
    if "--check" in sys.argv:
        print("%d lines over %d columns" % (len([l for l in lines if len(l) > width]), width))
        return
    open(path, "w").write(new)
    print("%s: %d lines, %d still over %d columns %s" % (path, new.count("\n") + 1, len(over), width, over[:10]))


if __name__ == "__main__":
    main()
