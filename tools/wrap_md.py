"""Re-wraps the prose of a markdown file at a column limit (tables, headings, code fences and HTML are left alone).

    python tools/wrap_md.py FILE [--width 120] [--check]

List items keep their marker and get a hanging indent; a line that ends in two blanks (a hard break) ends its paragraph."""
import re
import sys
import textwrap

ITEM = re.compile(r"^(\s*)([*+-]|\d+[.)])(\s+)")


def wrap_text(text, width):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = ITEM.match(first)
        if m:
            lead = m.group(0)
            hang = " " * len(lead)
            body = " ".join([first[len(lead):].strip()] + [p.strip() for p in para[1:]])
        else:
            lead = hang = re.match(r"^\s*", first).group(0)
            body = " ".join(p.strip() for p in para)
        out.extend(textwrap.wrap(body, width=width, initial_indent=lead, subsequent_indent=hang, break_long_words=False,
                                 break_on_hyphens=False) or [lead.rstrip()])
        para.clear()

    for line in text.split("\n"):
        stripped = line.strip()
        if stripped.startswith("```"):
            flush()
            fence = not fence
            out.append(line)
            continue
        if fence or not stripped or stripped.startswith(("|", "#", "<", ">", "---", "===")) or line.startswith("    "):
            flush()
            out.append(line)
            continue
        if ITEM.match(line) and para:
            flush()
        para.append(line)
        if line.endswith("  "):
            flush()
    flush()
    return "\n".join(out)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 120
    if "--width" in sys.argv:
        args.remove(str(width))
    bad = 0
    for path in args:
        text = open(path).read()
        new = wrap_text(text, width)
        if "--check" in sys.argv:
            over, fence = [], False
            for i, l in enumerate(text.split("\n")):
                if l.strip().startswith("```"):
                    fence = not fence
                elif not fence and len(l) > width and not l.lstrip().startswith("|") and not l.startswith("    "):
                    over.append(i + 1)
            if over:
                bad += 1
                print(f"{path}: {len(over)} prose lines over {width} columns (first: {over[:5]})")
        elif new != text:
            open(path, "w").write(new)
            print("wrapped", path)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
