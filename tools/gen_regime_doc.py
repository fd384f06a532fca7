"""csrc/regimes.h REGIME_TABLE / enum Engine -> the generated comment blocks of include/mmscore.h (between the GENERATED markers).
usage: python tools/gen_regime_doc.py          rewrite the header in place
       python tools/gen_regime_doc.py --check  exit 1 if the header's blocks differ from what the table gives (tests/test_abi.py runs this)"""
import os
import re
import sys
import textwrap

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = os.path.join(R, "kddcup_2020_multimodalitiesrecall_2nd_place_amd", "csrc", "regimes.h")
HDR = os.path.join(R, "include", "mmscore.h")


def parse(src=None):
    """-> (constants {name: int}, table [(name, cmp, const name, rows, numerical, text)], engines [(name, value, comment)])"""
    s = src if src is not None else open(REG).read()
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"constexpr\s+int(?:64_t)?\s+(\w+)\s*=\s*(\d+)\s*;", s)}
    body = s[s.index("constexpr RegimeBound REGIME_TABLE[] = {"):s.index("constexpr int REGIME_COUNT")]
    table = []
    for m in re.finditer(r'\{"(\w+)",\s*"([<>=]+)",\s*(\w+),\s*(true|false),\s*((?:"(?:[^"\\]|\\.)*"\s*)+)\}', body):
        text = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', m.group(5)))
        table.append((m.group(1), m.group(2), m.group(3), consts[m.group(3)], m.group(4) == "true", text))
    enum = s[s.index("enum Engine : int {"):s.index("};", s.index("enum Engine : int {"))]
    engines = [(m.group(1), int(m.group(2)), m.group(3).strip()) for m in re.finditer(r"(ENG_\w+)\s*=\s*(\d+),\s*//\s*(.*)", enum)]
    engines.insert(0, ("ENG_AUTO", 0, "the per-shape choice of a forward (gemm_dispatch.hip pick_engine)"))
    return consts, table, engines


def blocks():
    _c, table, engines = parse()
    reg = []
    for name, cmp_, _cn, rows, numerical, text in table:
        head = " *   rows %-2s %-7d %-16s " % (cmp_, rows, name)
        tag = "" if numerical else "[speed choice only: bit-identical across it]  "
        lines = textwrap.wrap(tag + text, 150 - len(head), break_on_hyphens=False)
        reg.append(head + lines[0])
        reg += [" *" + " " * (len(head) - 2) + l for l in lines[1:]]
    eng = []
    for name, val, text in engines:
        if name == "ENG_DIAG_BASE":
            continue
        head = " *   %3d %-17s " % (val, name)
        lines = textwrap.wrap(text, 150 - len(head), break_on_hyphens=False)
        eng.append(head + lines[0])
        eng += [" *" + " " * (len(head) - 2) + l for l in lines[1:]]
    return {"REGIMES": "\n".join(reg), "ENGINES": "\n".join(eng)}


def render(hdr):
    for key, text in blocks().items():
        a, b = " * GENERATED %s BEGIN (tools/gen_regime_doc.py from csrc/regimes.h -- do not edit)\n" % key, " * GENERATED %s END\n" % key
        i, j = hdr.index(a) + len(a), hdr.index(b)
        hdr = hdr[:i] + text + "\n" + hdr[j:]
    return hdr


if __name__ == "__main__":
    cur = open(HDR).read()
    new = render(cur)
    if "--check" in sys.argv:
        sys.exit(0 if new == cur else 1)
    open(HDR, "w").write(new)
    print("include/mmscore.h: %s" % ("unchanged" if new == cur else "regime / engine blocks rewritten"))
